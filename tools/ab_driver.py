#!/usr/bin/env python3
"""A/B of an environment switch the library reads per pass, inside ONE process (the clock a box holds drifts by more than
the effects worth measuring: alternate short blocks and compare medians).
usage: ab_driver.py VAR [base=512] [box=128] [blocks=8] [steps=15] [valA=1] [valB=0] [threshold=-1]
(box = 0: the irregular hierarchy of bench.py's secondary.irregular_amr instead of the nested one)"""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peleanalysis_amd import capi  # noqa: E402
from peleanalysis_amd.hierarchy import MultiFab, field_flame, nested_hierarchy, tagged_hierarchy  # noqa: E402

var = sys.argv[1]
base = int(sys.argv[2]) if len(sys.argv) > 2 else 512
box = int(sys.argv[3]) if len(sys.argv) > 3 else 128
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 8
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 15
VA = sys.argv[6] if len(sys.argv) > 6 else "1"
VB = sys.argv[7] if len(sys.argv) > 7 else "0"
THR = float(sys.argv[8]) if len(sys.argv) > 8 else -1.0
if box:
    H = nested_hierarchy(base, 3, box, is_per=(1, 1, 0))
else:
    H = tagged_hierarchy(base, 3, lambda x, y, z: field_flame(x, y, z, 0), bf=16, max_box=128, base_box=128, frac=(0.08, 0.16), is_per=(1, 1, 0))
if os.environ.get("PA_AB_RETILE", "0") == "1":  # swept on the internal tiling of the tools and bench.py
    from peleanalysis_amd.hierarchy import retile_hierarchy  # noqa: E402
    H = retile_hierarchy(H)
bc = capi.bc_from_flags((1, 1, 0))
ctx = capi.Context(0)
dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
rng = np.random.default_rng(1)
states, works, outs = [], [], []
for lv, dl in zip(H.levels, dls):
    s = MultiFab(lv, 1, 2)
    s.data[:] = np.resize(300.0 + 1700.0 * rng.random(1 << 22), s.total)
    states.append(capi.DevMF.from_host(ctx, dl, s))
    works.append(capi.DevMF(ctx, dl, 1, 2))
    outs.append(capi.DevMF(ctx, dl, 8, 0))
params = capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=(THR if THR >= 0 else None), fused=True)


GRAD = os.environ.get("PA_AB_GRAD", "0") == "1"  # time the gradient tool's pass (pa_grad_run) instead of grad->curvature
if GRAD:
    gouts = [capi.DevMF(ctx, dl, 4, 0) for dl in dls]


def one():
    if GRAD:
        capi.grad_run(ctx, states, 0, bc, gouts, 0)
    else:
        capi.gradcurv_run(ctx, states, 0, bc, params, works, outs, 0)


def block(val):
    os.environ[var] = val  # "1" vs "0": for a switch that is off by default, 0 is the current behaviour
    ctx.lib.pa_options_reload()  # the library reads its switches once (pa_options): re-read after the flip
    one()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    ctx.sync()
    return (time.perf_counter() - t0) / steps * 1e3


res = {VA: [], VB: []}
for b in range(blocks):
    for v in (VA, VB) if b % 2 == 0 else (VB, VA):
        res[v].append(block(v))
for v in (VA, VB):
    print(f"{var}={v}: median {statistics.median(res[v]):.3f} ms/step  (min {min(res[v]):.3f}, max {max(res[v]):.3f}, {len(res[v])} blocks of {steps})")
print(f"difference of medians ({VB} - {VA}): {statistics.median(res[VB]) - statistics.median(res[VA]):+.3f} ms")
