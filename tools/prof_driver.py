#!/usr/bin/env python3
"""Lean driver for rocprofv3 runs: fused grad->curvature over a 3-level hierarchy through the
C ABI only (no torch, few dispatches).  usage: prof_driver.py [base=256] [box=128] [steps=3] [retile=1]
retile = 1: swept on the internal tiling the tools and bench.py use (pa_level_retile of the box^3 tiling), 0: on the box^3 tiling itself"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peleanalysis_amd import capi  # noqa: E402
from peleanalysis_amd.hierarchy import MultiFab, nested_hierarchy  # noqa: E402

base = int(sys.argv[1]) if len(sys.argv) > 1 else 256
box = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
H = nested_hierarchy(base, 3, box, is_per=(1, 1, 0))
if (int(sys.argv[4]) if len(sys.argv) > 4 else 1):
    from peleanalysis_amd.hierarchy import retile_hierarchy  # noqa: E402
    H = retile_hierarchy(H)
bc = capi.bc_from_flags((1, 1, 0))
ctx = capi.Context(0)
dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
rng = np.random.default_rng(1)
states, works, outs = [], [], []
for lv, dl in zip(H.levels, dls):
    s = MultiFab(lv, 1, 2)
    blk = 300.0 + 1700.0 * rng.random(1 << 22)  # incompressible block, tiled (a full-size draw costs seconds per pass)
    s.data[:] = np.resize(blk, s.total)
    states.append(capi.DevMF.from_host(ctx, dl, s))
    works.append(capi.DevMF(ctx, dl, 1, 2))
    outs.append(capi.DevMF(ctx, dl, 8, 0))
params = capi.curv_params(prog_min=300.0, prog_max=2000.0, fused=True)
if os.environ.get("PA_PROF_MODE", "") == "options":
    # pa_curvature_run with do_gaussCurv + do_strain + do_velnormal (bench.py's secondary.f1_curvature_options_headline): progress source + 3 velocity
    # components in, 8 fields out -- `PA_PROF_MODE=options tools/prof.sh kernels <tag>` gives time + HBM traffic of every kernel of that pass
    st4, out8 = [], []
    for lv, dl in zip(H.levels, dls):
        s = MultiFab(lv, 4, 2)
        s.data[:] = np.resize(300.0 + 1700.0 * rng.random(1 << 22), s.total)
        st4.append(capi.DevMF.from_host(ctx, dl, s))
        out8.append(capi.DevMF(ctx, dl, 8, 0))
    popt = capi.curv_params(prog_min=300.0, prog_max=2000.0, fused=True, vel_comp=1, do_gauss=True, do_strain=True, do_velnormal=True)
    run = lambda: capi.curvature_run(ctx, st4, 0, bc, popt, out8, 0)
else:
    run = lambda: capi.gradcurv_run(ctx, states, 0, bc, params, works, outs, 0)
run()
ctx.sync()
t0 = time.perf_counter()
for _ in range(steps):
    run()
ctx.sync()
dt = (time.perf_counter() - t0) / steps
cells = sum(lv.ncells for lv in H.levels)
print(f"base {base} box {box}: {dt*1e3:.3f} ms/step, {cells/dt/1e6:.0f} Mcells/s")
