#!/usr/bin/env python3
"""Experiment: the fused sweep on one n^3 level tiled by boxes of bx x by x bz cells.  All tilings are built first and timed
in interleaved rounds (run-to-run differences of a box / of the clock state are larger than some of the effects looked for);
prints per tiling the sweep's kernel time of every round and the median.   usage: zlen.py n shape [shape ...]   shape = 32x32x64"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import Level, mf_layout
import bench
if os.environ.get("PA_LIB"):  # a diagnostic build of the library (tools/exp/_dbg/*.so)
    capi.LIB_PATH = os.environ["PA_LIB"]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(32, 32, 32), (32, 32, 64), (64, 64, 64), (128, 128, 128)]
dev = torch.device("cuda:0")
ctx = capi.Context(0)
per = (1, 1, 0)
bc = capi.bc_from_flags(per)
P = capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=None, fused=True)
cfg = []
for bx, by, bz in shapes:
    boxes = [(i, j, k, i + bx - 1, j + by - 1, k + bz - 1) for k in range(0, n, bz) for j in range(0, n, by) for i in range(0, n, bx)]
    lv = Level(np.asarray(boxes), (0, 0, 0), (n - 1,) * 3, per, (0, 0, 0), (1, 1, 1))
    dl = capi.DevLevel(ctx, lv)
    off, cs, tot = mf_layout(lv.boxes, 1, 2)
    t = torch.zeros(tot, dtype=torch.float64, device=dev)
    bench.fill_level_on_device(torch, lv, t, 1, 2, off, cs, dev, 7)
    torch.cuda.synchronize()
    st, wk, ou = capi.DevMF(ctx, dl, 1, 2, t.data_ptr()), capi.DevMF(ctx, dl, 1, 2), capi.DevMF(ctx, dl, 8, 0)
    cfg.append(dict(shape=(bx, by, bz), nb=len(boxes), keep=(lv, dl, t), st=st, wk=wk, ou=ou, k=[], w=[]))
ROUNDS, R = 5, 4
for r in range(ROUNDS + 1):
    for c in cfg:
        run = lambda: capi.gradcurv_run(ctx, [c["st"]], 0, bc, P, [c["wk"]], [c["ou"]], 0)
        run(); ctx.sync()
        ctx.profile_enable(1 << 1)  # tag 1: the fused sweep
        ctx.profile_read(1, True)
        t0 = time.perf_counter()
        for _ in range(R):
            run()
        ctx.sync()
        w = (time.perf_counter() - t0) / R * 1e3
        nk, kms = ctx.profile_read(1, True)
        ctx.profile_enable(False)
        if r:  # round 0 warms up
            c["k"].append(kms / max(nk, 1)); c["w"].append(w)
        c["name"] = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
for c in cfg:
    km, wm = float(np.median(c["k"])), float(np.median(c["w"]))
    print("boxes %3dx%3dx%3d (%6d): pass %.3f ms (with events), sweep median %.3f ms = %.3f of HBM  rounds [%s]  %s" % (*c["shape"], c["nb"], wm, km, n ** 3 * 72 / (km * 1e-3) / 8e12,
          " ".join("%.3f" % v for v in c["k"]), c["name"][:48]), flush=True)
