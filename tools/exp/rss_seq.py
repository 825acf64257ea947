#!/usr/bin/env python3
"""host memory of the process across many calls on the same levels: does anything grow per call?"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, psutil, torch
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
from util import make_states
P = psutil.Process()
ctx = capi.Context(0)
H = nested_hierarchy(32, 3, 16, is_per=(1, 1, 0))
bc = capi.bc_from_flags((1, 1, 0))
st = make_states(H, 4, 2, field_flame, seed=3)
dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, st)]
works = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
outs = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
o18 = [capi.DevMF(ctx, dl, 18, 0) for dl in dls]
g4 = [capi.DevMF(ctx, dl, 4, 0) for dl in dls]
loops = []
for lv in H.levels:
    lp = np.zeros((lv.nboxes, 6), np.int64); lp[:, :3], lp[:, 3:] = lv.boxes[:, :3], lv.boxes[:, 3:] - 1; loops.append(lp)
def block(n):
    for _ in range(n):
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=300.0, prog_max=2000.0, fused=True), works, outs, 0)
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=300.0, prog_max=2000.0, fused=True, threshold=0.05), works, outs, 0)
        capi.grad_run(ctx, dst, 0, bc, g4, 0)
        capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(fused=True, do_smooth=True, smoothing_time=2e-2, do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True, vel_comp=1), o18, 0)
        capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(fused=False, do_gauss=True, do_strain=True, do_velnormal=True, vel_comp=1), o18, 0)
        capi.mc_hierarchy(ctx, dst, [1, 1, 0], loops, 0, 1150.0)
    ctx.sync()
seq = []
for b in range(8):
    block(25)
    seq.append(P.memory_info().rss / 2**20)
print("RSS (MiB) after each block of 25 rounds of calls:", ["%.1f" % v for v in seq])
