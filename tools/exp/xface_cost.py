#!/usr/bin/env python3
"""What do the coarse-fine X-faces cost in the boundary kernels?  Two 3-level hierarchies under rocprofv3 --kernel-trace:
 "nested": bench.py's headline (levels 1-2 = the central half cube: x-, y- and z-faces are coarse-fine)
 "slab":   levels 1-2 refine a slab that spans the (periodic) domain in x: coarse-fine faces in y and z only
usage: xface_cost.py nested|slab [base=512]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import Hierarchy, Level, MultiFab, chop_box, nested_hierarchy, retile_hierarchy
kind = sys.argv[1]
base = int(sys.argv[2]) if len(sys.argv) > 2 else 512
per = (1, 1, 0)
if kind == "nested":
    H = nested_hierarchy(base, 3, 128, is_per=per)
else:
    levels = []
    lo, hi = np.zeros(3, np.int64), np.full(3, base - 1, np.int64)
    dlo, dhi = np.zeros(3, np.int64), np.full(3, base - 1, np.int64)
    for l in range(3):
        levels.append(Level(chop_box(lo, hi, 128), dlo.copy(), dhi.copy(), np.asarray(per), np.zeros(3), np.ones(3)))
        n = hi - lo + 1
        clo, chi = lo + n // 4, lo + n // 4 + n // 2 - 1
        clo[0], chi[0] = dlo[0], dhi[0]  # the whole (periodic) x extent
        lo, hi = 2 * clo, 2 * chi + 1
        dlo, dhi = 2 * dlo, 2 * dhi + 1
    H = Hierarchy(levels, 2)
H = retile_hierarchy(H)
print(kind, "cells per level", [lv.ncells for lv in H.levels], "boxes", [lv.nboxes for lv in H.levels], "box 0 of level 1", H.levels[1].boxes[0].tolist(), flush=True)
bc = capi.bc_from_flags(per)
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
params = capi.curv_params(prog_min=300.0, prog_max=2000.0, fused=True)
for _ in range(4):
    capi.gradcurv_run(ctx, states, 0, bc, params, works, outs, 0)
ctx.sync()
