#!/usr/bin/env python3
"""FillBoundary of the headline hierarchy level by level and with per-direction periodicity switched off: where do its ~200 us go?
usage: fb_time.py [base=512] [box=128]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import nested_hierarchy, retile_hierarchy
base = int(sys.argv[1]) if len(sys.argv) > 1 else 512
box = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = capi.Context(0)
def timed(fn, reps=50):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e6
for per in ((1, 1, 0), (0, 1, 0), (1, 0, 0), (0, 0, 0)):
    H = retile_hierarchy(nested_hierarchy(base, 3, box, is_per=per))
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    mfs = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    for m in mfs: m.setval(1.0)
    row = []
    for l, m in enumerate(mfs):
        row.append(timed(lambda m=m: ctx.check(ctx.lib.pa_fill_boundary(ctx.h, m.h, 0, 1, 2))))
    print("is_per", per, "boxes", [lv.nboxes for lv in H.levels], "box0", H.levels[0].boxes[0].tolist(), "us per level:", ["%.1f" % v for v in row], flush=True)
    del mfs, dls
