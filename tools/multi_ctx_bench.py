#!/usr/bin/env python3
"""Experiment: component-level concurrency.  NS library contexts (one HIP stream each) drive different components of the
same resident hierarchy at once (components are independent; each context owns its work / output multifabs, the state
multifabs alias the same memory).  usage: python tools/multi_ctx_bench.py base nlev box ncomp NS [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import mf_layout, nested_hierarchy
base, nlev, box, ncomp, NS = (int(v) for v in sys.argv[1:6])
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda", 0)
H = nested_hierarchy(base, nlev, box, is_per=(1, 1, 0))
bc = capi.bc_from_flags((1, 1, 0))
tins = []
for li, lv in enumerate(H.levels):
    off, cs, tot = mf_layout(lv.boxes, ncomp, 2)
    t = torch.zeros(tot, dtype=torch.float64, device=dev)
    bench.fill_level_on_device(torch, lv, t, ncomp, 2, off, cs, dev, 99 + li)
    tins.append(t)
torch.cuda.synchronize()
ctxs, sets, keep = [], [], []
for s in range(NS):
    ctx = capi.Context(0)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    st = [capi.DevMF(ctx, dl, ncomp, 2, t.data_ptr()) for dl, t in zip(dls, tins)]
    wk, ou = [], []
    for lv, dl in zip(H.levels, dls):
        tw = torch.zeros(mf_layout(lv.boxes, 1, 2)[2], dtype=torch.float64, device=dev)
        to = torch.zeros(mf_layout(lv.boxes, 8, 0)[2], dtype=torch.float64, device=dev)
        keep += [tw, to]
        wk.append(capi.DevMF(ctx, dl, 1, 2, tw.data_ptr()))
        ou.append(capi.DevMF(ctx, dl, 8, 0, to.data_ptr()))
    ctxs.append(ctx); sets.append((dls, st, wk, ou))
torch.cuda.synchronize()
par = capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=None, fused=True)


def step():
    for c in range(ncomp):
        s = c % NS
        capi.gradcurv_run(ctxs[s], sets[s][1], c, bc, par, sets[s][2], sets[s][3], 0)


for _ in range(2):
    step()
for c in ctxs:
    c.sync()
t0 = time.perf_counter()
for _ in range(steps):
    step()
for c in ctxs:
    c.sync()
dt = (time.perf_counter() - t0) / steps
cells = sum(lv.ncells for lv in H.levels)
print(f"NS={NS}: {dt * 1e3:.2f} ms per step, {cells * ncomp / dt / 1e9:.1f} Gcells/s")
