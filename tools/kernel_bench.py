#!/usr/bin/env python3
"""Per-kernel measurements of the non-headline kernels of the path: grad,
box filter, marching cubes, distance function, on one 512^3 level of 128^3 boxes (SURVEY 8d sizes),
timed with the library's own HIP events on its stream (pa_profile_*).  Algorithmic bytes per cell as in
SURVEY 8(d).  usage: python tools/kernel_bench.py [n=512] [box=128]   (prints one JSON object)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (torch first: one HIP runtime; used for device-side synthetic data)

from peleanalysis_amd import capi  # noqa: E402
from peleanalysis_amd.hierarchy import Level, chop_box, mf_layout  # noqa: E402

HBM = 8000.0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
box = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
ctx = capi.Context(0, stream.cuda_stream)
lv = Level(chop_box((0, 0, 0), (n - 1,) * 3, box), (0, 0, 0), (n - 1,) * 3, (1, 1, 1), (0, 0, 0), (1, 1, 1))
dl = capi.DevLevel(ctx, lv)
cells = lv.ncells
out = {"level": f"{n}^3, {lv.nboxes} boxes of {box}^3, periodic", "cells": cells, "hbm_peak_GBs": HBM, "kernels": {}}


def dev_mf(ncomp, ng, fill_random=True):
    _, _, tot = mf_layout(lv.boxes, ncomp, ng)
    with torch.cuda.stream(stream):
        t = (300.0 + 1700.0 * torch.rand(tot, dtype=torch.float64, device=dev)) if fill_random else torch.zeros(tot, dtype=torch.float64, device=dev)
    stream.synchronize()
    return t, capi.DevMF(ctx, dl, ncomp, ng, t.data_ptr())


def timed(tag, fn, reps=5):
    fn()
    ctx.sync()
    ctx.profile_read(tag, reset=True)
    ctx.profile_enable(True)
    for _ in range(reps):
        fn()
    ctx.sync()
    ctx.profile_enable(False)
    nl, ms = ctx.profile_read(tag, reset=True)
    return ms / reps


# ---- grad (grad.cpp:211-236): read phi, write gx,gy,gz,|g| = 40 B/cell
tphi, phi = dev_mf(1, 1)
tout, gout = dev_mf(4, 0, False)
ctx.check(ctx.lib.pa_fill_boundary(ctx.h, phi.h, 0, 1, 1))
ms = timed(5, lambda: ctx.check(ctx.lib.pa_grad_level(ctx.h, phi.h, 0, gout.h, 0)))
out["kernels"]["k_grad"] = {"ms": ms, "bytes_per_cell": 40, "GBs": cells * 40 / ms / 1e6, "frac_hbm": cells * 40 / ms / 1e6 / HBM, "Mcells_s": cells / ms / 1e3}
del gout, tout
if len(sys.argv) > 3 and sys.argv[3] == "gradonly":
    print(json.dumps(out))
    sys.exit(0)

MCONLY = len(sys.argv) > 3 and sys.argv[3] == "mconly"
# ---- box filter (filterPlt.cpp:217): 16 B/cell, fgr 2 / 4 / 8 -> 27 / 125 / 729 taps; the separable default and the tap-order form
FILTERONLY = len(sys.argv) > 3 and sys.argv[3] == "filteronly"
for fgr in (() if MCONLY else (2, 4, 8)):
    w = (C.c_double * (fgr + 2))()
    ng = ctx.lib.pa_box_filter_weights(fgr, w)
    tin, fin = dev_mf(1, ng)
    tfo, fo = dev_mf(1, 0, False)
    ctx.check(ctx.lib.pa_fill_boundary(ctx.h, fin.h, 0, 1, ng))
    taps = (2 * ng + 1) ** 3
    for mode in ("separable", "exact"):
        if mode == "exact":
            os.environ["PA_FILTER_EXACT"] = "1"
        ctx.lib.pa_options_reload()  # the library reads its switches once
        ms = timed(7, lambda: ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, fin.h, fo.h, 0, 1, ng, w)))
        os.environ.pop("PA_FILTER_EXACT", None)
        ctx.lib.pa_options_reload()
        key = f"k_filter_sep fgr={fgr} (3 x {2 * ng + 1} taps)" if mode == "separable" else f"k_boxfilter fgr={fgr} ({taps} taps, PA_FILTER_EXACT=1)"
        out["kernels"][key] = {"ms": ms, "bytes_per_cell": 16, "GBs": cells * 16 / ms / 1e6, "frac_hbm": cells * 16 / ms / 1e6 / HBM, "Mcells_s": cells / ms / 1e3}
    del fin, fo, tin, tfo
if FILTERONLY:
    print(json.dumps(out))
    sys.exit(0)

# ---- marching cubes on one FAB (isosurface.cpp:1566-1592): scan 8 B/cell (+ mask 8 B/cell as the reference stores it)
g = box + 2
x = (torch.arange(g, device=dev, dtype=torch.float64) - 0.5) / box
X, Y, Z = x[None, None, :].expand(g, g, g), x[None, :, None].expand(g, g, g), x[:, None, None].expand(g, g, g)
r = torch.sqrt((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2)
with torch.cuda.stream(stream):
    st = torch.stack([X, Y, Z, 300.0 + 1700.0 * 0.5 * (1 + torch.tanh((r - 0.3) / 0.05)), r]).contiguous()
    mk = torch.ones((g, g, g), dtype=torch.float64, device=dev)
stream.synchronize()
fs, fm, bx = capi.PaFab(), capi.PaFab(), capi.PaBox()
fs.p, fs.ncomp, fs.nstride = st.data_ptr(), 5, 0
fm.p, fm.ncomp, fm.nstride = mk.data_ptr(), 1, 0
for d in range(3):
    fs.lo[d] = fm.lo[d] = -1
    fs.hi[d] = fm.hi[d] = box
    bx.lo[d], bx.hi[d] = -1, box - 1
nv, nt = C.c_int64(0), C.c_int64(0)
ms_count = timed(8, lambda: ctx.check(ctx.lib.pa_mc_count_fab(ctx.h, bx, fs, fm, 3, 1150.0, C.byref(nv), C.byref(nt))))
tv = torch.empty(max(nv.value, 1) * 5, dtype=torch.float64, device=dev)
tk = torch.empty(max(nv.value, 1) * 6, dtype=torch.int32, device=dev)
tt = torch.empty(max(nt.value, 1) * 3, dtype=torch.int32, device=dev)
ms_emit = timed(8, lambda: ctx.check(ctx.lib.pa_mc_emit_fab(ctx.h, bx, fs, fm, 3, 1150.0, tv.data_ptr(), tk.data_ptr(), tt.data_ptr(), nv.value, nt.value)))
fc = g ** 3
out["kernels"]["k_mc count (classify+count+scan, 1 FAB)"] = {"ms": ms_count, "cells": fc, "Mcells_s": fc / ms_count / 1e3, "bytes_per_cell": 16,
                                                            "GBs": fc * 16 / ms_count / 1e6, "note": "includes the synchronous 16-byte D2H of the totals"}
out["kernels"]["k_mc emit (count again + vertices + triangles, 1 FAB)"] = {"ms": ms_emit, "vertices": nv.value, "triangles": nt.value,
                                                                         "Mtriangles_s": nt.value / ms_emit / 1e3, "Mcells_s": fc / ms_emit / 1e3}

# ---- marching cubes, level-batched (pa_iso_mask_level + pa_mc_level): every FAB of the level in one pass
off5, cs5, tot5 = mf_layout(lv.boxes, 5, 1)
with torch.cuda.stream(stream):
    t5 = torch.empty(tot5, dtype=torch.float64, device=dev)
    for b in range(lv.nboxes):
        lo = lv.boxes[b, :3] - 1
        xs = [(torch.arange(int(lo[d]), int(lo[d]) + g, device=dev, dtype=torch.float64) + 0.5) / n for d in range(3)]
        Xb, Yb, Zb = xs[0][None, None, :].expand(g, g, g), xs[1][None, :, None].expand(g, g, g), xs[2][:, None, None].expand(g, g, g)
        rb = torch.sqrt((Xb - 0.5) ** 2 + (Yb - 0.5) ** 2 + (Zb - 0.5) ** 2)
        for c, v in enumerate((Xb, Yb, Zb, 300.0 + 1700.0 * 0.5 * (1 + torch.tanh((rb - 0.3) / 0.05)), rb)):
            t5[off5[b] + c * cs5[b]: off5[b] + c * cs5[b] + g ** 3] = v.reshape(-1)
stream.synchronize()
st5 = capi.DevMF(ctx, dl, 5, 1, t5.data_ptr())
tmk, mk5 = dev_mf(1, 1, False)
ctx.check(ctx.lib.pa_iso_mask_level(ctx.h, mk5.h, 0, None, 2))
loops = (capi.PaBox * lv.nboxes)()
for b in range(lv.nboxes):
    for d in range(3):
        loops[b].lo[d] = max(int(lv.boxes[b, d]) - 1, 0)
        loops[b].hi[d] = min(int(lv.boxes[b, 3 + d]) + 1, n - 1) - 1
nvb, ntb = (C.c_int64 * lv.nboxes)(), (C.c_int64 * lv.nboxes)()
pv, pk, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()


def mc_level_once():
    ctx.check(ctx.lib.pa_mc_level(ctx.h, st5.h, mk5.h, 0, loops, 3, 1150.0, nvb, ntb, C.byref(pv), C.byref(pk), C.byref(pt)))
    ctx.lib.pa_device_free(ctx.h, pv)  # one allocation


mc_level_once()
ctx.sync()
t0 = time.perf_counter()
for _ in range(3):
    mc_level_once()
ctx.sync()
ms_lvl = (time.perf_counter() - t0) * 1e3 / 3
lc = lv.nboxes * g ** 3
out["kernels"][f"pa_mc_level ({lv.nboxes} FABs of {g}^3 in one pass: flags, classify, count, scan, vertices, triangles; wall clock incl. allocation)"] = {
    "ms": ms_lvl, "cells": lc, "Mcells_s": lc / ms_lvl / 1e3, "bytes_per_cell": 16, "GBs": lc * 16 / ms_lvl / 1e6, "frac_hbm": lc * 16 / ms_lvl / 1e6 / HBM,
    "triangles": int(sum(ntb)), "Mtriangles_s": sum(ntb) / ms_lvl / 1e3}
ms_mask = timed(8, lambda: ctx.check(ctx.lib.pa_iso_mask_level(ctx.h, mk5.h, 0, None, 2)))


# same pass with the mask evaluated from a "finer level" (here: the level itself as a stand-in owner map; nothing read back)
mc_level_once()
t0 = time.perf_counter()
for _ in range(3):
    ctx.check(ctx.lib.pa_mc_level_fine(ctx.h, st5.h, None, 2, loops, 3, 1150.0, nvb, ntb, C.byref(pv), C.byref(pk), C.byref(pt)))
    ctx.lib.pa_device_free(ctx.h, pv)
ctx.sync()
ms_fine = (time.perf_counter() - t0) * 1e3 / 3
out["kernels"]["pa_mc_level_fine (same level, mask evaluated in the cell pass: 8 B/cell read)"] = {
    "ms": ms_fine, "cells": lc, "Mcells_s": lc / ms_fine / 1e3, "bytes_per_cell": 8, "GBs": lc * 8 / ms_fine / 1e6, "frac_hbm": lc * 8 / ms_fine / 1e6 / HBM,
    "triangles": int(sum(ntb)), "Mtriangles_s": sum(ntb) / ms_fine / 1e3}
del st5, mk5, t5, tmk
if MCONLY and not (len(sys.argv) > 4 and sys.argv[4] == "sdf"):
    print(json.dumps(out))
    sys.exit(0)

# ---- distance function (make_level_set3) on that FAB's mesh: one grid, and a batch of 64 grids
xf = tv.view(-1, 5)[:, :3].to(torch.float32).contiguous()
phis = [torch.empty(g ** 3, dtype=torch.float32, device=dev) for _ in range(64)]
grids = (capi.PaSdfGrid * 64)()
for q, gr in enumerate(grids):
    gr.ntri, gr.tri, gr.nvert, gr.x = nt.value, tt.data_ptr(), nv.value, xf.data_ptr()
    for d in range(3):
        gr.origin[d] = np.float32(-1.0 / box)
        gr.n[d] = g
    gr.dx = np.float32(1.0 / box)
    gr.phi = phis[q].data_ptr()
for nb in (1, 64):
    ctx.check(ctx.lib.pa_sdf_level_set3(ctx.h, nb, grids, 1))
    ctx.sync()
    t0 = time.perf_counter()
    ctx.check(ctx.lib.pa_sdf_level_set3(ctx.h, nb, grids, 1))
    ctx.sync()
    dt = (time.perf_counter() - t0) * 1e3
    out["kernels"][f"sdf make_level_set3, {nb} grid(s) of {g}^3, {nt.value} triangles each"] = {"ms": dt, "Mpoints_s": nb * g ** 3 / dt / 1e3}
print(json.dumps(out) if MCONLY else json.dumps(out, indent=1))
