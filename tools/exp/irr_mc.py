#!/usr/bin/env python3
"""Experiment: the isosurface path (state build + pa_mc_hierarchy_fine) on the bench's IRREGULAR hierarchy against a nested one
of 128^3 boxes, per cell -- do the marching-cubes / FillPatch launches lose on boxes of different sizes like the sweeps did?"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import mf_layout, nested_hierarchy, tagged_hierarchy, field_flame
import bench

dev = torch.device("cuda:0")
ctx = capi.Context(0)


def alloc(lv, dl, ncomp, ng, fill=False, seed=1):
    off, cs, tot = mf_layout(lv.boxes, ncomp, ng)
    t = torch.zeros(max(tot, 1), dtype=torch.float64, device=dev)
    if fill:
        bench.fill_level_on_device(torch, lv, t, 1, ng, off, cs, dev, seed)
    torch.cuda.synchronize()
    return t, capi.DevMF(ctx, dl, ncomp, ng, t.data_ptr())


def timed(fn, reps=3):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for name, H in (("nested 512^3 base, 128^3 boxes", nested_hierarchy(512, 3, 128, is_per=(0, 0, 0))),
                ("irregular (bench's tagged hierarchy)", tagged_hierarchy(512, 3, lambda x, y, z: field_flame(x, y, z, 0), bf=16, max_box=128, base_box=128, frac=(0.08, 0.16), is_per=(0, 0, 0)))):
    nl = H.nlev
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    fld = [alloc(lv, dl, 1, 1, True, 91 + l) for l, (lv, dl) in enumerate(zip(H.levels, dls))]
    sts = [alloc(lv, dl, 4, 1) for lv, dl in zip(H.levels, dls)]
    loops = []
    for lv in H.levels:
        arr = (capi.PaBox * lv.nboxes)()
        for b in range(lv.nboxes):
            for d in range(3):
                arr[b].lo[d] = max(int(lv.boxes[b, d]) - 1, int(lv.domlo[d]))
                arr[b].hi[d] = min(int(lv.boxes[b, 3 + d]) + 1, int(lv.domhi[d])) - 1
        loops.append(arr)

    def iso_state():
        for l in range(nl):
            ctx.check(ctx.lib.pa_iso_coords_level(ctx.h, sts[l][1].h, 0))
            ctx.check(ctx.lib.pa_mf_copy(ctx.h, fld[l][1].h, 0, sts[l][1].h, 3, 1, 1))
            ctx.check(ctx.lib.pa_fill_boundary(ctx.h, sts[l][1].h, 0, 4, 1))
            if l > 0:
                ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, sts[l][1].h, sts[l - 1][1].h, 0, 4, 1, 2, 0))

    nvs = [(C.c_int64 * lv.nboxes)() for lv in H.levels]
    nts = [(C.c_int64 * lv.nboxes)() for lv in H.levels]
    parr = (C.POINTER(capi.PaBox) * nl)(*[C.cast(a, C.POINTER(capi.PaBox)) for a in loops])
    pnv = (C.POINTER(C.c_int64) * nl)(*[C.cast(a, C.POINTER(C.c_int64)) for a in nvs])
    pnt = (C.POINTER(C.c_int64) * nl)(*[C.cast(a, C.POINTER(C.c_int64)) for a in nts])
    fm = (C.c_int32 * nl)(*([1] * (nl - 1) + [0]))
    hst = (C.c_void_p * nl)(*[s_[1].h for s_ in sts])
    tri = [0]

    def iso_mc():
        pv, pk, pt = (C.c_void_p * nl)(), (C.c_void_p * nl)(), (C.c_void_p * nl)()
        block = C.c_void_p()
        ctx.check(ctx.lib.pa_mc_hierarchy_fine(ctx.h, nl, hst, fm, 2, parr, 3, 1150.0, pnv, pnt, pv, pk, pt, C.byref(block)))
        tri[0] = sum(int(sum(nts[l][:H.levels[l].nboxes])) for l in range(nl))
        if block.value:
            ctx.lib.pa_device_free(ctx.h, block)

    iso_state()
    cells = sum(lv.ncells for lv in H.levels)
    ms_state, ms_mc = timed(iso_state), timed(iso_mc)
    print("%-40s cells %.3e boxes %s: state build %.3f ms (%.2f ns/cell), marching cubes %.3f ms (%.3f ns/cell = %.2f of HBM at 8 B/cell), %d triangles" %
          (name, cells, [lv.nboxes for lv in H.levels], ms_state, ms_state * 1e6 / cells, ms_mc, ms_mc * 1e6 / cells, cells * 8 / (ms_mc * 1e-3) / 8e12, tri[0]), flush=True)
    del sts, fld, dls
    torch.cuda.empty_cache()
