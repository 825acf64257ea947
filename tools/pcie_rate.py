#!/usr/bin/env python3
"""PCIe-inclusive rate of the fused grad->curvature path on one 512^3 level of 128^3 boxes: pageable
host arrays -> pa_mf_upload -> ghost fills + sweep (+ face fix-up: none on a periodic single level) -> pa_mf_download of
the 8 output components.  Never the bench value; it prices the host link, not the kernels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import Level, MultiFab, chop_box
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = capi.Context(0)
lv = Level(chop_box((0, 0, 0), (n - 1,) * 3, 128), (0, 0, 0), (n - 1,) * 3, (1, 1, 1), (0, 0, 0), (1, 1, 1))
dl = capi.DevLevel(ctx, lv)
hin = MultiFab(lv, 1, 2)
hin.data[:] = 300.0 + 1700.0 * np.random.default_rng(1).random(hin.total)
hout = MultiFab(lv, 8, 0)
dst, dwk, dout = capi.DevMF(ctx, dl, 1, 2), capi.DevMF(ctx, dl, 1, 2), capi.DevMF(ctx, dl, 8, 0)
bc = capi.bc_from_flags((1, 1, 1))
par = capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=None, fused=True)
import ctypes as C
for rep in range(3):
    t0 = time.perf_counter()
    ctx.check(ctx.lib.pa_mf_upload(ctx.h, dst.h, hin.data.ctypes.data_as(C.POINTER(C.c_double))))
    t1 = time.perf_counter()
    capi.gradcurv_run(ctx, [dst], 0, bc, par, [dwk], [dout], 0)
    ctx.sync()
    t2 = time.perf_counter()
    ctx.check(ctx.lib.pa_mf_download(ctx.h, dout.h, hout.data.ctypes.data_as(C.POINTER(C.c_double))))
    t3 = time.perf_counter()
    print(f"rep {rep}: upload {hin.data.nbytes / 1e9:.2f} GB {t1 - t0:.3f} s ({hin.data.nbytes / 1e9 / (t1 - t0):.1f} GB/s), compute {1e3 * (t2 - t1):.2f} ms, "
          f"download {hout.data.nbytes / 1e9:.2f} GB {t3 - t2:.3f} s ({hout.data.nbytes / 1e9 / (t3 - t2):.1f} GB/s); "
          f"PCIe-inclusive {lv.ncells / (t3 - t0) / 1e6:.0f} Mcells/s", flush=True)
