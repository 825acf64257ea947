#!/usr/bin/env python3
"""pa_iso_merge timing on synthetic fragments (device pointers prepared outside the timed call).
usage: python tools/merge_bench.py [nfrag=192] [verts_per_frag=1500]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peleanalysis_amd import capi  # noqa: E402

nfrag = int(sys.argv[1]) if len(sys.argv) > 1 else 192
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
nc = 5
rng = np.random.default_rng(1)
ctx = capi.Context(0)
pool = rng.random((nfrag * nv, 3))
bufs, arr = [], (capi.PaIsoFrag * nfrag)()
tot_v = tot_t = 0
for f in range(nfrag):
    own = pool[f * nv:(f + 1) * nv]
    shared = pool[rng.integers(0, max(f, 1) * nv, nv // 8)] if f else own[:0]  # copies of earlier fragments' vertices
    v = np.concatenate([np.concatenate([own, shared]), rng.random((len(own) + len(shared), nc - 3))], axis=1)
    t = rng.integers(0, len(v), (2 * len(v), 3)).astype(np.int32)
    bv, bt = capi.DevBuf.from_numpy(ctx, v), capi.DevBuf.from_numpy(ctx, t)
    bufs += [bv, bt]
    arr[f].verts, arr[f].nvert, arr[f].tris, arr[f].ntri = bv.ptr, len(v), bt.ptr, len(t)
    tot_v += len(v); tot_t += len(t)
for rep in range(4):
    nn, ne, pn, pe = C.c_int64(0), C.c_int64(0), C.c_void_p(), C.c_void_p()
    t0 = time.perf_counter()
    rc = ctx.lib.pa_iso_merge(ctx.h, nfrag, arr, nc, C.byref(nn), C.byref(pn), C.byref(ne), C.byref(pe))
    dt = time.perf_counter() - t0
    assert rc == 0, rc
    ctx.lib.pa_device_free(ctx.h, pn); ctx.lib.pa_device_free(ctx.h, pe)
    print(f"call {rep}: {dt * 1e3:.2f} ms  ({tot_v} raw vertices -> {nn.value} nodes, {tot_t} raw elements -> {ne.value})", flush=True)
