"""GPU tier: device memory comes back.  A context with its levels, multifabs, cached plans, work multifabs (pa_level_scratch), compact
ghost arrays, marching-cubes blocks and solver vectors is built, driven through the pipelines of the four tools and destroyed, several
times over; the free memory of the device after the last cycle must be what it was after the second (the first two pay for the runtime's
own one-time allocations: code objects, queues, scratch)."""
import ctypes as C

import numpy as np
import pytest

from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import MultiFab, field_flame, nested_hierarchy

pytestmark = pytest.mark.gpu


def _cycle(per=(1, 1, 0), stages="abcdefgh", ctx=None):
    from util import make_states
    own_ctx = ctx is None
    if own_ctx:
        ctx = capi.Context(0)
    H = nested_hierarchy(32, 3, 16, is_per=per)
    bc = capi.bc_from_flags(per)
    st = make_states(H, 4, 2, field_flame, seed=3)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    own = []

    def mf(dl, nc, ng):
        m = capi.DevMF(ctx, dl, nc, ng)
        own.append(m)
        return m

    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, st)]
    own += dst
    works = [mf(dl, 1, 2) for dl in dls]
    outs = [mf(dl, 8, 0) for dl in dls]
    if "a" in stages:
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=300.0, prog_max=2000.0, fused=True), works, outs, 0)  # fused sweeps, compact ghost arrays, patches
    if "b" in stages:
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=300.0, prog_max=2000.0, fused=True, threshold=0.05), works, outs, 0)  # clip: slow lists
    g4 = [mf(dl, 4, 0) for dl in dls]
    if "c" in stages:
        capi.grad_run(ctx, dst, 0, bc, g4, 0)
    o18 = [mf(dl, 18, 0) for dl in dls]
    for fused in (True, False):  # options (work multifabs kept with the levels) + the smoothing solve, plain and multigrid-preconditioned
        for dt in (1e-4, 2e-2):
            if ("d" if dt < 1e-3 else "e") in stages:
                capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(fused=fused, do_smooth=True, smoothing_time=dt, do_gauss=True, do_strain=True,
                                                                     strain_tensor=True, do_velnormal=True, vel_comp=1), o18, 0)
    # filterPlt: ghost fill of the hierarchy + box filter
    ngs = [1, 2, 4]
    fin = [mf(dl, 1, ngs[l]) for l, dl in enumerate(dls)]
    fout = [mf(dl, 1, 0) for dl in dls]
    for m in fin:
        m.setval(1.0)
    hfin = (C.c_void_p * 3)(*[m.h for m in fin])
    if "f" in stages:
        ctx.check(ctx.lib.pa_fill_ghosts_hierarchy(ctx.h, 3, hfin, 0, 1, (C.c_int32 * 3)(*ngs), 2, 1, 1))
        for l, f in enumerate((2, 4, 8)):
            w = (C.c_double * (f + 2))()
            assert ctx.lib.pa_box_filter_weights(f, w) == f // 2
            ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, fin[l].h, fout[l].h, 0, 1, ngs[l], w))
    # isosurface: marching cubes of the hierarchy (pooled output block), coordinates from the cell indices
    loops = []
    for lv in H.levels:
        lp = np.zeros((lv.nboxes, 6), np.int64)
        lp[:, :3], lp[:, 3:] = lv.boxes[:, :3], lv.boxes[:, 3:] - 1
        loops.append(lp)
    for _ in range(2 if "g" in stages else 0):
        capi.mc_hierarchy(ctx, dst, [1, 1, 0], loops, 0, 1150.0)
    # streamlines through the hierarchy
    v3 = [mf(dl, 3, 2) for dl in dls]
    for m in v3:
        m.setval(0.25)
    if "h" in stages:
        capi.stream_trace(ctx, v3, 0, np.array([[0.5, 0.5, 0.5], [0.3, 0.6, 0.4]]), 8, 0.005)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for m in own:
        m.close()
    for dl in dls:
        dl.close()
    if own_ctx:
        ctx.close()


def test_device_memory_comes_back_after_contexts_are_destroyed():
    """(two warm-up cycles: the runtime keeps 356 MiB after the first context of a process and another 88 MiB after the second -- code
    objects, queues and their scratch -- and nothing after that: ten cycles of the round-5 leak hunt)"""
    torch = pytest.importorskip("torch")
    for _ in range(2):
        _cycle()
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info(0)
    for _ in range(3):
        _cycle()
    torch.cuda.synchronize()
    free5, _ = torch.cuda.mem_get_info(0)
    assert free2 - free5 <= 4 << 20, f"{(free2 - free5) / 2**20:.1f} MiB of device memory did not come back over three create / run / destroy cycles"


def test_device_memory_comes_back_when_levels_are_rebuilt_under_one_context():
    """what a regridding AMR code does: ONE long-lived context, the levels (BoxArrays) and their multifabs created and destroyed again and
    again -- plans, tables and work multifabs cached for a level must go with the level"""
    torch = pytest.importorskip("torch")
    ctx = capi.Context(0)
    for _ in range(2):
        _cycle(ctx=ctx)
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info(0)
    for _ in range(4):
        _cycle(ctx=ctx)
    torch.cuda.synchronize()
    free6, _ = torch.cuda.mem_get_info(0)
    ctx.close()
    assert free2 - free6 <= 4 << 20, f"{(free2 - free6) / 2**20:.1f} MiB of device memory did not come back over four rebuilds of the levels under one context"


def _sharded_worker(rank, world, port):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    torch.cuda.init()  # torch first: one HIP runtime per process
    from peleanalysis_amd import capi as cp
    from peleanalysis_amd import dist as padist
    from peleanalysis_amd.hierarchy import MultiFab as MF, field_flame as ff, nested_hierarchy as nh
    from test_dist_gloo import scattered_owner
    from util import make_states
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = cp.Context(0)
        comm = padist.GlooComm(ctx)
        per = (1, 1, 0)
        bc = cp.bc_from_flags(per)
        used = []
        for cycle in range(6):
            H = nh(40, 3, 20, is_per=per)
            owners = [scattered_owner(lv.nboxes, world, 31 + l + cycle) for l, lv in enumerate(H.levels)]  # another owner map every time: new plans
            gst = make_states(H, 4, 2, ff, seed=5)
            dls = [cp.DevLevel(ctx, lv, owners[l], rank, world) for l, lv in enumerate(H.levels)]
            own = []
            lst = []
            for l, dl in enumerate(dls):
                s = MF(dl.level, 4, 2, fill=0.0)
                for i, g in enumerate(dl.gids):
                    s.valid(i)[...] = gst[l].valid(int(g))
                lst.append(cp.DevMF.from_host(ctx, dl, s))
            own += lst
            work = [cp.DevMF(ctx, dl, 1, 2) for dl in dls]
            out8 = [cp.DevMF(ctx, dl, 8, 0) for dl in dls]
            out18 = [cp.DevMF(ctx, dl, 18, 0) for dl in dls]
            own += work + out8 + out18
            cp.gradcurv_run(ctx, lst, 0, bc, cp.curv_params(fused=True), work, out8, 0)
            cp.gradcurv_run(ctx, lst, 0, bc, cp.curv_params(fused=False), work, out8, 0)
            for dt in (1e-4, 2e-2):  # the distributed smoothing solve, plain and multigrid-preconditioned, feeding the options
                cp.curvature_run(ctx, lst, 0, bc, cp.curv_params(fused=True, do_smooth=True, smoothing_time=dt, do_gauss=True, do_strain=True, strain_tensor=True,
                                                                do_velnormal=True, vel_comp=1), out18, 0)
            ctx.sync()
            assert ctx.bc_errors() == 0
            for m in own:
                m.close()
            for dl in dls:
                dl.close()
            dist.barrier()
            torch.cuda.synchronize()
            dist.barrier()
            free, total = torch.cuda.mem_get_info(0)
            used.append((total - free) / 2**20)
            dist.barrier()
        if rank == 0:
            assert used[-1] - used[2] <= 8.0, f"device memory in use after each rebuild of the sharded levels (MiB): {[round(u) for u in used]}"
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_device_memory_comes_back_when_sharded_levels_are_rebuilt():
    """the same for a hierarchy sharded over two ranks that share the card (the transport's plans, coarse-source copies, restriction
    plans and flux registers of the distributed smoothing solve live with the levels): six rebuilds with a different owner map each"""
    import socket
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_worker, args=(2, port), nprocs=2, join=True)
