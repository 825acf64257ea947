"""GPU tier, N > 1 path on ONE device: 2 and 4 processes share cuda:0; ONE 3-level hierarchy is sharded over them with a
SCATTERED owner map (neighbouring boxes, and the coarse parents of most fine boxes, live on other ranks), so every ghost
fill crosses ranks: same-level ghost cells, the coarse data under coarse-fine faces (phi, the flame normal, the Hessian
rows, the velocity) and the progress-range reduction.  Transport: the library's pa_comm callbacks over gloo (RCCL needs
one GPU per rank); the pack / unpack kernels, region plans and pipelines are the ones an 8-GPU run uses.  The union of
the ranks' results must equal the oracle on the UNDISTRIBUTED hierarchy bit for bit -- fused and pass by pass."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(b).view(np.int64))


def _worker(rank, world, port, case, transport="gloo"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    torch.cuda.init()  # torch first: one HIP runtime per process
    from oracle import oracle as O
    from peleanalysis_amd import capi
    from peleanalysis_amd import dist as padist
    from peleanalysis_amd.hierarchy import MultiFab, cell_centers, nested_hierarchy
    from test_dist_gloo import scattered_owner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        per, sym = ((1, 1, 0), (0, 0, 0)) if case != "sym" else ((0, 1, 1), (1, 0, 0))
        if case.startswith("rand"):  # randomly drawn hierarchies (tests/test_gpu_random.py): odd extents, uneven chops, levels with
            import test_gpu_random as R  # fewer boxes than ranks (some ranks own nothing there), every boundary-condition mix
            draw = {"w": R._draw_wide, "u": R._union_case, "U": R._union_wide_case}.get(case[4])  # u / U: general BoxArrays (unions of rectangles)
            H, per, sym, _fn = draw(int(case[5:])) if draw else R._draw(int(case[4:]))
        else:
            H = nested_hierarchy(16, 3, 8, is_per=per) if case != "wide" else nested_hierarchy(80, 3, 40, is_per=per)  # wide: the exact-normal pipeline
        owners = [scattered_owner(lv.nboxes, world, 31 + l) if case != "sfc" else padist.distribution_map(lv.boxes, world) for l, lv in enumerate(H.levels)]
        rng = np.random.default_rng(99)
        ncomp = 4
        f = lambda x, y, z, c: (1 + 0.2 * c) * (1000.0 + 500.0 * np.tanh((np.sqrt((x - 0.5) ** 2 + (y - 0.45) ** 2 + (z - 0.5) ** 2) - 0.3) / 0.1)) \
            + 20.0 * np.sin(2 * np.pi * (x + 0.1 * c)) * np.cos(2 * np.pi * y) + 0 * x * y * z
        bc = capi.bc_from_flags(per, sym)
        gst = []
        for lv in H.levels:
            s = MultiFab(lv, ncomp, 2)
            for b in range(lv.nboxes):
                for c in range(ncomp):
                    s.valid(b)[c] = f(*cell_centers(lv, b, 0), c) + 1e-3 * rng.uniform(-1, 1, size=s.valid(b)[c].shape)
            gst.append(s)
        thr = 0.03 if case == "thr" else None
        og = [MultiFab(lv, 4, 0) for lv in H.levels]
        oc = [MultiFab(lv, 17, 0) for lv in H.levels]
        O.grad_pipeline(H.levels, [s.copy() for s in gst], 0, bc, og, 0)
        pm = O.curvature_pipeline(H.levels, [s.copy() for s in gst], 0, bc, oc, 0, MultiFab, threshold=thr, do_gauss=True, do_strain=True, strain_tensor=True,
                                  do_velnormal=True, vel_comp=1)

        if transport == "rccl":  # one GPU per rank, the library's own communicator: grouped ncclSend / ncclRecv over xGMI
            torch.cuda.set_device(rank)
            ctx = capi.Context(rank)
            padist.init_rccl(ctx)
            comm = None
        else:
            ctx = capi.Context(0)
            comm = padist.GlooComm(ctx)
        ctx.comm_selftest(1000)  # the transport itself: ring exchange + reduction with known answers
        dls = [capi.DevLevel(ctx, lv, owners[l], rank, world) for l, lv in enumerate(H.levels)]
        lst = []
        for l, dl in enumerate(dls):
            s = MultiFab(dl.level, ncomp, 2, fill=np.nan)  # ghost cells poisoned: every one the kernels read must have been filled
            for i, g in enumerate(dl.gids):
                s.valid(i)[...] = gst[l].valid(int(g))
            lst.append(capi.DevMF.from_host(ctx, dl, s))
        work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]

        def check(out, pairs, what):
            for l, dl in enumerate(dls):
                got = out[l].download()
                for i, g in enumerate(dl.gids):
                    for gc, (ref, rc) in pairs.items():
                        assert _same(got.valid(i)[gc], ref[l].valid(int(g))[rc]), f"rank {rank}/{world} {what}: level {l} box {g} comp {gc} differs from the undistributed oracle"

        # fused and pass-by-pass grad -> curvature; the progress range comes from the min / max reduction over the ranks
        for fused in (True, False):
            out = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
            n0 = comm.nexchange if comm else 0
            capi.gradcurv_run(ctx, lst, 0, bc, capi.curv_params(threshold=thr, fused=fused), work, out, 0)
            ctx.sync()
            assert ctx.bc_errors() == 0
            if comm and fused and not case.startswith("rand"):  # exchange A for every level at once; B (the coarse normals) for every level at once
                assert comm.nexchange - n0 == 2, "the fused pipeline batches its cross-rank traffic"
            check(out, {0: (og, 0), 1: (og, 1), 2: (og, 2), 3: (og, 3), 4: (oc, 2), 5: (oc, 3), 6: (oc, 4), 7: (oc, 1)}, f"gradcurv fused={fused}")
        # PA_DIST_EARLY=1 (read per pass): the tiles whose input the local FillBoundary completes are swept on the side stream under
        # exchange A and the ghost preparation, the others after it -- the same tiles, so the same bits
        os.environ["PA_DIST_EARLY"] = "1"
        capi.reload_options()
        out = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
        capi.gradcurv_run(ctx, lst, 0, bc, capi.curv_params(threshold=thr, fused=True), work, out, 0)
        ctx.sync()
        del os.environ["PA_DIST_EARLY"]
        capi.reload_options()
        assert ctx.bc_errors() == 0
        check(out, {0: (og, 0), 1: (og, 1), 2: (og, 2), 3: (og, 3), 4: (oc, 2), 5: (oc, 3), 6: (oc, 4), 7: (oc, 1)}, "gradcurv fused, early tiles")
        # several components at once: exchange A carries all of them, one exchange (B) per component
        if case.startswith("rand"):
            H.levels  # the option / multi-component runs below assume the fixed hierarchies' component layout: skip for random draws
            dist.barrier()
            ctx.close()
            return
        out = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
        seen = {}

        def done(c):
            ctx.sync()
            seen[c] = [o.download() for o in out]
        n0 = comm.nexchange if comm else 0
        capi.gradcurv_run_comps(ctx, lst, 0, 2, bc, capi.curv_params(prog_min=pm[0], prog_max=pm[1], threshold=thr, fused=True), work, out, 0, done)
        ctx.sync()
        assert sorted(seen) == [0, 1]
        if comm and case == "wide" and thr is None:
            assert comm.nexchange - n0 == 3, "exact-normal pipeline, 2 components: one exchange A for both + one exchange B per component"
        for l, dl in enumerate(dls):
            for i, g in enumerate(dl.gids):
                v = seen[0][l].valid(i)
                assert _same(v[0:4], og[l].valid(int(g))) and _same(v[4:7], oc[l].valid(int(g))[2:5]) and _same(v[7], oc[l].valid(int(g))[1]), \
                    f"rank {rank}/{world} run_comps: level {l} box {g} differs from the undistributed oracle"
        # the same two components as ONE batch (pa_gradcurv_run_comps2): exchange B carries the normals of both slots
        out2 = [capi.DevMF(ctx, dl, 16, 0) for dl in dls]
        seen2 = {}

        def done2(c, oc):
            ctx.sync()
            seen2[c] = (oc, [o.download() for o in out2])
        n0 = comm.nexchange if comm else 0
        capi.gradcurv_run_comps2(ctx, lst, 0, 2, bc, capi.curv_params(prog_min=pm[0], prog_max=pm[1], threshold=thr, fused=True), work, out2, 0, 2, done2)
        ctx.sync()
        assert sorted(seen2) == [0, 1]
        if comm and case == "wide" and thr is None:
            assert comm.nexchange - n0 == 2, "one batch of 2 components: one exchange A + ONE exchange B"
        for c in (0, 1):
            oc_, mfs = seen2[c]
            assert oc_ == 8 * c
            for l, dl in enumerate(dls):
                for i in range(len(dl.gids)):
                    assert _same(mfs[l].valid(i)[oc_:oc_ + 8], seen[c][l].valid(i)[0:8]), f"rank {rank}/{world} run_comps2: component {c} level {l} box {i}"
        # the curvature tool's pipeline with every option (Hessian rows and velocity need their own coarse data)
        # (fused=False: pass by pass; fused=True: the sharded exact-normal pipeline's G-output sweeps + one options pass per level where
        # every rank's share takes it -- the ranks agree on the path through one reduction)
        for fused in (False, True):
            out = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
            capi.curvature_run(ctx, lst, 0, bc, capi.curv_params(threshold=thr, fused=fused, do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True,
                                                                vel_comp=1), out, 0)
            ctx.sync()
            assert ctx.bc_errors() == 0
            check(out, {c: (oc, c) for c in range(17)}, f"curvature_run with options, fused {fused}")
            path = ctx.lib.pa_curvature_last_path(ctx.h)
            assert path == 0 if not fused else path in (0, 1)
            if fused and case in ("wide", "thr", "sym", "sfc", "scatter"):  # nested hierarchies: every rank's share takes the all-levels sweeps
                assert path == 1, f"rank {rank}/{world} case {case}: the sharded options pass fell back to the pass-by-pass kernels"
        # the gradient tool's pipeline
        out = [capi.DevMF(ctx, dl, 4, 0) for dl in dls]
        capi.grad_run(ctx, lst, 0, bc, out, 0)
        ctx.sync()
        check(out, {c: (og, c) for c in range(4)}, "grad_run")
        assert comm is None or comm.bytes_sent > 0
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


_HUNT = int(os.environ.get("PA_DIST_RANDOM_SEEDS", "0"))  # PA_DIST_RANDOM_SEEDS=30: that many more random draws (a longer hunt)


@pytest.mark.parametrize("world,case", [(2, "scatter"), (4, "scatter"), (4, "thr"), (2, "sym"), (4, "sfc"), (4, "wide"), (2, "wide"), (3, "rand2"), (4, "rand5"),
                                        (3, "rand11"), (3, "randw1"), (4, "randw4"), (3, "randu0"), (4, "randu13"), (2, "randu21"), (3, "randU0"), (4, "randU4")] + [(2 + s % 3, ("randw" if s % 4 == 3 else "rand") + str(20 + s)) for s in range(_HUNT)])
def test_sharded_hierarchy_on_shared_gpu_matches_undistributed_oracle(world, case):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), case), nprocs=world, join=True)


def _smooth_worker(rank, world, port, case, transport="gloo"):
    """do_smooth (curvature.cpp:328-406) on a sharded hierarchy: the DISTRIBUTED composite solve against the undistributed oracle"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    replicated = case.endswith("+rep")
    case = case.replace("+rep", "")
    if replicated:
        os.environ["PA_SMOOTH_REPLICATED"] = "1"  # read once, when the library first solves
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    from oracle import oracle as O
    from peleanalysis_amd import capi
    from peleanalysis_amd import dist as padist
    from peleanalysis_amd.hierarchy import MultiFab, cell_centers, nested_hierarchy
    from test_dist_gloo import scattered_owner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        per = (1, 1, 0)
        if case.startswith("rand"):
            import test_gpu_random as R
            draw = {"u": R._union_case, "U": R._union_wide_case}.get(case[4])
            H, per, _sym, _fn = draw(int(case[5:])) if draw else R._draw(int(case[4:]))
        elif case.startswith("walls"):
            per = (0, 0, 0)
            H = nested_hierarchy(16, 3, 8, is_per=per)
        elif case.startswith("2d"):  # the AMREX_SPACEDIM == 2 build: one plane of cells per level, refined in x and y only
            from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
            per = (1, 0, 0)
            l0 = Level(chop_box((0, 0, 0), (31, 31, 0), 8), (0, 0, 0), (31, 31, 0), np.array(per), np.zeros(3), np.ones(3))
            l1 = Level(chop_box((16, 16, 0), (47, 47, 0), 8), (0, 0, 0), (63, 63, 0), np.array(per), np.zeros(3), np.ones(3))
            H = Hierarchy([l0, l1], 2)
        else:
            H = nested_hierarchy(16, 3, 8, is_per=per)
        for lv in H.levels[1:]:  # the composite operator needs fine boxes aligned to the ratio (pa_smooth_solve checks it)
            for b in lv.boxes:
                assert not any(b[d] % 2 or (b[3 + d] + 1) % 2 for d in range(2 if case.startswith("2d") else 3)), "pick a draw whose fine boxes are aligned to the ratio"
        owners = [scattered_owner(lv.nboxes, world, 57 + l) for l, lv in enumerate(H.levels)]
        bc = capi.bc_from_flags(per)
        rng = np.random.default_rng(5)
        rhs = []
        for lv in H.levels:
            m = MultiFab(lv, 1, 0)
            for b in range(lv.nboxes):
                x, y, z = cell_centers(lv, b, 0)
                m.valid(b)[0] = 0.5 + 0.5 * np.tanh((np.sqrt((x - 0.5) ** 2 + (y - 0.45) ** 2 + (z - 0.5) ** 2) - 0.3) / 0.08) + 1e-2 * rng.uniform(-1, 1, size=m.valid(b)[0].shape)
            rhs.append(m)
        dt = 5e-4
        stiff = case.endswith("+mg")  # dt / dx^2 = 100 on the finest level: the multigrid-preconditioned solve (restriction, prolongation and ghost fills across ranks)
        if stiff:
            dt = 100.0 / float(H.levels[-1].domhi[0] + 1) ** 2
            want, oit, ores = O.smooth_solve(H.levels, rhs, 0, dt, bc, MultiFab, tol=1e-13, maxiter=3000)
            assert ores <= 1e-13
        else:
            want, oit, ores = O.smooth_solve(H.levels, rhs, 0, dt, bc, MultiFab, tol=1e-14)

        if transport == "rccl":
            torch.cuda.set_device(rank)
            ctx = capi.Context(rank)
            padist.init_rccl(ctx)
            comm = None
        else:
            ctx = capi.Context(0)
            comm = padist.GlooComm(ctx)
        dls = [capi.DevLevel(ctx, lv, owners[l], rank, world) for l, lv in enumerate(H.levels)]
        drhs, dsol = [], []
        for l, dl in enumerate(dls):
            s = MultiFab(dl.level, 1, 0)
            for i, g in enumerate(dl.gids):
                s.valid(i)[...] = rhs[l].valid(int(g))
            drhs.append(capi.DevMF.from_host(ctx, dl, s))
            dsol.append(capi.DevMF(ctx, dl, 1, 0))
        n0 = comm.nexchange if comm else 0
        it, res = capi.smooth_solve(ctx, drhs, 0, dsol, 0, dt, bc, tol=1e-13 if stiff else 1e-14, maxiter=200)
        if stiff:
            assert 0 < it <= 40 and res <= 1e-13 and it * 3 <= oit, (it, oit, res)  # the unpreconditioned oracle needs several times as many
        else:
            # at 1e-14 the last steps sit on the rounding floor: the count wanders with the summation order; where dt / dx^2 > 8 on the finest
            # level (the wide random draws) the library preconditions with a V-cycle and needs FEWER iterations than the plain oracle
            qf = dt * float(H.levels[-1].domhi[0] + 1) ** 2
            assert 0 < it < 100 and res <= 1e-14 and (it <= oit if qf > 8.0 else abs(it - oit) <= 12), (it, oit, res, qf)
        if comm:
            nx = comm.nexchange - n0
            if replicated:
                assert nx == 1, "replicated solve: one gather of the right-hand side"
            else:
                assert nx > 4 * it, "distributed solve: ghost fills, restriction and flux registers cross ranks in every operator application"
        worst = 0.0
        for l, dl in enumerate(dls):
            got = dsol[l].download()
            for i, g in enumerate(dl.gids):
                worst = max(worst, float(np.abs(got.valid(i)[0] - want[l].valid(int(g))[0]).max()))
        assert worst <= (1e-10 if stiff else 1e-12), f"rank {rank}/{world}: smoothed field differs from the undistributed oracle by {worst}"
        dist.barrier()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "nested"), (4, "nested"), (3, "walls"), (4, "nested+rep"), (3, "randu0"), (4, "randu13"), (2, "randu2"), (3, "randU0"), (4, "rand7"), (3, "2d"),
                                        (2, "nested+mg"), (4, "nested+mg"), (3, "walls+mg"), (3, "2d+mg")])
def test_sharded_smoothing_solve_matches_undistributed_oracle(world, case):
    """do_smooth with the hierarchy dealt to `world` ranks (scattered owners: fine boxes, their coarse parents and their
    neighbours mostly on different ranks): average_down and the flux register through the restriction plans, dot products
    through the transport's allreduce; the field equals the oracle's one-process composite solve to 1e-12 (both iterated to
    1e-14), with the one-rank iteration count.  `+rep`: the replicated form (PA_SMOOTH_REPLICATED=1)."""
    import torch.multiprocessing as mp
    mp.spawn(_smooth_worker, args=(world, _free_port(), case), nprocs=world, join=True)


def _ngpus():
    import torch
    return torch.cuda.device_count()  # (counting devices does not initialise the GPU)


@pytest.mark.skipif(_ngpus() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device (a 1-GPU box runs the single-rank RCCL test below)")
@pytest.mark.parametrize("case", ["wide", "thr", "randU0"])
def test_sharded_hierarchy_over_rccl_two_gpus(case):
    """the same sharded-hierarchy checks with the library's built-in transport and REAL peers: two ranks, one GPU each, the
    cross-rank FillBoundary / coarse-source / coarse-normal exchanges as ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd on
    the library's stream, the progress range through ncclAllReduce; results against the undistributed oracle, bit for bit.
    The first multi-GPU box that runs the GPU tier exercises it."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), case, "rccl"), nprocs=2, join=True)
    if case == "wide":  # + the distributed smoothing solve: its restriction / flux-register exchanges and ncclAllReduce dot products
        mp.spawn(_smooth_worker, args=(2, _free_port(), "nested", "rccl"), nprocs=2, join=True)


_RCCL_SELF = """
import sys
sys.path.insert(0, {root!r})
{torch_first}
from peleanalysis_amd import capi
c = capi.Context(0)
uid = c.rccl_unique_id()
assert len(uid) == 128 and any(uid)
c.init_rccl(1, 0, uid)
assert c.lib.pa_ctx_nranks(c.h) == 1
c.comm_selftest(1 << 16)
c.close()
print("RCCL_SELF_OK")
"""


@pytest.mark.parametrize("torch_first", [False, True])
def test_rccl_transport_single_rank(torch_first):
    """the built-in RCCL transport on the one GPU of the box: communicator of one rank, grouped ncclSend / ncclRecv to itself
    and ncclAllReduce, all issued by the library on its own stream (the multi-rank form of exactly these calls is what a
    multi-GPU run uses; the box has one GPU and RCCL refuses two ranks on one device).  In a fresh process, in the two
    library constellations that occur: a C++ tool (system HIP runtime + system RCCL) and bench.py (torch imported first:
    the HIP runtime and the RCCL that PyTorch bundles).  Mixing the two in one process is what INTEGRATION.md warns about."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _RCCL_SELF.format(root=root, torch_first="import torch; torch.cuda.init()" if torch_first else "")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_SELF_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
