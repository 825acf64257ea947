#!/usr/bin/env python3
"""bench.py -- headline benchmark: fused grad->curvature over a 512^3-base 3-level AMR hierarchy.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; N>1 is launched by
torch.distributed.run (one rank per GPU, RCCL).  One "step" = one pass of the hot path (cross-rank
ghost exchange when N>1, ghost fills, fused grad->curvature sweep, coarse-fine/wall face fix-up;
every level, every component) over synthetic input already resident in HBM.  ONE JSON line on rank 0.

metric  : Mcells/s = sum over levels of valid cells * ncomp / t   (BASELINE.json)
roofline: dominant kernel = the fused grad->curvature sweep; achieved = 72 B (read phi once, write
          gx,gy,gz,|g|,Nx,Ny,Nz,K) * cells per launch / average launch duration, measured with HIP
          events recorded by the library on its own stream inside the timed region.
cpu_baseline: the CPU oracle (oracle/, OpenMP over boxes, kind "port") on a bounded sample of the
          same workload (a smaller hierarchy of the same shape) on this node's host cores.
N > 1   : STRONG scaling by default -- the ONE headline hierarchy is sharded over the N ranks (Morton order + equal-volume
          cuts per level, pa_distribution_map); per step and component the library batches every cross-rank ghost fill
          (same-level ghost cells, coarse phi and coarse flame normal under the coarse-fine faces) into two grouped
          RCCL point-to-point exchanges on its own stream.  --scaling weak: N copies of the hierarchy side by side in x.
          `python bench.py --gpus N` without a launcher starts its own N ranks (torch.distributed.run).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
BYTES_PER_CELL = 72.0  # fused grad->curvature algorithmic bytes per cell per component


def torch_field_flame(torch, x, y, z, m):
    # periodic images of the flame kernel in x so that every slab of a multi-rank run holds a front
    xc, yc, zc = (x - torch.floor(x)) - 0.5, y - 0.5, z - 0.5
    r = torch.sqrt((xc / 0.30) ** 2 + (yc / 0.15) ** 2 + (zc / 0.18) ** 2)
    theta = torch.atan2(yc + 0 * xc, xc + 0 * yc)
    rho = torch.sqrt(xc * xc + yc * yc + zc * zc) + 1e-30
    phi = torch.acos(torch.clamp(zc / rho, -1.0, 1.0))
    s = r - 0.03 * torch.sin(6 * theta) * torch.sin(5 * phi)
    return (1.0 + 0.1 * m) * (300.0 + 850.0 * (1.0 + torch.tanh((s - 1.0) / 0.08))) + 3.0 * m * torch.sin(2 * np.pi * (x + 0.37 * m))


def fill_level_on_device(torch, level, buf, ncomp, ng, off, cs, dev, seed):
    """synthetic flame field + noise, written straight into HBM"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    dx = level.dx
    for b in range(level.nboxes):
        lo = level.boxes[b, :3]
        nz, ny, nx = level.box_shape(b, ng)
        x = (torch.arange(lo[0] - ng, lo[0] - ng + nx, device=dev, dtype=torch.float64) + 0.5) * dx[0] + level.prob_lo[0]
        y = (torch.arange(lo[1] - ng, lo[1] - ng + ny, device=dev, dtype=torch.float64) + 0.5) * dx[1] + level.prob_lo[1]
        z = (torch.arange(lo[2] - ng, lo[2] - ng + nz, device=dev, dtype=torch.float64) + 0.5) * dx[2] + level.prob_lo[2]
        X, Y, Z = x[None, None, :], y[None, :, None], z[:, None, None]
        n = nz * ny * nx
        for c in range(ncomp):
            v = torch_field_flame(torch, X, Y, Z, c).expand(nz, ny, nx)
            v = v + 1e-3 * (2.0 * torch.rand((nz, ny, nx), generator=g, device=dev, dtype=torch.float64) - 1.0)
            buf[off[b] + c * cs[b]: off[b] + c * cs[b] + n] = v.reshape(-1)


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a 16-CPU share
    of a 256-thread host to a 1-GPU job; OpenMP would otherwise start 256 threads on 16 CPUs' worth of time)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(base, nlev, box):
    """Oracle (kind 'port') timed on the host cores: the same grad + curvature pipelines on a hierarchy of the headline's
    shape and BOX SIZE with a smaller base (the sample).  Both variants of BASELINE.md section 3 that the oracle has:
      multipass  the gradient as the reference computes it (face-gradient arrays -> 1/bscalar -> average_face_to_cellcenter
                 -> mult(-1) -> magnitude, grad.cpp:211-236), the curvature pass by pass (curvature.cpp:310-570);
      stencil    the gradient as one central-difference sweep per level; the curvature pass by pass as above (the oracle has
                 no single-sweep CPU curvature: the boundary conditions on n sit between its passes).
    Scratch multifabs are allocated in an untimed first pass (MFPool); timed: whole passes until ~10 s per variant."""
    cores = usable_cpus()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # before the OpenMP build of the oracle is loaded
    from oracle import oracle as O
    from peleanalysis_amd.hierarchy import MultiFab, fill_analytic, nested_hierarchy, field_flame
    O.build()
    H = nested_hierarchy(base, nlev, box, is_per=(1, 1, 0))
    bc = O.bc_from_flags((1, 1, 0))
    states = []
    for lv in H.levels:
        s = MultiFab(lv, 1, 2)
        fill_analytic(s, 0, lambda x, y, z: field_flame(x, y, z, 0), valid_only=False)
        states.append(s)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    cells = sum(lv.ncells for lv in H.levels)
    variants = {}
    for name, multipass in (("stencil", False), ("multipass", True)):
        pool = O.MFPool(MultiFab)

        def one_pass():
            pool.start_pass()
            O.grad_pipeline(H.levels, states, 0, bc, og, 0, multipass=multipass, omp=True)
            O.curvature_pipeline(H.levels, states, 0, bc, oc, 0, pool, prog_min=300.0, prog_max=2000.0, omp=True)
        one_pass()  # untimed: allocates the scratch multifabs, pages everything in
        reps, t0 = 0, time.perf_counter()
        while True:
            one_pass()
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= 10.0 or reps >= 8:
                break
        variants[name] = {"Mcells/s": cells * reps / dt / 1e6, "passes": reps, "seconds": round(dt, 2)}
    best = max(variants, key=lambda k: variants[k]["Mcells/s"])
    return {"value": variants[best]["Mcells/s"], "unit": "Mcells/s", "cores": cores, "kind": "port", "variant": best, "variants": variants,
            "sample": f"oracle grad+curvature pipelines (C restatement, OpenMP over boxes, scratch preallocated), {nlev}-level base {base}^3, "
                      f"{box}^3 boxes{' (the GPU line box size)' if box == 128 else ''}, {cells} cells = {cells / (3 * 512 ** 3):.3f} of the headline hierarchy, 1 comp"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--base", type=int, default=512, help="base-level cells per direction (headline: 512)")
    ap.add_argument("--nlev", type=int, default=3)
    ap.add_argument("--box", type=int, default=128)
    ap.add_argument("--ncomp", type=int, default=1, help="components pushed through grad->curvature per step")
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--per", type=str, default="1 1 0", help="periodicity flags x y z (headline: periodic x/y, wall z); single GPU only")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: no HIP events in the timed region (no roofline object)")
    ap.add_argument("--cpu-base", type=int, default=0, help="base size of the cpu_baseline sample (0: from the core count)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong", help="N > 1: shard ONE hierarchy (strong) or one hierarchy per GPU (weak)")
    ap.add_argument("--sim-of", type=int, default=0, help="diagnostic, 1 GPU: time rank 0's share of an N-rank strong-scaling run with no-op exchanges")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.sim_of:
        # No launcher: this process becomes one and never touches the GPU (a process that has initialised HIP must not be
        # replaced or forked into ranks).  One rank per GPU, exactly as the driver's torch.distributed.run line does.
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch  # torch first: one HIP runtime in the process
    import torch.distributed as dist
    from peleanalysis_amd import capi
    from peleanalysis_amd import dist as padist
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box, mf_layout, nested_hierarchy

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    # PA_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a box with FEWER GPUs than ranks -- ranks share the cards,
    # transport is the pa_comm callback over gloo through host memory (RCCL wants one GPU per rank).  Same sharding, same
    # region plans, same pack / unpack kernels, same reductions; the number it prints is not a scaling measurement.
    rehearse = os.environ.get("PA_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # torch.distributed (gloo) is the control plane only: rendezvous, the 128-byte RCCL id, barriers, the max over
        # ranks of the elapsed time.  The data plane is the library's own RCCL communicator.
        dist.init_process_group("gloo")
    stream = torch.cuda.Stream(device=dev)
    ctx = capi.Context(local, stream.cuda_stream)

    per = tuple(int(v) for v in args.per.split())
    if world > 1 and per != (1, 1, 0):
        raise SystemExit("--per is a single-GPU diagnostic option")
    nshard, myrank = (args.sim_of, 0) if args.sim_of else (world, rank)
    if args.scaling == "weak" and nshard > 1:
        # N copies of the headline hierarchy side by side in x (global level 0 is N*base x base x base, periodic in x);
        # rank r owns copy r on every level
        H1 = nested_hierarchy(args.base, args.nlev, args.box, is_per=per)
        levels, owners = [], []
        for l, lv in enumerate(H1.levels):
            n0 = int(lv.domhi[0] - lv.domlo[0] + 1)
            bx = np.vstack([lv.boxes + np.array([r * n0, 0, 0, r * n0, 0, 0], dtype=np.int32) for r in range(nshard)])
            levels.append(Level(bx, lv.domlo, lv.domhi + np.array([(nshard - 1) * n0, 0, 0]), lv.is_per, lv.prob_lo, lv.prob_hi * np.array([nshard, 1, 1])))
            owners.append(np.repeat(np.arange(nshard, dtype=np.int32), lv.nboxes))
        H = Hierarchy(levels, 2)
    else:
        H = nested_hierarchy(args.base, args.nlev, args.box, is_per=per)
        owners = padist.shard(H, nshard) if nshard > 1 else [None] * args.nlev
    bc = capi.bc_from_flags(per)
    xch = {"mode": "none"}
    gcomm = None
    if args.sim_of:  # one rank's share of an N-rank run on this GPU, exchanges replaced by no-ops: compute time per rank
        nullx = capi.EXCHANGE_FN(lambda user, st, n, x: 0)
        nullr = capi.ALLREDUCE_FN(lambda user, v, n, op: 0)
        simc = capi.PaComm(None, 0, nshard, nullx, nullr)
        ctx.set_comm(simc)
        xch["mode"] = f"SIMULATION of rank 0 of {nshard}: exchanges are no-ops (results wrong in ghost cells, timing = compute only)"
    elif world > 1:
        err = ""
        if not rehearse:
            # built-in transport: grouped ncclSend / ncclRecv on the library's stream; a ring exchange + a reduction with known
            # answers before it is trusted.  Bring-up runs on a helper thread under a time limit: a communicator that never
            # forms (ncclCommInitRank blocks until every rank has joined) must degrade to the gloo transport, not eat the run.
            import threading
            box = {}

            def bring_up():
                try:
                    padist.init_rccl(ctx)
                    ctx.comm_selftest(1 << 16)
                    box["ok"] = True
                except Exception as e:
                    box["err"] = repr(e)[:300]

            limit = float(os.environ.get("PA_RCCL_TIMEOUT", "120"))
            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(limit)
            if th.is_alive():
                err = f"RCCL bring-up did not finish within {limit:.0f} s (ncclCommInitRank / self-test)"
                # the helper may still be inside the library with this context: leave it behind, continue on a fresh one
                ctx = capi.Context(local, stream.cuda_stream)
            else:
                err = box.get("err", "")
        ok = [None] * world
        dist.all_gather_object(ok, err)
        if rehearse or any(ok):
            gcomm = padist.GlooComm(ctx)    # pa_comm callbacks: packed buffers staged through host memory + gloo
            ctx.comm_selftest(1 << 12)
            xch["mode"] = "host-staged gloo point-to-point (pa_comm callbacks)" + ("" if rehearse else " -- RCCL transport failed: " + next(e for e in ok if e))
        else:
            xch["mode"] = "RCCL point-to-point (grouped ncclSend/ncclRecv issued by the library on its stream), 2 exchanges per component per step"
    dls = [capi.DevLevel(ctx, lv, owners[l], myrank, nshard) if nshard > 1 else capi.DevLevel(ctx, lv) for l, lv in enumerate(H.levels)]
    cells = sum(lv.ncells for lv in H.levels)                # the whole job
    cells_local = sum(dl.level.ncells for dl in dls)         # this rank's share
    hold, states, works, outs = [], [], [], []
    with torch.cuda.stream(stream):
        for li, dl in enumerate(dls):
            lv = dl.level
            off, cs, tot = mf_layout(lv.boxes, args.ncomp, 2)
            tin = torch.zeros(max(tot, 1), dtype=torch.float64, device=dev)
            fill_level_on_device(torch, lv, tin, args.ncomp, 2, off, cs, dev, 1234 + 100 * rank + li)
            _, _, tw = mf_layout(lv.boxes, 1, 2)
            _, _, to = mf_layout(lv.boxes, 8, 0)
            twk = torch.zeros(max(tw, 1), dtype=torch.float64, device=dev)
            tout = torch.zeros(max(to, 1), dtype=torch.float64, device=dev)
            hold += [tin, twk, tout]
            states.append(capi.DevMF(ctx, dl, args.ncomp, 2, tin.data_ptr()))
            works.append(capi.DevMF(ctx, dl, 1, 2, twk.data_ptr()))
            outs.append(capi.DevMF(ctx, dl, 8, 0, tout.data_ptr()))
    stream.synchronize()
    params = capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=None, fused=bool(args.fused))

    def step():
        # every component through the pipeline into recycled output buffers (SURVEY 8d memory budget); cross-rank ghost fills
        # happen inside; result-independent ghost fills are done once for all components
        capi.gradcurv_run_comps(ctx, states, 0, args.ncomp, bc, params, works, outs, 0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_read(1, reset=True)
    ctx.profile_enable(0 if args.no_profile else (1 << 1))  # timed region: HIP events around the dominant kernel only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    nk, ms_k = ctx.profile_read(1, reset=True)
    assert ctx.bc_errors() == 0
    # per-stage breakdown from two further, untimed steps with every stage's launches timed
    nbd = 2
    ctx.profile_enable(True)
    for _ in range(nbd):
        step()
    barrier()
    ctx.profile_enable(False)
    bd = {name: ctx.profile_read(tag)[1] / nbd for name, tag in (("gradcurv", 1), ("faces", 2), ("fill_boundary", 3), ("apply_bc", 4), ("progress", 6),
                                                                  ("exchange", 9))}
    ctx.profile_read(1, reset=True)  # a reset drops every tag's records
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    weak = args.scaling == "weak" and nshard > 1
    nb0 = H.levels[0].nboxes
    value = cells * args.ncomp * args.steps / dt / 1e6  # whole job: every cell of every level and rank
    if args.sim_of:  # a simulation is not a measurement of the whole job: report what this GPU really processed
        value = cells_local * args.ncomp * args.steps / dt / 1e6
    per_txt = {(1, 1, 0): "periodic x/y + wall z"}.get(per, f"is_per {per}")
    if world == 1 and not args.sim_of:
        par = "single GPU"
    elif weak:
        par = f"{nshard} hierarchies side by side in x (weak scaling), rank r owns copy r"
    else:
        par = (f"ONE hierarchy sharded over {nshard} ranks (Morton order + equal-volume cuts per level): "
               f"{', '.join(str(dl.level.nboxes) for dl in dls)} boxes per level on rank {myrank}")
    res = {
        "metric": "Mcells/s for grad+curvature on 512^3-base 3-level AMR; % HBM roofline",
        "value": value, "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"fused grad->curvature, {args.nlev}-level AMR, base {args.base}^3{' per GPU' if weak else ''}, ref_ratio 2, {args.box}^3 boxes "
                               f"({nb0} per level), {args.ncomp} comp(s), {per_txt}, {cells} cells in the job",
                   "cells": cells, "cells_this_rank": cells_local, "ncomp": args.ncomp, "fused": bool(args.fused),
                   "parallelism": par, "exchange": xch},
    }
    if args.sim_of:
        res["simulated"] = True
        res["sim_of"] = nshard
        res["projected_whole_job_Mcells_s_compute_only"] = cells * args.ncomp * args.steps / dt / 1e6
        res["note"] = ("--sim-of: rank 0's share of an N-rank strong-scaling run on ONE GPU with no-op exchanges; value = the cells this GPU "
                       "processed / time; the projected whole-job figure ignores communication and is not a measurement")
    if nk:
        # one launch of the fused kernel = this rank's boxes of every level (k_gradcurv_march3_levels) or of one level;
        # achieved = algorithmic bytes of all timed launches / their time
        avg_ms = ms_k / nk
        ach = cells_local * args.ncomp * args.steps * BYTES_PER_CELL / (ms_k * 1e-3) / 1e9
        # HBM bytes per launch from this round's rocprofv3 PMC passes of the same workload (profiles/, tools/r2_pmc.sh): only
        # valid for the kernel variant it was measured on -- null when the library launched another one
        traffic, kern = None, ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
        tj = os.path.join(ROOT, "profiles", "r02_headline_traffic.json")
        if os.path.exists(tj) and (args.base, args.nlev, args.box, args.ncomp, world, args.sim_of) == (512, 3, 128, 1, 1, 0):
            rec = json.load(open(tj))
            if rec.get("kernel") == kern:
                traffic = rec.get("traffic_bytes_per_launch")
        res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "kernel": kern + " (fused grad->curvature sweep)", "avg_launch_ms": avg_ms,
                           "launches": nk, "bytes_per_cell": BYTES_PER_CELL, "cells_per_launch": cells_local * args.ncomp * args.steps / nk}
        res["breakdown_ms_per_step"] = bd  # from the untimed steps after the timed region (rank 0)
        res["step_frac_of_hbm_roofline"] = (cells_local * args.ncomp * BYTES_PER_CELL / (dt / args.steps) / 1e9) / HBM_PEAK_GBS
    if rank == 0 and world == 1 and not args.no_cpu and not args.sim_of:
        try:
            # bounded sample (~10-30 s of CPU work): a 3-level hierarchy sized from the host core count
            cores = usable_cpus()
            base = args.cpu_base or (384 if cores >= 16 else (256 if cores >= 8 else 128))
            res["cpu_baseline"] = cpu_baseline(base, args.nlev, min(args.box, base // 2))
        except Exception as e:  # the baseline is reported, never required for the GPU number
            res["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
