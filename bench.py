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
N > 1   : weak scaling -- N copies of the hierarchy side by side in x (periodic), rank r owns slab r;
          the level-0 slab faces exchange 2 ghost layers per step over RCCL point-to-point.
"""
import argparse
import json
import os
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
    """Oracle (kind 'port') timed on the host cores: same pipeline, smaller hierarchy of the same shape."""
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
    reps, t0 = 0, time.perf_counter()
    while True:  # bounded sample: whole passes over the sample hierarchy until ~12 s of CPU work (at most 16 passes)
        O.grad_pipeline(H.levels, states, 0, bc, og, 0, multipass=False, omp=True)
        O.curvature_pipeline(H.levels, states, 0, bc, oc, 0, MultiFab, prog_min=300.0, prog_max=2000.0, omp=True)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= 12.0 or reps >= 16:
            break
    return {"value": cells * reps / dt / 1e6, "unit": "Mcells/s", "cores": cores, "kind": "port",
            "sample": f"oracle grad+curvature pipelines (C restatement, OpenMP over boxes), {nlev}-level base {base}^3, {box}^3 boxes, "
                      f"{cells} cells, 1 comp, {reps} passes, {dt:.1f} s"}


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
    args = ap.parse_args()

    import torch  # torch first: one HIP runtime in the process
    import torch.distributed as dist
    from peleanalysis_amd import capi
    from peleanalysis_amd import dist as padist
    from peleanalysis_amd.hierarchy import mf_layout, nested_hierarchy

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    # PA_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a box with FEWER GPUs than ranks -- ranks share the cards,
    # transport is gloo through host memory (RCCL wants one GPU per rank).  Same hierarchy split, same region lists,
    # same pack / unpack kernels, same reductions; the number it prints is not a scaling measurement.
    rehearse = os.environ.get("PA_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rdev = torch.device("cpu") if rehearse else dev  # where the tiny reduction tensors live
    gloo = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    stream = torch.cuda.Stream(device=dev)
    ctx = capi.Context(local, stream.cuda_stream)

    # weak scaling: every rank owns one full copy of the headline hierarchy (fixed work per GPU)
    per = tuple(int(v) for v in args.per.split())
    if world > 1 and per != (1, 1, 0):
        raise SystemExit("--per is a single-GPU diagnostic option")
    if world == 1:
        H = nested_hierarchy(args.base, args.nlev, args.box, is_per=per)
        remotes, plans = [None] * args.nlev, None
    else:
        R = padist.slab_hierarchy(args.base, args.nlev, args.box, world, rank, 2)
        H, remotes, plans = R.local, R.remote, R.plans
    bc = capi.bc_from_flags(per)
    dls = [capi.DevLevel(ctx, lv, remotes[l]) for l, lv in enumerate(H.levels)]
    cells = sum(lv.ncells for lv in H.levels)
    hold, states, works, outs = [], [], [], []
    with torch.cuda.stream(stream):
        for li, (lv, dl) in enumerate(zip(H.levels, dls)):
            off, cs, tot = mf_layout(lv.boxes, args.ncomp, 2)
            tin = torch.zeros(tot, dtype=torch.float64, device=dev)
            fill_level_on_device(torch, lv, tin, args.ncomp, 2, off, cs, dev, 1234 + 100 * rank + li)
            _, _, tw = mf_layout(lv.boxes, 1, 2)
            _, _, to = mf_layout(lv.boxes, 8, 0)
            twk = torch.zeros(tw, dtype=torch.float64, device=dev)
            tout = torch.zeros(to, dtype=torch.float64, device=dev)
            hold += [tin, twk, tout]
            states.append(capi.DevMF(ctx, dl, args.ncomp, 2, tin.data_ptr()))
            works.append(capi.DevMF(ctx, dl, 1, 2, twk.data_ptr()))
            outs.append(capi.DevMF(ctx, dl, 8, 0, tout.data_ptr()))
    stream.synchronize()
    params = capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=None, fused=bool(args.fused))
    xch = {"mode": "none", "bytes_per_step": 0}

    def exchange_rccl(c):
        for l in range(args.nlev):
            if plans[l].send or plans[l].recv:
                padist.exchange_device(plans[l], ctx, states[l], c, 1, dev)

    def exchange_gloo(c):  # host-staged fallback: same region lists and HIP pack/unpack, gloo transport
        for l in range(args.nlev):
            if plans[l].send or plans[l].recv:
                padist.exchange_device_staged(plans[l], ctx, states[l], c, 1, dev, group=gloo)

    do_exchange = None
    if world > 1:
        xch["bytes_per_step"] = int(sum(8 * pl.size(v, 1) for pl in plans for v in pl.send.values()) * args.ncomp)
        try:
            if rehearse:
                raise RuntimeError("PA_BENCH_REHEARSE=1: gloo transport")
            exchange_rccl(0)
            ok = torch.tensor([1], device=rdev)
        except Exception as e:  # keep the measurement valid (all work done) if RCCL p2p is unavailable
            xch["rccl_error"] = repr(e)[:300]
            ok = torch.tensor([0], device=rdev)
        try:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            use_rccl = bool(ok.item())
        except Exception:
            use_rccl = False
        if use_rccl:
            do_exchange, xch["mode"] = exchange_rccl, "RCCL point-to-point (batch_isend_irecv), one packed buffer per peer"
        else:
            gloo = None if rehearse else dist.new_group(backend="gloo")  # rehearsal: the default group is gloo already
            do_exchange, xch["mode"] = exchange_gloo, "host-staged gloo point-to-point (RCCL p2p failed)"

    def step():
        for c in range(args.ncomp):  # output buffers are recycled per component (SURVEY 8d memory budget)
            if do_exchange is not None:
                do_exchange(c)
            capi.gradcurv_run(ctx, states, c, bc, params, works, outs, 0)

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
    bd = {name: ctx.profile_read(tag)[1] / nbd for name, tag in (("gradcurv", 1), ("faces", 2), ("fill_boundary", 3), ("apply_bc", 4), ("progress", 6))}
    ctx.profile_read(1, reset=True)  # a reset drops every tag's records
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_cells = cells * args.ncomp * world
    value = total_cells * args.steps / dt / 1e6
    res = {
        "metric": "Mcells/s for grad+curvature on 512^3-base 3-level AMR; % HBM roofline",
        "value": value, "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"fused grad->curvature, {args.nlev}-level AMR, base {args.base}^3, ref_ratio 2, {args.box}^3 boxes "
                               f"({H.levels[0].nboxes} per level), {args.ncomp} comp(s), periodic x/y + wall z, {cells} cells per GPU",
                   "cells_per_gpu": cells, "ncomp": args.ncomp, "fused": bool(args.fused),
                   "parallelism": (f"{world} x-slabs (one hierarchy per GPU), level-0 slab faces exchanged per step"
                                   if world > 1 else "single GPU"),
                   "exchange": xch},
    }
    if nk:
        # one launch of the fused kernel = one level = cells/nlev cells (all levels have base^3 cells here)
        avg_ms = ms_k / nk
        cells_per_launch = cells / args.nlev
        ach = cells_per_launch * BYTES_PER_CELL / (avg_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per launch from the committed rocprofv3 PMC pass of the same workload (profiles/)
        tj = os.path.join(ROOT, "profiles", "r01_headline_traffic.json")  # refreshed by tools/prof.sh + tools/prof_traffic.py
        if os.path.exists(tj) and (args.base, args.nlev, args.box, args.ncomp) == (512, 3, 128, 1):
            traffic = json.load(open(tj)).get("traffic_bytes_per_launch")
        res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "kernel": "k_gradcurv_march3 (fused grad->curvature sweep)", "avg_launch_ms": avg_ms,
                           "launches": nk, "bytes_per_cell": BYTES_PER_CELL}
        res["breakdown_ms_per_step"] = bd  # from the untimed steps after the timed region
        res["step_frac_of_hbm_roofline"] = (cells * args.ncomp * BYTES_PER_CELL / (dt / args.steps) / 1e9) / HBM_PEAK_GBS
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            # bounded sample (~10-30 s of CPU work): a 3-level hierarchy sized from the host core count
            cores = usable_cpus()
            base = args.cpu_base or (256 if cores >= 16 else (128 if cores >= 8 else 64))
            res["cpu_baseline"] = cpu_baseline(base, args.nlev, max(base // 4, 8))
        except Exception as e:  # the baseline is reported, never required for the GPU number
            res["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
