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
      stencil    the gradient as one central-difference sweep per level; the curvature pass by pass as above;
      fused      BASELINE.md section 3 (ii): gradient, normal and curvature of a level in ONE sweep over its cells (every cell rebuilds
                 its six neighbours' normals from c), then the first layer of every box from the stored normals with the reference's
                 boundary conditions on n (oracle.gradcurv_fused_pipeline); its outputs are compared bit for bit with multipass's.
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
    g2 = [MultiFab(lv, 4, 0) for lv in H.levels]
    n2 = [MultiFab(lv, 3, 1) for lv in H.levels]
    k2 = [MultiFab(lv, 1, 0) for lv in H.levels]
    for name, multipass in (("fused", None), ("stencil", False), ("multipass", True)):  # multipass last: og / oc then hold the reference-shaped results
        pool = O.MFPool(MultiFab)

        def one_pass():
            pool.start_pass()
            if multipass is None:  # one sweep per level + the first layer of every box (orc_gradcurv_fused)
                O.gradcurv_fused_pipeline(H.levels, states, 0, bc, g2, n2, k2, pool, 300.0, 2000.0, omp=True)
                return
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
    # the fused variant's outputs against the reference-shaped multipass ones, bit for bit (a variant that computed something else would
    # not be a baseline of this path)
    same = True
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            same = same and np.array_equal(g2[l].valid(b).view(np.int64), og[l].valid(b).view(np.int64))
            same = same and np.array_equal(np.ascontiguousarray(n2[l].valid(b)).view(np.int64), np.ascontiguousarray(oc[l].valid(b)[2:5]).view(np.int64))
            same = same and np.array_equal(k2[l].valid(b)[0].view(np.int64), np.ascontiguousarray(oc[l].valid(b)[1]).view(np.int64))
    variants["fused"]["bits_equal_to_multipass"] = bool(same)
    del g2, n2, k2
    best = max(variants, key=lambda k: variants[k]["Mcells/s"])
    res = {"value": variants[best]["Mcells/s"], "unit": "Mcells/s", "cores": cores, "kind": "port", "variant": best, "variants": variants,
           "sample": f"oracle grad+curvature pipelines (C restatement, OpenMP over boxes, scratch preallocated), {nlev}-level base {base}^3, "
                     f"{box}^3 boxes{' (the GPU line box size)' if box == 128 else ''}, {cells} cells = {cells / (3 * 512 ** 3):.3f} of the headline hierarchy, 1 comp"}
    return res, (H, states, og, oc)


def parity_check(ctx, sample, retile=1):
    """The oracle's outputs of the cpu_baseline sample (production geometry: 128^3 boxes, two x tiles x ten row tiles x two z
    segments per box, all levels in one sweep launch) against the HIP path on the SAME inputs, bit for bit: the checker's work
    is already paid for by the baseline leg, the GPU pass and the download add a few seconds."""
    from peleanalysis_amd import capi
    from peleanalysis_amd.hierarchy import MultiFab, regrid_copy, retile_hierarchy
    H, states, og, oc = sample
    t0 = time.perf_counter()
    bc = capi.bc_from_flags((1, 1, 0))
    # as the tools: the file's FABs into the internal tiling, the HIP path there, the results back per FILE box -- against the oracle
    # run on the file's boxes
    T = retile_hierarchy(H) if retile else H
    if retile:
        tst = []
        for s, tv in zip(states, T.levels):
            m = MultiFab(tv, s.ncomp, s.ng)
            regrid_copy(s, m)
            tst.append(m)
    else:
        tst = states
    dls = [capi.DevLevel(ctx, lv) for lv in T.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, tst)]
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=None, fused=True), work, dout, 0)
    ctx.sync()
    kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
    nbad, ncmp = 0, 0
    pairs = [(0, og, 0), (1, og, 1), (2, og, 2), (3, og, 3), (4, oc, 2), (5, oc, 3), (6, oc, 4), (7, oc, 1)]
    for l, lv in enumerate(H.levels):
        got = dout[l].download()
        if retile:
            back = MultiFab(lv, 8, 0)
            regrid_copy(got, back)
            got = back
        for b in range(lv.nboxes):
            g = got.valid(b)
            for gc, ref, rc in pairs:
                w = ref[l].valid(b)[rc]
                nbad += int(np.count_nonzero(np.ascontiguousarray(g[gc]).view(np.int64) != np.ascontiguousarray(w).view(np.int64)))
                ncmp += w.size
        del got
    return {"cells": sum(lv.ncells for lv in H.levels), "values_compared": ncmp, "values_differing": nbad, "bits_equal": nbad == 0 and ctx.bc_errors() == 0,
            "kernel": kn, "tiling": {"file_boxes_per_level": [lv.nboxes for lv in H.levels], "swept_boxes_per_level": [lv.nboxes for lv in T.levels]}, "outputs": "gx gy gz |g| Nx Ny Nz K of every valid cell of every level, oracle (cpu_baseline sample) vs HIP path, int64 view",
            "seconds": round(time.perf_counter() - t0, 2)}


def live_traffic(timeout_s=150, retile=1):
    """HBM bytes per launch of the fused sweep, measured NOW on this box: two child runs of the torch-free driver on the
    headline hierarchy under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950: 64 B counted per 128-B request).  Returns (bytes or None, note)."""
    import csv
    import glob
    import shutil
    import tempfile
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_OUTPUT_PATH"):
        return None, "this process runs under a profiler: no nested counter pass"
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    per = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pa_pmc_", dir="/tmp")
        try:
            subprocess.run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "prof_driver.py"), "512", "128", "2", str(int(retile))],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, env=env, cwd=ROOT, check=True)
            tot, ids = 0.0, set()
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_gradcurv_march3" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        tot += float(row["Counter_Value"])
                        ids.add(row["Dispatch_Id"])
            if not ids:
                return None, f"no {ctr} rows for the sweep kernel"
            per[ctr] = tot / len(ids) * 1024.0  # KiB -> bytes per launch
        except Exception as e:  # a missing tool, a timeout, a refused counter: the committed figure is used instead
            return None, f"{ctr} pass failed: {repr(e)[:120]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]), "measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 tools/prof_driver.py 512 128 2 " + str(int(retile)) + " (FETCH x2)"


def secondary(ctx, torch, stream, dev, only=None, retile=1):
    """The other kernel families of the path in front of the driver (N = 1 only, after the timed headline region; a few
    seconds in all): BASELINE configs 2, 3 and 4, the gradient alone, and the headline hierarchy in 64^3 and 32^3 boxes.
    Each entry: wall-clock ms per pass on the library's stream (launches + stream sync, second and third pass), the
    algorithmic bytes per cell of SURVEY 8(d) and the fraction of the 8 TB/s HBM figure they amount to."""
    import ctypes as C
    from peleanalysis_amd import capi
    from peleanalysis_amd.hierarchy import mf_layout, nested_hierarchy

    def alloc(lv, dl, ncomp, ng, fill=None, seed=1):
        off, cs, tot = mf_layout(lv.boxes, ncomp, ng)
        with torch.cuda.stream(stream):
            t = torch.zeros(max(tot, 1), dtype=torch.float64, device=dev)
            if fill == "flame":  # component 0 the flame field + noise, the others cheap affine images of it
                fill_level_on_device(torch, lv, t, 1, ng, off, cs, dev, seed)
                for b in range(lv.nboxes):
                    nz, ny, nx = lv.box_shape(b, ng)
                    n = nz * ny * nx
                    for c in range(1, ncomp):
                        t[off[b] + c * cs[b]: off[b] + c * cs[b] + n] = (1.0 + 0.1 * c) * t[off[b]: off[b] + n] + 3.0 * c
        return t, capi.DevMF(ctx, dl, ncomp, ng, t.data_ptr())

    def timed(fn, reps=8):
        # steady state, as the headline's K timed steps: one untimed call, then `reps` calls back to back (the host enqueues ahead of the
        # GPU; with 2 calls -- rounds 1-5 -- the first call's enqueue latency, ~0.3 ms, was 3-5 % of a 6-ms pass)
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    def entry(ms, cells, bpc, **kw):
        e = {"ms": ms, "cells": cells, "bytes_per_cell": bpc, "Mcells_s": cells / ms / 1e3}
        if bpc:
            e["frac_hbm"] = cells * bpc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        e.update(kw)
        return e

    out = {}

    def want(name):  # --secondary-only <name>: one entry (kernel work on one family without the rest of the bench)
        return only is None or only == name

    def tiling_txt(Hf, T):
        return (f"file tiling {[lv.nboxes for lv in Hf.levels]} boxes per level, swept on the internal tiling pa_level_retile makes of it: "
                f"{[lv.nboxes for lv in T.levels]} boxes per level") if T is not Hf else f"swept on the file's tiling ({[lv.nboxes for lv in Hf.levels]} boxes per level)"

    def gradcurv_on(T, ncomp, per, nbatch, seed0=77):
        """ms per pass of pa_gradcurv_run_comps2 on the tiling T (allocated, run, released)"""
        dls = [capi.DevLevel(ctx, lv) for lv in T.levels]
        keep, st, wk, ou = [], [], [], []
        for li, (lv, dl) in enumerate(zip(T.levels, dls)):
            a, b_, c_ = alloc(lv, dl, ncomp, 2, "flame", seed0 + li), alloc(lv, dl, 1, 2), alloc(lv, dl, 8 * nbatch, 0)
            keep += [a[0], b_[0], c_[0]]
            st.append(a[1]); wk.append(b_[1]); ou.append(c_[1])
        stream.synchronize()
        bc = capi.bc_from_flags(per)
        params = capi.curv_params(prog_min=300.0, prog_max=2000.0 * (1.0 + 0.1 * ncomp) + 3.0 * ncomp, threshold=None, fused=True)
        ms = timed(lambda: capi.gradcurv_run_comps2(ctx, st, 0, ncomp, bc, params, wk, ou, 0, nbatch))
        assert ctx.bc_errors() == 0
        kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
        for x in st + wk + ou:
            x.close()
        for d in dls:
            d.close()
        del keep
        torch.cuda.empty_cache()
        return ms, kn

    def gradcurv_case(name, base, nlev, box, ncomp, per, nbatch=1, also_file=False):
        # as the tools run a plotfile: the file's boxes (box^3) merged by pa_level_retile, the sweep on the merged boxes; also_file:
        # the same pass on the file's own boxes beside it (retile=0)
        from peleanalysis_amd.hierarchy import retile_hierarchy
        H = nested_hierarchy(base, nlev, box, is_per=per)
        T = retile_hierarchy(H) if retile else H
        cells = sum(lv.ncells for lv in H.levels) * ncomp
        ms, kn = gradcurv_on(T, ncomp, per, nbatch)
        extra = {}
        if also_file and T is not H:
            ms_f, kn_f = gradcurv_on(H, ncomp, per, nbatch)
            extra = {"file_tiling_ms": ms_f, "file_tiling_frac_hbm": cells * 72 / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, "file_tiling_sweep_kernel": kn_f}
        out[name] = entry(ms, cells, 72, sweep_kernel=kn, workload=f"fused grad->curvature, {nlev}-level base {base}^3, {box}^3 boxes in the file, {ncomp} comp(s)" +
                          (f" in batches of {nbatch} (pa_gradcurv_run_comps2)" if nbatch > 1 else "") + f", is_per {per}; " + tiling_txt(H, T), **extra)

    # BASELINE config 2: single level 512^3, 10 components
    if want("c2_1lev_512_10comp"):
        gradcurv_case("c2_1lev_512_10comp", 512, 1, 128, 10, (1, 1, 1), nbatch=int(os.environ.get("PA_C2_NBATCH", "10")))  # all 10 components in one batch: one FillBoundary launch and one sweep launch (slots) for all of them, 86 GB of outputs; 1: component by component (20.8 against 19.0 ms)
    # BASELINE config 5's shape on one GPU, 8 of its 55 components: 4 levels of 256^3 cells in 64^3 boxes, components in one batch
    if want("c5_shape_4lev_256_8comp"):
        gradcurv_case("c5_shape_4lev_256_8comp", 256, 4, 64, 8, (1, 1, 0), nbatch=8, also_file=True)
    # the headline itself on the FILE's own 128^3 boxes (--retile 0) beside the default, so that rounds before and after the
    # internal tiling stay comparable (advisor, round 5): ms = the default (what the headline line measures), file_tiling_ms = retile 0
    if want("headline_box128"):
        gradcurv_case("headline_box128", 512, 3, 128, 1, (1, 1, 0), also_file=True)
    # the headline hierarchy in smaller boxes
    for box in (64, 32):
        if not want(f"headline_box{box}"):
            continue
        gradcurv_case(f"headline_box{box}", 512, 3, box, 1, (1, 1, 0), also_file=True)
    def irregular_case():
        # An IRREGULAR hierarchy, what a Pele plotfile holds (grad.cpp:173-213 / curvature.cpp:426-457 run on whatever BoxArray the
        # file has): the headline field on a 512^3 base, finer levels tagged where |grad T| is largest -- the wrinkled flame sheet --
        # in blocks of 32 fine cells, merged into boxes of 32 .. 128 cells per side.  L-shaped regions, faces that are partly
        # covered by a neighbour and partly coarse-fine, concave coarse-fine corners; the irregular cells go through k_curv_general.
        # As the tools do, the pass runs on the internal tiling pa_level_retile makes of those boxes; the file's own tiling beside it.
        from peleanalysis_amd.hierarchy import tagged_hierarchy, field_flame, retile_hierarchy
        Hf = tagged_hierarchy(512, 3, lambda x, y, z: field_flame(x, y, z, 0), bf=16, max_box=128, base_box=128, frac=(0.08, 0.16), is_per=(1, 1, 0))
        Hi = retile_hierarchy(Hf) if retile else Hf
        ci = sum(lv.ncells for lv in Hf.levels)
        extra = {}
        if Hi is not Hf:
            ms_f, kn_f = gradcurv_on(Hf, 1, (1, 1, 0), 1, seed0=177)
            extra = {"file_tiling_ms": ms_f, "file_tiling_frac_hbm": ci * 72 / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, "file_tiling_sweep_kernel": kn_f}
        dli = [capi.DevLevel(ctx, lv) for lv in Hi.levels]
        keep, st, wk, ou = [], [], [], []
        for li, (lv, dl) in enumerate(zip(Hi.levels, dli)):
            a, b_, c_ = alloc(lv, dl, 1, 2, "flame", 177 + li), alloc(lv, dl, 1, 2), alloc(lv, dl, 8, 0)
            keep += [a[0], b_[0], c_[0]]
            st.append(a[1]); wk.append(b_[1]); ou.append(c_[1])
        stream.synchronize()
        bci = capi.bc_from_flags((1, 1, 0))
        pari = capi.curv_params(prog_min=300.0, prog_max=2003.0, threshold=None, fused=True)
        nirr = [int(ctx.lib.pa_level_irregular_cells(ctx.h, dl.h)) for dl in dli]
        ms = timed(lambda: capi.gradcurv_run(ctx, st, 0, bci, pari, wk, ou, 0))
        assert ctx.bc_errors() == 0
        kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
        # the two independent kernel sets on the same irregular hierarchy, bit for bit on the device: the exact-normal pipeline with
        # its irregular-cell list against the pass-by-pass kernels (each parity-tested against the oracle at small sizes)
        capi.gradcurv_run(ctx, st, 0, bci, pari, wk, ou, 0)
        ctx.sync()
        keep_f = [k.clone() for k in keep[2::3]]  # the output multifabs of the three levels
        # (the pass-by-pass path allocates its multi-GB work multifabs per call; a hipMalloc that does not get the previous call's block back costs
        # hundreds of ms, so the smaller of two single passes is reported)
        pp = lambda: capi.gradcurv_run(ctx, st, 0, bci, capi.curv_params(prog_min=300.0, prog_max=2003.0, threshold=None, fused=False), wk, ou, 0)
        ms_pp = min(timed(pp, reps=1), timed(pp, reps=1))
        ctx.sync()
        gou = [alloc(lv, dl, 4, 0) for lv, dl in zip(Hi.levels, dli)]  # the gradient tool's pass on the same BoxArrays (40 B/cell)
        stream.synchronize()
        ms_grad = timed(lambda: capi.grad_run(ctx, st, 0, bci, [g[1] for g in gou], 0))
        del gou
        ndiff = 0
        for a, b_ in zip(keep_f, keep[2::3]):
            ndiff += int((a.view(torch.int64) != b_.view(torch.int64)).sum().item())
        del keep_f
        wid = lambda H_: [np.bincount(np.minimum((lv.boxes[:, 3] - lv.boxes[:, 0] + 1 + 31) // 32, 9).astype(int), minlength=10)[1:].tolist() for lv in H_.levels]
        out["irregular_amr"] = entry(ms, ci, 72, boxes_per_level=[lv.nboxes for lv in Hi.levels], cells_per_level=[lv.ncells for lv in Hi.levels],
                                     file_boxes_per_level=[lv.nboxes for lv in Hf.levels],
                                     boxes_by_width_in_32s_per_level=wid(Hi), file_boxes_by_width_in_32s_per_level=wid(Hf),
                                     irregular_cells_per_level=nirr, irregular_cell_share=sum(nirr) / ci,
                                     share_of_boxes_on_fused_pipeline=1.0 if ("CG=1" in kn or "march3_levels" in kn) else 0.0, sweep_kernel=kn,
                                     pass_by_pass_ms=ms_pp, pass_by_pass_frac_hbm=ci * 72 / (ms_pp * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     grad_only_ms=ms_grad, grad_only_frac_hbm=ci * 40 / (ms_grad * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     fused_vs_pass_by_pass_values_differing=ndiff,
                                     workload="fused grad->curvature, 3-level AMR, base 512^3, levels 1-2 = the blocks of 32 fine cells with the largest |grad T| (8 % / 16 % of "
                                              "the coarser level's blocks: the wrinkled flame sheet), boxes of 32..128 cells per side in the file, 1 comp, periodic x/y + wall z; " +
                                              tiling_txt(Hf, Hi) + "; pass_by_pass_ms = the same tiling with fused=0 (the pre-round-4 path for such BoxArrays)", **extra)
        del keep, st, wk, ou, dli, Hi
        torch.cuda.empty_cache()
    if want("irregular_amr"):
        irregular_case()
    def grad_only_case():
        # the gradient alone on the headline hierarchy (grad.cpp:211-236; 40 B/cell)
        from peleanalysis_amd.hierarchy import retile_hierarchy
        Hf = nested_hierarchy(512, 3, 128, is_per=(1, 1, 0))
        H = retile_hierarchy(Hf) if retile else Hf
        dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
        ins = [alloc(lv, dl, 1, 1, "flame", 5 + li) for li, (lv, dl) in enumerate(zip(H.levels, dls))]
        gos = [alloc(lv, dl, 4, 0) for lv, dl in zip(H.levels, dls)]
        stream.synchronize()
        bc = capi.bc_from_flags((1, 1, 0))
        ms = timed(lambda: capi.grad_run(ctx, [a[1] for a in ins], 0, bc, [g[1] for g in gos], 0))
        out["grad_only_headline"] = entry(ms, sum(lv.ncells for lv in H.levels), 40, workload="grad (ghost fills + k_grad_march), 3-level base 512^3, 128^3 boxes in the file, 1 comp; " + tiling_txt(Hf, H))
        del ins, gos, dls, H
        torch.cuda.empty_cache()

    if want("grad_only_headline"):
        grad_only_case()
    def c3_case():
        # BASELINE config 3: filterPlt's ghost fill + box filter (fgr 2 / 4 / 8 on levels 0 / 1 / 2) + grad of the filtered field
        from peleanalysis_amd.hierarchy import retile_hierarchy
        Hf = nested_hierarchy(256, 3, 64, is_per=(1, 1, 0))
        H = retile_hierarchy(Hf, (128, 128, 128)) if retile else Hf  # as filterPlt3d: the re-chopped file boxes merged (at most 128^3: k_filter_sep deals boxes to the XCDs), written back per file box
        bc = capi.bc_from_flags((1, 1, 0))
        dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
        ngs, ws = [1, 2, 4], []
        for f in (2, 4, 8):
            w = (C.c_double * (f + 2))()
            assert ctx.lib.pa_box_filter_weights(f, w) == f // 2
            ws.append(w)
        fin = [alloc(lv, dl, 1, ngs[l], "flame", 31 + l) for l, (lv, dl) in enumerate(zip(H.levels, dls))]
        fout = [alloc(lv, dl, 1, 1) for lv, dl in zip(H.levels, dls)]
        gout = [alloc(lv, dl, 4, 0) for lv, dl in zip(H.levels, dls)]
        stream.synchronize()

        def ghosts_levels():  # rounds 1-4: three calls per level
            for l in range(3):
                ctx.check(ctx.lib.pa_fill_boundary(ctx.h, fin[l][1].h, 0, 1, ngs[l]))
                if l > 0:
                    ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, fin[l][1].h, fin[l - 1][1].h, 0, 1, ngs[l], 2, 1))
                ctx.check(ctx.lib.pa_foextrap(ctx.h, fin[l][1].h, 0, 1, ngs[l]))

        hfin = (C.c_void_p * 3)(*[f[1].h for f in fin])
        hngs = (C.c_int32 * 3)(*ngs)

        def ghosts():  # filterPlt.cpp:159-203 for the whole hierarchy: FillBoundary / FillPatchTwoLevels / foextrap, one launch each
            ctx.check(ctx.lib.pa_fill_ghosts_hierarchy(ctx.h, 3, hfin, 0, 1, hngs, 2, 1, 1))

        def filt(l):
            ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, fin[l][1].h, fout[l][1].h, 0, 1, ngs[l], ws[l]))

        hin = (C.c_void_p * 3)(*[f[1].h for f in fin])
        hout = (C.c_void_p * 3)(*[f[1].h for f in fout])
        hng = (C.c_int32 * 3)(*ngs)
        hws = (C.POINTER(C.c_double) * 3)(*[C.cast(w, C.POINTER(C.c_double)) for w in ws])

        def filt_all():  # the three levels in one call (pa_boxfilter_hierarchy)
            ctx.check(ctx.lib.pa_boxfilter_hierarchy(ctx.h, 3, hin, hout, 0, 1, hng, hws))

        def c3_all():
            ghosts()
            filt_all()
            capi.grad_run(ctx, [f[1] for f in fout], 0, bc, [g[1] for g in gout], 0)

        c3cells = sum(lv.ncells for lv in H.levels)
        c3 = {"ghost_fill_ms": timed(ghosts), "ghost_fill_level_by_level_ms": timed(ghosts_levels)}
        for l in range(3):
            m = timed(lambda l=l: filt(l))
            c3[f"filter_fgr{2 << l}_level{l}"] = entry(m, H.levels[l].ncells, 16)
        c3["filter_all_levels_ms"] = timed(filt_all)
        c3["grad_ms"] = timed(lambda: capi.grad_run(ctx, [f[1] for f in fout], 0, bc, [g[1] for g in gout], 0))
        c3.update(entry(timed(c3_all), c3cells, None, workload="filterPlt ghost fill + separable box filter fgr 2/4/8 + grad, 3-level base 256^3, 64^3 boxes in the file, 1 comp; " + tiling_txt(Hf, H)))
        assert ctx.bc_errors() == 0
        out["c3_filter_grad_base256"] = c3
        del fin, fout, gout
        torch.cuda.empty_cache()

    if want("c3_filter_grad_base256"):
        c3_case()
    def c4_case():
        # BASELINE config 4: isosurface (isosurface.cpp:1434-1592) -- state build (coordinates, ghost fill) + level-batched marching cubes
        Hn = nested_hierarchy(256, 3, 64, is_per=(0, 0, 0))
        dln = [capi.DevLevel(ctx, lv) for lv in Hn.levels]
        fld = [alloc(lv, dl, 1, 1, "flame", 91 + l) for l, (lv, dl) in enumerate(zip(Hn.levels, dln))]
        sts = [alloc(lv, dl, 4, 1) for lv, dl in zip(Hn.levels, dln)]
        loops = []
        for lv in Hn.levels:
            arr = (capi.PaBox * lv.nboxes)()
            for b in range(lv.nboxes):
                for d in range(3):
                    arr[b].lo[d] = max(int(lv.boxes[b, d]) - 1, int(lv.domlo[d]))
                    arr[b].hi[d] = min(int(lv.boxes[b, 3 + d]) + 1, int(lv.domhi[d])) - 1
            loops.append(arr)
        stream.synchronize()
        tri = [0]

        def iso_state():
            for l in range(3):
                ctx.check(ctx.lib.pa_iso_coords_level(ctx.h, sts[l][1].h, 0))
                ctx.check(ctx.lib.pa_mf_copy(ctx.h, fld[l][1].h, 0, sts[l][1].h, 3, 1, 1))
                ctx.check(ctx.lib.pa_fill_boundary(ctx.h, sts[l][1].h, 0, 4, 1))
                if l > 0:
                    ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, sts[l][1].h, sts[l - 1][1].h, 0, 4, 1, 2, 0))

        def iso_mc_levels():  # one call per level (rounds 1-3): two host round trips and an allocation per level
            tri[0] = 0
            for l in range(3):
                nb = Hn.levels[l].nboxes
                nv, nt = (C.c_int64 * nb)(), (C.c_int64 * nb)()
                pv, pk, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()
                ctx.check(ctx.lib.pa_mc_level_fine(ctx.h, sts[l][1].h, dln[l + 1].h if l < 2 else None, 2, loops[l], 3, 1150.0, nv, nt, C.byref(pv), C.byref(pk), C.byref(pt)))
                tri[0] += int(sum(nt[:nb]))
                if pv.value:
                    ctx.lib.pa_device_free(ctx.h, pv)

        nvs = [(C.c_int64 * lv.nboxes)() for lv in Hn.levels]
        nts = [(C.c_int64 * lv.nboxes)() for lv in Hn.levels]
        parr = (C.POINTER(capi.PaBox) * 3)(*[C.cast(a, C.POINTER(capi.PaBox)) for a in loops])
        pnv = (C.POINTER(C.c_int64) * 3)(*[C.cast(a, C.POINTER(C.c_int64)) for a in nvs])
        pnt = (C.POINTER(C.c_int64) * 3)(*[C.cast(a, C.POINTER(C.c_int64)) for a in nts])
        fm = (C.c_int32 * 3)(1, 1, 0)
        hst = (C.c_void_p * 3)(*[s_[1].h for s_ in sts])

        def iso_mc():  # the whole hierarchy in one call: one count read-back, one pooled allocation, one final sync
            pv, pk, pt = (C.c_void_p * 3)(), (C.c_void_p * 3)(), (C.c_void_p * 3)()
            block = C.c_void_p()
            ctx.check(ctx.lib.pa_mc_hierarchy_fine(ctx.h, 3, hst, fm, 2, parr, 3, 1150.0, pnv, pnt, pv, pk, pt, C.byref(block)))
            tri[0] = sum(int(sum(nts[l][:Hn.levels[l].nboxes])) for l in range(3))
            if block.value:
                ctx.lib.pa_device_free(ctx.h, block)

        # round 5: the state is the field alone -- its ghost cells of all levels in two launches, vertex coordinates from cell indices
        hfld = (C.c_void_p * 3)(*[f_[1].h for f_ in fld])
        hng1 = (C.c_int32 * 3)(1, 1, 1)

        def iso_state_xyz():
            ctx.check(ctx.lib.pa_fill_ghosts_hierarchy(ctx.h, 3, hfld, 0, 1, hng1, 2, 0, 0))

        def iso_mc_xyz():
            pv, pk, pt = (C.c_void_p * 3)(), (C.c_void_p * 3)(), (C.c_void_p * 3)()
            block = C.c_void_p()
            ctx.check(ctx.lib.pa_mc_hierarchy_xyz(ctx.h, 3, hfld, fm, 2, parr, 0, 1150.0, pnv, pnt, pv, pk, pt, C.byref(block)))
            tri[0] = sum(int(sum(nts[l][:Hn.levels[l].nboxes])) for l in range(3))
            if block.value:
                ctx.lib.pa_device_free(ctx.h, block)

        def iso_all():  # isosurface.cpp:1434-1592 as the tool runs it: ghost fill of the state + marching cubes of the hierarchy
            iso_state_xyz()
            iso_mc_xyz()

        iso_state()
        c4cells = sum(lv.ncells for lv in Hn.levels)
        ms_state, ms_lev, ms_mc = timed(iso_state), timed(iso_mc_levels), timed(iso_mc)
        tri_stored = tri[0]
        ms_state_xyz, ms_mc_xyz, ms_all = timed(iso_state_xyz), timed(iso_mc_xyz), timed(iso_all)
        assert ctx.bc_errors() == 0 and tri[0] == tri_stored
        out["c4_isosurface_base256"] = entry(ms_all, c4cells, 8, state_build_ms=ms_state_xyz, marching_cubes_ms=ms_mc_xyz, triangles=tri[0], Mtriangles_s=tri[0] / ms_all / 1e3,
                                             stored_coordinates={"state_build_ms": ms_state, "marching_cubes_ms": ms_mc, "level_by_level_ms": ms_lev},
                                             workload="isosurface.cpp:1434-1592 for the hierarchy: ghost fill of the state (pa_fill_ghosts_hierarchy: FillBoundary + FillPatchTwoLevels "
                                                      "with PCInterp, one launch each) + pa_mc_hierarchy_xyz (marching cubes on 3 levels in one call, finer level as mask, vertex coordinates "
                                                      "from cell indices), base 256^3, 64^3 boxes, T = 1150 isotherm; ms = BOTH, incl. the count read-back, the pooled output block and the "
                                                      "final sync; stored_coordinates = the round-4 path (3 coordinate components per cell written, FillBoundaried and FillPatched: 12 "
                                                      "launches; pa_mc_hierarchy_fine; level_by_level_ms = one pa_mc_level_fine call per level)")
    if want("c4_isosurface_base256"):
        c4_case()

    # ------------------------------------------------------------------ options, smoothing, distance function, streamlines
    def curv_options_case():
        # curvature.cpp:575-789 on the headline hierarchy: pa_curvature_run with do_gaussCurv + do_strain + do_velnormal (pass-by-pass
        # kernels; quirk Q3 kept).  Algorithmic bytes: read the progress source + 3 velocity components (32 B), write Progress,
        # MeanCurvature, FlameNormal x3, GaussianCurvature, StrainRate, VelFlameNormal (64 B) = 96 B/cell
        from peleanalysis_amd.hierarchy import retile_hierarchy
        Hf = nested_hierarchy(512, 3, 128, is_per=(1, 1, 0))
        H = retile_hierarchy(Hf) if retile else Hf
        dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
        ins = [alloc(lv, dl, 4, 2, "flame", 401 + l) for l, (lv, dl) in enumerate(zip(H.levels, dls))]
        ous = [alloc(lv, dl, 18, 0) for lv, dl in zip(H.levels, dls)]
        stream.synchronize()
        bc = capi.bc_from_flags((1, 1, 0))
        cells = sum(lv.ncells for lv in H.levels)
        # fast path (round 5, second session): Progress / K / N from the exact-normal pipeline whose sweeps store G = the gradient of c instead
        # of grad phi (72 B/cell), then ONE options pass per level (G, u with their stencils, N, c in; Kg, SR, Vn out); the pass-by-pass
        # kernels (fused = 0) on the same inputs beside it, all 8 output fields compared on the device bit for bit
        st_, ou_ = [a[1] for a in ins], [o[1] for o in ous]
        Pf = capi.curv_params(prog_min=300.0, prog_max=2600.0, threshold=None, fused=True, do_gauss=True, do_strain=True, do_velnormal=True, vel_comp=1)
        Pp = capi.curv_params(prog_min=300.0, prog_max=2600.0, threshold=None, fused=False, do_gauss=True, do_strain=True, do_velnormal=True, vel_comp=1)
        ms = timed(lambda: capi.curvature_run(ctx, st_, 0, bc, Pf, ou_, 0), reps=6)
        kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
        assert ctx.bc_errors() == 0
        fast_out = [o[0].clone() for o in ous]
        ms_pp = min(timed(lambda: capi.curvature_run(ctx, st_, 0, bc, Pp, ou_, 0), reps=1) for _ in range(2))
        assert ctx.bc_errors() == 0
        ndiff = 0
        for (lv, a_, o_) in zip(H.levels, fast_out, ous):
            off, cs, _ = mf_layout(lv.boxes, 18, 0)
            for b_ in range(lv.nboxes):
                n = int(np.prod(lv.box_shape(b_, 0)))
                for c_ in range(8):
                    lo_ = int(off[b_] + c_ * cs[b_])
                    ndiff += int((a_[lo_:lo_ + n].view(torch.int64) != o_[0][lo_:lo_ + n].view(torch.int64)).sum().item())
        del fast_out
        out["f1_curvature_options_headline"] = entry(ms, cells, 96, pass_by_pass_ms=ms_pp, pass_by_pass_frac_hbm=cells * 96 / (ms_pp * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                     fast_vs_pass_by_pass_values_differing=ndiff, sweep_kernel=kn,
                                                     workload="pa_curvature_run (curvature.cpp:283-789) with do_gaussCurv + do_strain + do_velnormal, 3-level base 512^3, "
                                                     "128^3 boxes in the file, progress source + 3 velocity components in, 8 fields out; " + tiling_txt(Hf, H) +
                                                     "; ms = the fast path (G-output sweeps + one options pass per level), pass_by_pass_ms = fused=0, the smaller of two "
                                                     "single passes (one kernel per AMReX call of the reference)")
        if not want("f1_do_smooth_headline"):
            del ins, ous, dls
            torch.cuda.empty_cache()
            return
        # curvature.cpp:328-406: the implicit smoothing solve alone (BiCGStab on the composite operator), same hierarchy, one rank
        c = [alloc(lv, dl, 1, 1) for lv, dl in zip(H.levels, dls)]
        sol = [alloc(lv, dl, 1, 1) for lv, dl in zip(H.levels, dls)]
        for a, ci in zip(ins, c):  # progress variable in [0, 1] as the right-hand side
            ctx.check(ctx.lib.pa_progress_level(ctx.h, a[1].h, 0, 300.0, 2003.0, ci[1].h, 0, 0))
        ctx.sync()
        sm = {}
        hr = (C.c_void_p * 3)(*[x[1].h for x in c])
        hs = (C.c_void_p * 3)(*[x[1].h for x in sol])
        # the reference's default smoothing_time on this unit-cube test domain (dt / dx^2 = 0.42 on the finest level: plain BiCGStab), 1e-5
        # (dt / dx^2 = 42) with and without the V-cycle preconditioner, and 6e-5 (dt / dx^2 = 250: what 1e-7 is for a plotfile in physical units)
        for dt_s, mg_env in ((1e-7, None), (1e-5, None), (1e-5, "0"), (6e-5, None)):
            it_, rs_ = C.c_int(0), C.c_double(0.0)
            if mg_env is not None:
                os.environ["PA_SMOOTH_MG"] = mg_env
                ctx.lib.pa_options_reload()
            for _ in range(2 if dt_s < 1e-6 else 1):
                t0 = time.perf_counter()
                rc_ = ctx.lib.pa_smooth_solve(ctx.h, 3, hr, 0, hs, 0, dt_s, (C.c_int32 * 3)(*bc), 1e-12, 600, C.byref(it_), C.byref(rs_))
                msq = (time.perf_counter() - t0) * 1e3
            os.environ.pop("PA_SMOOTH_MG", None)
            ctx.lib.pa_options_reload()
            sm[f"smoothing_time_{dt_s:g}" + ("_unpreconditioned" if mg_env == "0" else "")] = {
                "iterations": it_.value, "rel_residual": rs_.value, "converged_to_1e-12": rc_ == 0, "ms": msq, "ms_per_iteration": msq / max(it_.value, 1),
                "solver": "BiCGStab" if (mg_env == "0" or dt_s < 4e-6) else "BiCGStab + V(2,4) multigrid preconditioner (damped Jacobi, AMR levels + coarsened copies of level 0)"}
        out["f1_do_smooth_headline"] = dict(sm, cells=cells, workload="pa_smooth_solve (curvature.cpp:328-406: (I - dt Lap) c~ = c, composite over the levels, tol 1e-12), 3-level base "
                                            "512^3, 1 rank; ms = one whole solve incl. its allocations; per iteration 2 operator applications + 5 dot products (+ 2 V-cycles where the finest "
                                            "level's dt / dx^2 > 8); " + tiling_txt(Hf, H))
        del ins, ous, c, sol, dls
        torch.cuda.empty_cache()

    if want("f1_curvature_options_headline") or want("f1_do_smooth_headline"):
        curv_options_case()

    def sdf_case():
        # isosurface.cpp:1595-1655 (build_distance_function): make_level_set3 on 130^3 grids (a 128^3 FAB + 1 ghost layer), a sphere of
        # 32768 triangles through the grid; 1 grid and a batch of 16
        v = [np.array(p_, float) for p_ in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))]
        f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
        for _ in range(6):
            mid, nf = {}, []

            def m(a, b):
                k = (min(a, b), max(a, b))
                if k not in mid:
                    q = v[a] + v[b]
                    v.append(q / np.linalg.norm(q))
                    mid[k] = len(v) - 1
                return mid[k]
            for a, b, c_ in f:
                ab, bc_, ca = m(a, b), m(b, c_), m(c_, a)
                nf += [(a, ab, ca), (ab, b, bc_), (ca, bc_, c_), (ab, bc_, ca)]
            f = nf
        verts = (0.5 + 0.31 * np.array(v)).astype(np.float32)
        tris = np.array(f, dtype=np.uint32)
        g = 130
        tt = torch.from_numpy(tris.astype(np.int32)).to(dev)
        tx = torch.from_numpy(verts).to(dev)
        nb = 16
        phis = [torch.empty(g ** 3, dtype=torch.float32, device=dev) for _ in range(nb)]
        grids = (capi.PaSdfGrid * nb)()
        for q, gr in enumerate(grids):
            gr.ntri, gr.tri, gr.nvert, gr.x = len(tris), tt.data_ptr(), len(verts), tx.data_ptr()
            for d in range(3):
                gr.origin[d] = np.float32(0.0)
                gr.n[d] = g
            gr.dx = np.float32(1.0 / g)
            gr.phi = phis[q].data_ptr()
        torch.cuda.synchronize()
        res = {}
        for n_ in (1, 4, nb):
            ctx.check(ctx.lib.pa_sdf_level_set3(ctx.h, n_, grids, 1))
            ctx.sync()
            t0 = time.perf_counter()
            ctx.check(ctx.lib.pa_sdf_level_set3(ctx.h, n_, grids, 1))
            ctx.sync()
            ms_ = (time.perf_counter() - t0) * 1e3
            res[f"grids_per_call_{n_}"] = {"ms": ms_, "ms_per_grid": ms_ / n_, "Mpoints_s": n_ * g ** 3 / ms_ / 1e3}
        out["f2_distance_function"] = dict(res, triangles=int(len(tris)), grid=f"{g}^3", workload="pa_sdf_level_set3 = make_level_set3 (isosurface.cpp:1625, Tools/SDFGen/makelevelset3.cpp:118-185), "
                                           f"exact band 1, float32, a sphere of {len(tris)} triangles through 130^3 grids; block wavefronts: 16 sweeps x (block planes) dependent launches, each a workgroup "
                                           "per block walking its inner hyperplanes -- a latency chain, no roofline")
        del phis, tt, tx
        torch.cuda.empty_cache()

    if want("f2_distance_function"):
        sdf_case()

    def stream_case():
        # partStream.cpp:121-207 / StreamPC.cpp: RK4 lines through the trilinear interpolant of a velocity field on a 3-level hierarchy
        # (base 128^3, nGrow 3, PCInterp ghost fill), two lines per seed
        H = nested_hierarchy(128, 3, 64, is_per=(0, 0, 0))
        dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
        vf = [alloc(lv, dl, 3, 3, "flame", 611 + l) for l, (lv, dl) in enumerate(zip(H.levels, dls))]
        for l in range(3):
            with torch.cuda.stream(stream):
                vf[l][0].mul_(1.0e-3)  # O(1) velocities
            ctx.check(ctx.lib.pa_fill_boundary(ctx.h, vf[l][1].h, 0, 3, 3))
            if l > 0:
                ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, vf[l][1].h, vf[l - 1][1].h, 0, 3, 3, 2, 0))
        ctx.sync()
        rng = np.random.default_rng(5)
        nseed, nsteps = 50000, 50
        seeds = 0.3 + 0.4 * rng.random((nseed, 3))
        dts = 0.1 / 512.0  # hRK * finest dx
        buf = torch.empty(2 * nseed * nsteps * 3, dtype=torch.float64, device=dev)
        nred = C.c_int32(0)
        hv = (C.c_void_p * 3)(*[x[1].h for x in vf])

        def go():
            ctx.check(ctx.lib.pa_stream_trace(ctx.h, 3, hv, 0, nseed, seeds.ctypes.data_as(C.POINTER(C.c_double)), nsteps, dts, C.c_void_p(buf.data_ptr()), C.byref(nred)))
        go()
        t0 = time.perf_counter()
        go()
        ms_ = (time.perf_counter() - t0) * 1e3
        out["f4_partstream"] = {"ms": ms_, "seeds": nseed, "lines": 2 * nseed, "steps_per_line": nsteps - 1, "Mline_steps_s": 2 * nseed * (nsteps - 1) / ms_ / 1e3, "redistributions": int(nred.value),
                                "workload": "pa_stream_trace (partStream.cpp:121-207, StreamPC.cpp:88-260): RK4 through the trilinear interpolant, 3-level base 128^3, 64^3 boxes, nGrow 3, "
                                            "50000 random seeds x 2 lines x 49 steps, hRK 0.1; synchronous call incl. seed upload and flag read-back"}
        del vf, buf, dls
        torch.cuda.empty_cache()

    if want("f4_partstream"):
        stream_case()

    def tools_case():
        # the drop-in binaries end to end on a plotfile of config 3 / 4's size (3 levels, base 256^3, 64^3 boxes, 3 components: 1.2 GB), SECOND run of
        # each in the same directory (tools/tool_e2e.py does this at the headline size: profiles/r05_tool_e2e_512.txt).  Bounded and guarded: whatever
        # goes wrong here (no room in the scratch directory, a missing binary) is reported in the entry, the bench line itself is not at risk
        import shutil
        import tempfile
        ent = {"workload": "grad3d / curvature3d / filterPlt3d / isosurface3d (tools/bin/*.ex) on a synthetic plotfile, 3 levels, base 256^3, 64^3 boxes, 3 components "
                           "(temp, x_velocity, density), periodic x/y: wall seconds of the SECOND run of each tool (process start to exit) and the phases the tool reports"}
        d = None
        try:
            from peleanalysis_amd.hierarchy import MultiFab, field_flame, fill_analytic
            from peleanalysis_amd.plotfile import write_plotfile
            bindir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "bin")
            Ht = nested_hierarchy(256, 3, 64, is_per=(1, 1, 0))
            names = ["temp", "x_velocity", "density"]
            mfs = []
            for lv in Ht.levels:
                m = MultiFab(lv, 3, 0, fill=0.0)
                for c in range(3):
                    fill_analytic(m, c, (lambda x, y, z, c=c: field_flame(x, y, z, c)))
                mfs.append(m)
            d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
            pth = os.path.join(d, "plt00000")
            write_plotfile(pth, Ht, mfs, names, time=0.0, level_steps=[0, 0, 0])
            del mfs
            ent["cells"] = sum(lv.ncells for lv in Ht.levels)
            runs = (("grad3d.ex", ["gradVar=temp", "is_per=1 1 0"]), ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0"]),
                    ("filterPlt3d.ex", ["is_per=1 1 0"]), ("isosurface3d.ex", ["isoCompName=temp", "isoVal=1150", "comps=0 1 2"]))
            for tool, targs in runs:
                wall, js = None, None
                for rep in range(2):
                    t0 = time.perf_counter()
                    r = subprocess.run([os.path.join(bindir, tool), "infile=" + pth, "bench_json=1"] + targs, cwd=d, capture_output=True, text=True, timeout=120)
                    wall = time.perf_counter() - t0
                    if r.returncode != 0:
                        raise RuntimeError(tool + ": " + r.stderr[-300:])
                    js = [ln for ln in r.stdout.splitlines() if ln.startswith('{"tool"')]
                ent[tool] = {"wall_s": round(wall, 3), "phases_s": (json.loads(js[-1]).get("phases_s") if js else None)}
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the bench line
            ent["error"] = repr(e)[:400]
        finally:
            if d:
                shutil.rmtree(d, ignore_errors=True)
        out["tools_end_to_end_base256"] = ent

    if want("tools_end_to_end_base256"):
        tools_case()
    return out


def dry_run(args, rank, world):
    """PA_BENCH_REHEARSE=dry (see main): what the driver's multi-GPU command does before any rank touches its card -- the
    launcher's environment, the gloo rendezvous, the hierarchy, the internal tiling per rank count and the shard every rank
    would build, one collective of each kind the control plane uses -- without a GPU.  Not a measurement."""
    import torch.distributed as dist
    from peleanalysis_amd import dist as padist
    from peleanalysis_amd.hierarchy import nested_hierarchy, retile_hierarchy
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    per = tuple(int(v) for v in args.per.split())
    Hfile = nested_hierarchy(args.base, args.nlev, args.box, is_per=per)
    H = retile_hierarchy(Hfile, nranks=world) if args.retile else Hfile
    owners = padist.shard(H, world) if world > 1 else [np.zeros(lv.nboxes, dtype=np.int32) for lv in H.levels]
    mine = [int((np.asarray(o) == rank).sum()) for o in owners]
    cells_mine = 0
    for lv, o in zip(H.levels, owners):
        bx = lv.boxes[np.asarray(o) == rank].astype(np.int64)
        cells_mine += int((bx[:, 3:] - bx[:, :3] + 1).prod(axis=1).sum())
    got = [None] * world
    if world > 1:
        import torch
        dist.all_gather_object(got, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "boxes": mine, "cells": cells_mine})
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == world - 1
        dist.barrier()
    else:
        got = [{"rank": 0, "local_rank": 0, "boxes": mine, "cells": cells_mine}]
    if rank == 0:
        cells = sum(lv.ncells for lv in H.levels)
        assert sorted(g["rank"] for g in got) == list(range(world)) and sum(g["cells"] for g in got) == cells
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "cells": cells,
                          "swept_boxes_per_level": [lv.nboxes for lv in H.levels], "boxes_per_level_by_rank": [g["boxes"] for g in sorted(got, key=lambda g: g["rank"])],
                          "cells_by_rank": [g["cells"] for g in sorted(got, key=lambda g: g["rank"])],
                          "note": "PA_BENCH_REHEARSE=dry: launcher / rendezvous / sharding plumbing only, no GPU, not a measurement"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--base", type=int, default=512, help="base-level cells per direction (headline: 512)")
    ap.add_argument("--nlev", type=int, default=3)
    ap.add_argument("--box", type=int, default=128)
    ap.add_argument("--ncomp", type=int, default=1, help="components pushed through grad->curvature per step")
    ap.add_argument("--nbatch", type=int, default=8, help="components per batch of the boundary kernels (pa_gradcurv_run_comps2; 8 output components per slot)")
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--retile", type=int, default=1, help="1 (default, as the tools): the levels are held and swept on the internal tiling pa_level_retile makes of the "
                                                           "file's boxes (--box); 0: on the file's boxes themselves")
    ap.add_argument("--per", type=str, default="1 1 0", help="periodicity flags x y z (headline: periodic x/y, wall z); single GPU only")
    ap.add_argument("--threshold", type=float, default=-1.0, help="diagnostic: threshold_prog / threshold_value of curvature.cpp:549-570 (< 0: off, the headline)")
    ap.add_argument("--traffic", choices=("live", "file", "none"), default="live",
                    help="roofline.traffic: measured now by two rocprofv3 --pmc child passes (falls back to the committed figure), the committed figure, or null")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (configs 2-4, grad only, small boxes)")
    ap.add_argument("--secondary-only", type=str, default="", help="diagnostic: skip the headline region and the CPU leg, run ONE secondary entry (e.g. irregular_amr)")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: no HIP events in the timed region (no roofline object)")
    ap.add_argument("--cpu-base", type=int, default=0, help="base size of the cpu_baseline sample (0: from the core count)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong", help="N > 1: shard ONE hierarchy (strong) or one hierarchy per GPU (weak)")
    ap.add_argument("--sim-of", type=int, default=0, help="diagnostic, 1 GPU: time rank 0's share of an N-rank strong-scaling run with no-op exchanges")
    ap.add_argument("--xdelay-us", type=float, default=-1.0, help="--sim-of: every exchange costs this many microseconds (+ bytes / --xlink-GBs) on the stream it is issued on "
                                                                  "(pa_ctx_set_delay_comm) instead of nothing: how much of an exchange the schedule hides")
    ap.add_argument("--ab", type=str, default="", help="diagnostic: VAR=A,B -- alternate an environment switch the library reads per pass in blocks of --steps passes inside this "
                                                        "process (8 blocks each), print the two medians and exit (works with --sim-of)")
    ap.add_argument("--xlink-GBs", type=float, default=0.0, help="--sim-of with --xdelay-us: per-peer link bandwidth of the delay model (xGMI: ~153 GB/s nominal); 0: fixed delay only")
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
    # PA_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a box with FEWER GPUs than ranks -- ranks share the cards,
    # transport is the pa_comm callback over gloo through host memory (RCCL wants one GPU per rank).  Same sharding, same
    # region plans, same pack / unpack kernels, same reductions; the number it prints is not a scaling measurement.
    # PA_BENCH_REHEARSE=rccl: the same, but the RCCL bring-up is ATTEMPTED first (it cannot succeed with two ranks on one card):
    # rehearses the degradation to the gloo transport -- by error or, with a small PA_RCCL_TIMEOUT, by time limit.
    # PA_BENCH_REHEARSE=dry: no GPU at all -- launcher, rendezvous, argument plumbing, the hierarchy / re-tiling / sharding every
    # rank would build and the collectives of the control plane, for any N (the CPU tier runs it with N = 8); prints a
    # {"dry_run": true, ...} line that is not a measurement of anything.
    rehearse_mode = os.environ.get("PA_BENCH_REHEARSE", "0")
    if rehearse_mode == "dry":
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    rehearse = rehearse_mode in ("1", "rccl")
    try_rccl = rehearse_mode in ("0", "rccl")
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

    if args.secondary_only:
        t0s = time.perf_counter()
        res = {"secondary_only": args.secondary_only, "secondary": secondary(ctx, torch, stream, dev, only=args.secondary_only, retile=args.retile)}
        res["wall_s"] = round(time.perf_counter() - t0s, 2)
        print(json.dumps(res))
        return
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
    # internal re-tiling (what grad3d / curvature3d do with a plotfile's BoxArray): same cells, merged boxes -- results identical in
    # every cell (tests/test_retile.py), with the tools' limits: pa_hierarchy_retile_limits on one GPU, ..._ranks when sharded (every
    # rank keeps at least four boxes per level).  Weak-scaling copies stay as built (a copy per rank).
    Hfile = H
    if args.retile and not (args.scaling == "weak" and nshard > 1):
        from peleanalysis_amd.hierarchy import retile_hierarchy
        H = retile_hierarchy(Hfile, nranks=nshard)
        owners = padist.shard(H, nshard) if nshard > 1 else [None] * args.nlev
    bc = capi.bc_from_flags(per)
    xch = {"mode": "none"}
    gcomm = None
    helper = None
    if args.sim_of:  # one rank's share of an N-rank run on this GPU, exchanges replaced by no-ops: compute time per rank
        if args.xdelay_us >= 0:
            ctx.check(ctx.lib.pa_ctx_set_delay_comm(ctx.h, nshard, 0, float(args.xdelay_us), float(args.xlink_GBs)))
            xch["mode"] = (f"SIMULATION of rank 0 of {nshard}: every exchange is a spin of {args.xdelay_us:g} us" +
                           (f" + bytes to the busiest peer / {args.xlink_GBs:g} GB/s" if args.xlink_GBs > 0 else "") + " on its stream (delay-model transport; results wrong in ghost cells)")
        else:
            nullx = capi.EXCHANGE_FN(lambda user, st, n, x: 0)
            nullr = capi.ALLREDUCE_FN(lambda user, v, n, op: 0)
            simc = capi.PaComm(None, 0, nshard, nullx, nullr)
            ctx.set_comm(simc)
            xch["mode"] = f"SIMULATION of rank 0 of {nshard}: exchanges are no-ops (results wrong in ghost cells, timing = compute only)"
    elif world > 1:
        err = ""
        helper = None
        if not try_rccl:
            err = "PA_BENCH_REHEARSE=1: ranks share the cards, RCCL not attempted"
        else:
            # built-in transport: grouped ncclSend / ncclRecv on the library's stream; a ring exchange + a reduction with known
            # answers before it is trusted.  Bring-up runs on a helper thread under a time limit: a communicator that never
            # forms (ncclCommInitRank blocks until every rank has joined) must degrade to the gloo transport, not eat the run.
            import threading
            box = {}
            # the id broadcast is a torch.distributed collective: it runs HERE, on the thread that issues all the others, so that
            # the order of collectives is the same on every rank whatever the helper thread does (advisor finding, round 3)
            try:
                uid = padist.broadcast_rccl_id(ctx)
            except Exception as e:
                uid, box["err"] = None, repr(e)[:300]

            def bring_up():
                if uid is None:
                    return
                try:
                    ctx.init_rccl(world, rank, uid)  # ncclCommInitRank: blocks until every rank has joined
                    ctx.comm_selftest(1 << 16)
                    box["ok"] = True
                except Exception as e:
                    box["err"] = repr(e)[:300]

            limit = float(os.environ.get("PA_RCCL_TIMEOUT", "120"))
            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(limit)
            helper = th
            if th.is_alive():
                err = f"RCCL bring-up did not finish within {limit:.0f} s (ncclCommInitRank / self-test)"
                # the helper may still be inside the library with this context: leave it behind, continue on a fresh one
                ctx = capi.Context(local, stream.cuda_stream)
            else:
                err = box.get("err", "")
        ok = [None] * world
        dist.all_gather_object(ok, err)
        if any(ok):
            gcomm = padist.GlooComm(ctx)    # pa_comm callbacks: packed buffers staged through host memory + gloo
            ctx.comm_selftest(1 << 12)
            xch["mode"] = "host-staged gloo point-to-point (pa_comm callbacks)" + ("" if not try_rccl else " -- RCCL transport failed: " + next(e for e in ok if e))
        else:
            xch["mode"] = "RCCL point-to-point (grouped ncclSend/ncclRecv issued by the library on its stream): exchange A (ghost cells of phi + coarse phi, all components) once per step, exchange B (coarse normals) once per batch of components"
    dls = [capi.DevLevel(ctx, lv, owners[l], myrank, nshard) if nshard > 1 else capi.DevLevel(ctx, lv) for l, lv in enumerate(H.levels)]
    cells = sum(lv.ncells for lv in H.levels)                # the whole job
    cells_local = sum(dl.level.ncells for dl in dls)         # this rank's share
    hold, states, works, outs = [], [], [], []
    nslot = max(1, min(args.nbatch, args.ncomp, 16))  # output slots: the boundary kernels run once per batch of nslot components
    with torch.cuda.stream(stream):
        for li, dl in enumerate(dls):
            lv = dl.level
            off, cs, tot = mf_layout(lv.boxes, args.ncomp, 2)
            tin = torch.zeros(max(tot, 1), dtype=torch.float64, device=dev)
            fill_level_on_device(torch, lv, tin, args.ncomp, 2, off, cs, dev, 1234 + 100 * rank + li)
            _, _, tw = mf_layout(lv.boxes, 1, 2)
            _, _, to = mf_layout(lv.boxes, 8 * nslot, 0)
            twk = torch.zeros(max(tw, 1), dtype=torch.float64, device=dev)
            tout = torch.zeros(max(to, 1), dtype=torch.float64, device=dev)
            hold += [tin, twk, tout]
            states.append(capi.DevMF(ctx, dl, args.ncomp, 2, tin.data_ptr()))
            works.append(capi.DevMF(ctx, dl, 1, 2, twk.data_ptr()))
            outs.append(capi.DevMF(ctx, dl, 8 * nslot, 0, tout.data_ptr()))
    stream.synchronize()
    params = capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=(args.threshold if args.threshold >= 0 else None), fused=bool(args.fused))

    def step():
        # every component through the pipeline into recycled output buffers (SURVEY 8d memory budget); cross-rank ghost fills
        # happen inside; result-independent ghost fills are done once for all components
        capi.gradcurv_run_comps2(ctx, states, 0, args.ncomp, bc, params, works, outs, 0, nslot)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if args.ab:
        import statistics
        var, vals = args.ab.split("=")
        va, vb = vals.split(",")
        tm = {va: [], vb: []}
        for blk in range(16):
            v = (va, vb)[blk & 1]
            os.environ[var] = v
            ctx.lib.pa_options_reload()  # the library reads its switches once: re-read after the flip
            step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            tm[v].append((time.perf_counter() - t0) / args.steps * 1e3)
        if rank == 0:
            print(json.dumps({"ab": var, "median_ms": {k: statistics.median(v) for k, v in tm.items()}, "blocks_ms": {k: [round(x, 4) for x in v] for k, v in tm.items()},
                              "delta_ms_B_minus_A": statistics.median(tm[vb]) - statistics.median(tm[va])}))
        return
    ctx.profile_read(1, reset=True)
    ctx.profile_enable(0 if args.no_profile else (1 << 1))  # timed region: HIP events around the dominant kernel only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0  # the host's share: every pass enqueued (nothing in a pass waits for the GPU)
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
        "ms_per_step": dt / args.steps * 1e3, "host_enqueue_ms_per_step": t_enq / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"fused grad->curvature, {args.nlev}-level AMR, base {args.base}^3{' per GPU' if weak else ''}, ref_ratio 2, {args.box}^3 boxes "
                               f"({Hfile.levels[0].nboxes} per level) in the file" +
                               (f", swept on the internal tiling pa_level_retile makes of them ({', '.join(str(lv.nboxes) for lv in H.levels)} boxes per level, "
                                f"largest {'x'.join(str(int(v)) for v in (H.levels[0].boxes[:, 3:] - H.levels[0].boxes[:, :3] + 1).max(axis=0))})" if H is not Hfile else ", swept as they are (--retile 0)") +
                               f", {args.ncomp} comp(s), {per_txt}, {cells} cells in the job",
                   "tiling": {"file_boxes_per_level": [lv.nboxes for lv in Hfile.levels], "swept_boxes_per_level": [lv.nboxes for lv in H.levels], "retile": bool(H is not Hfile)},
                   "cells": cells, "cells_this_rank": cells_local, "ncomp": args.ncomp, "components_per_batch": nslot, "fused": bool(args.fused),
                   "parallelism": par, "exchange": xch},
    }
    if args.sim_of and args.xdelay_us >= 0:
        import ctypes as C_
        nc_, us_ = C_.c_int64(), C_.c_double()
        if ctx.lib.pa_delay_comm_stats(ctx.h, C_.byref(nc_), C_.byref(us_)) == 0 and nc_.value:
            res["delay_model"] = {"exchanges": nc_.value, "modelled_us_per_exchange": us_.value / nc_.value,
                                  "modelled_us_per_step": us_.value / max(1, args.steps + args.warmup + 2)}
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
        # HBM bytes per launch from this round's rocprofv3 PMC passes of the same workload (profiles/, tools/prof.sh bench): only
        # valid for the kernel variant it was measured on -- null when the library launched another one
        traffic, kern = None, ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
        traffic_src = None
        headline_cfg = (args.base, args.nlev, args.box, args.ncomp, world, args.sim_of) == (512, 3, 128, 1, 1, 0) and per == (1, 1, 0) and args.threshold < 0
        if headline_cfg and args.traffic == "live" and rank == 0:
            traffic, traffic_src = live_traffic(retile=args.retile)
        tj = os.path.join(ROOT, "profiles", "r06_headline_traffic.json" if args.retile else "r04_headline_traffic.json")
        if traffic is None and args.traffic != "none" and os.path.exists(tj) and headline_cfg:
            rec = json.load(open(tj))
            if rec.get("kernel") == kern:
                traffic = rec.get("traffic_bytes_per_launch")
                traffic_src = "profiles/" + os.path.basename(tj) + " (PMC passes of the same workload and kernel variant, tools/prof.sh bench)" + (f"; live pass: {traffic_src}" if traffic_src else "")
        res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": traffic_src, "kernel": kern + " (fused grad->curvature sweep)", "avg_launch_ms": avg_ms,
                           "launches": nk, "bytes_per_cell": BYTES_PER_CELL, "cells_per_launch": cells_local * args.ncomp * args.steps / nk}
        res["breakdown_ms_per_step"] = bd  # from the untimed steps after the timed region (rank 0)
        res["step_frac_of_hbm_roofline"] = (cells_local * args.ncomp * BYTES_PER_CELL / (dt / args.steps) / 1e9) / HBM_PEAK_GBS
    if rank == 0 and world == 1 and not args.sim_of and not args.no_secondary:
        try:
            del hold, states, works, outs  # the headline buffers (33 GB) make room for the secondary workloads
            torch.cuda.empty_cache()
            t0s = time.perf_counter()
            res["secondary"] = secondary(ctx, torch, stream, dev, retile=args.retile)
            res["secondary"]["wall_s_incl_data_generation"] = round(time.perf_counter() - t0s, 2)
        except Exception as e:  # the headline line must survive a failure here
            res["secondary"] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_cpu and not args.sim_of:
        try:
            # bounded sample (~10-30 s of CPU work): a 3-level hierarchy sized from the host core count
            cores = usable_cpus()
            base = args.cpu_base or (384 if cores >= 16 else (256 if cores >= 8 else 128))
            res["cpu_baseline"], sample = cpu_baseline(base, args.nlev, min(args.box, base // 2))
        except Exception as e:  # the baseline is reported, never required for the GPU number
            res["cpu_baseline"], sample = {"error": repr(e)}, None
        if sample is not None:
            try:
                torch.cuda.empty_cache()
                res["parity_check"] = parity_check(ctx, sample, retile=args.retile)
            except Exception as e:
                res["parity_check"] = {"error": repr(e)[:300]}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if helper is not None and helper.is_alive():
            # an RCCL bring-up that never returned is still inside the library on its abandoned context: no interpreter / runtime
            # teardown under it -- the line is out, leave at once
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)


if __name__ == "__main__":
    main()
