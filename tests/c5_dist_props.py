"""BASELINE config 5 at full size, as far as one GPU allows: the PeleLMeX-style hierarchy (4 levels of 256^3 cells, 64^3
boxes, 55 components = 53 species + T + rho) through the fused grad->curvature pipeline, (a) undistributed and (b) with
the BoxArray of every level sharded over 4 ranks (Morton order + equal-volume cuts) that SHARE the GPU -- the library's
plans, pack / unpack kernels, coarse-source copies and the multi-component entry point (exchange A once for all 55
components, one exchange B per batch of 8 components: pa_gradcurv_run_comps2) are exactly what 8 GPUs run; only the transport is the gloo callback instead
of RCCL.  Property: for every box and component the checksum (wrap-around sum of the bit patterns of its 8 result
fields) of the sharded run equals the undistributed one -- bit-identical results, 1.5e10 values.

usage: python tests/c5_dist_props.py [ncomp=55] [world=4]        (prints "c5 dist properties OK")"""
import os
import socket
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def field(torch, lv, b, ng, c, dev):
    """deterministic synthetic species-like field of GLOBAL box b (the same bits whoever owns the box)"""
    lo = lv.boxes[b, :3]
    nz, ny, nx = lv.box_shape(b, ng)
    dx = lv.dx
    i = torch.arange(int(lo[0]) - ng, int(lo[0]) - ng + nx, device=dev, dtype=torch.float64)
    j = torch.arange(int(lo[1]) - ng, int(lo[1]) - ng + ny, device=dev, dtype=torch.float64)
    k = torch.arange(int(lo[2]) - ng, int(lo[2]) - ng + nz, device=dev, dtype=torch.float64)
    X, Y, Z = ((i + 0.5) * dx[0])[None, None, :], ((j + 0.5) * dx[1])[None, :, None], ((k + 0.5) * dx[2])[:, None, None]
    r = torch.sqrt(((X - 0.5) / 0.30) ** 2 + ((Y - 0.5) / 0.15) ** 2 + ((Z - 0.5) / 0.18) ** 2)
    base = (1.0 + 0.02 * c) * (300.0 + 850.0 * (1.0 + torch.tanh((r - 1.0) / 0.08))) + 3.0 * torch.sin(2 * np.pi * (X + 0.37 * c)) * torch.cos(2 * np.pi * Y)
    noise = torch.frac(torch.sin(i[None, None, :] * 12.9898 + j[None, :, None] * 78.233 + k[:, None, None] * 37.719 + 1.7 * c) * 43758.5453)
    return (base + 1e-3 * noise).expand(nz, ny, nx)


def run(rank, world, port, ncomp, outdir):
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    from peleanalysis_amd import capi
    from peleanalysis_amd import dist as padist
    from peleanalysis_amd.hierarchy import mf_layout, nested_hierarchy
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = capi.Context(0)
    nbatch = 8 if world > 1 else 1
    comm = padist.GlooComm(ctx) if world > 1 else None
    H = nested_hierarchy(256, 4, 64, is_per=(1, 1, 0))
    owners = padist.shard(H, world) if world > 1 else [None] * 4
    dls = [capi.DevLevel(ctx, lv, owners[l], rank, world) if world > 1 else capi.DevLevel(ctx, lv) for l, lv in enumerate(H.levels)]
    hold, states, works, outs = [], [], [], []
    for l, dl in enumerate(dls):
        lv = dl.level
        off, cs, tot = mf_layout(lv.boxes, ncomp, 2)
        t = torch.zeros(max(tot, 1), dtype=torch.float64, device=dev)
        for i, g in enumerate(dl.gids):
            nz, ny, nx = lv.box_shape(i, 2)
            for c in range(ncomp):
                t[off[i] + c * cs[i]: off[i] + c * cs[i] + nz * ny * nx] = field(torch, H.levels[l], int(g), 2, c, dev).reshape(-1)
        _, _, tw = mf_layout(lv.boxes, 1, 2)
        _, _, to = mf_layout(lv.boxes, 8 * nbatch, 0)
        w, o = torch.zeros(max(tw, 1), dtype=torch.float64, device=dev), torch.zeros(max(to, 1), dtype=torch.float64, device=dev)
        hold += [t, w, o]
        states.append(capi.DevMF(ctx, dl, ncomp, 2, t.data_ptr()))
        works.append(capi.DevMF(ctx, dl, 1, 2, w.data_ptr()))
        outs.append((capi.DevMF(ctx, dl, 8 * nbatch, 0, o.data_ptr()), o))
    torch.cuda.synchronize()
    sums = {}

    def done(c, oc):
        ctx.sync()
        for l, dl in enumerate(dls):
            off8, cs8, _ = mf_layout(dl.level.boxes, 8 * nbatch, 0)
            o = outs[l][1]
            for i, g in enumerate(dl.gids):
                sums[(l, int(g), c)] = int(o[off8[i] + oc * cs8[i]: off8[i] + (oc + 8) * cs8[i]].view(torch.int64).sum().item())
    params = capi.curv_params(prog_min=200.0, prog_max=5000.0, fused=True)
    # the undistributed reference one component at a time, the sharded run in batches of 8 (pa_gradcurv_run_comps2: boundary
    # kernels and exchange B once per batch)
    capi.gradcurv_run_comps2(ctx, states, 0, ncomp, capi.bc_from_flags((1, 1, 0)), params, works, [o[0] for o in outs], 0, nbatch, done)
    ctx.sync()
    assert ctx.bc_errors() == 0
    kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
    assert kn.endswith("CG=1>") or "_levels<" in kn, kn  # the exact-normal sweep, level by level or all levels in one launch
    if comm is not None:
        assert comm.nexchange == 1 + (ncomp + nbatch - 1) // nbatch, (comm.nexchange, ncomp)  # exchange A once for all components + one exchange B per batch
    keys = np.array(sorted(sums), dtype=np.int64).reshape(-1, 3)
    np.savez(os.path.join(outdir, f"sums_w{world}_r{rank}.npz"), keys=keys, vals=np.array([sums[tuple(k)] for k in keys], dtype=np.int64))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def main():
    ncomp = int(sys.argv[1]) if len(sys.argv) > 1 else 55
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    import torch.multiprocessing as mp
    d = tempfile.mkdtemp()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(run, args=(1, port, ncomp, d), nprocs=1, join=True)       # undistributed, in its own process (its memory is gone afterwards)
    mp.spawn(run, args=(world, port, ncomp, d), nprocs=world, join=True)
    ref = np.load(os.path.join(d, "sums_w1_r0.npz"))
    want = {tuple(k): v for k, v in zip(ref["keys"], ref["vals"])}
    assert len(want) == 4 * 64 * ncomp
    seen = 0
    for r in range(world):
        got = np.load(os.path.join(d, f"sums_w{world}_r{r}.npz"))
        for k, v in zip(got["keys"], got["vals"]):
            assert want[tuple(k)] == v, f"rank {r}: level {k[0]} box {k[1]} component {k[2]} differs from the undistributed run"
            seen += 1
    assert seen == len(want), (seen, len(want))
    print(f"c5 dist properties OK: {len(want)} box-component checksums of {world} ranks equal the undistributed run")


if __name__ == "__main__":
    main()
