"""(run as a child process by tests/test_gpu_fullsize.py -- torch has to initialise HIP before the library does)
GPU parity at BASELINE.json's FULL size (3-level AMR, base 512^3, 128^3 boxes, 4.0e8 cells), where the CPU oracle
would take minutes: size-independent properties of the path instead, every comparison bit for bit on the device.

  1. two independent kernel sets agree: the fused sweep + face fix-up (the headline path) against the pass-by-pass
     kernels (k_progress / k_normal / k_div / applyBC per pass; each parity-tested against the oracle at small sizes),
     and its gradient components against the gradient tool's own kernel (k_grad_march);
  2. exact homogeneity: phi -> 2 phi (a power of two: every operation of the path commutes with it exactly) doubles
     the gradient and its magnitude bit for bit and leaves Progress' normalisation, hence N and K, unchanged;
  3. determinism: a second run reproduces every output bit (no atomics / order dependence on the path);
  4. |N| = 1 wherever the gradient is not floored, K finite (catches a wrong-but-consistent pipeline);
  5. box filter on a 512^3 level (fgr = 4, 125 taps): the streaming kernel and the LDS-tile kernel (two independent
     implementations of the reference tap order) agree bit for bit on random data, and a constant field is
     reproduced exactly (the box weights are dyadic and sum to one without rounding);
  6. marching cubes on a 512^3 level (64 FABs of 130^3): the level-batched pass reproduces the per-FAB entry points
     bit for bit on FABs that hold surface, per-FAB counts add up, connectivity stays inside each FAB's vertex range.
  9. 32^3 boxes (512 per level, base 256^3): the narrow exact-normal sweep + one-layer fix-up == the pass-by-pass kernels ==
     the first fused pipeline, with and without the threshold clip.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # torch first: one HIP runtime in the process
    import bench
    from peleanalysis_amd import capi
    from peleanalysis_amd.hierarchy import mf_layout, nested_hierarchy

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = capi.Context(0)
    base, nlev, box = 512, 3, 128
    H = nested_hierarchy(base, nlev, box, is_per=(1, 1, 0))
    assert sum(lv.ncells for lv in H.levels) == 3 * 512 ** 3
    bc = capi.bc_from_flags((1, 1, 0))
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    def mfs(ncomp, ng):
        """(torch tensors, DevMF views of them) of one multifab per level; the tensors must outlive the views"""
        ts, ms = [], []
        for lv, dl in zip(H.levels, dls):
            _, _, tot = mf_layout(lv.boxes, ncomp, ng)
            t = torch.zeros(tot, dtype=torch.float64, device=dev)
            ts.append(t)
            ms.append(capi.DevMF(ctx, dl, ncomp, ng, t.data_ptr()))
        return ts, ms

    tins, states = mfs(1, 2)
    for li, lv in enumerate(H.levels):
        off, cs, _ = mf_layout(lv.boxes, 1, 2)
        bench.fill_level_on_device(torch, lv, tins[li], 1, 2, off, cs, dev, 4321 + li)
    pristine = [t.clone() for t in tins]
    twork, works = mfs(1, 2)

    def run(fused, thr, scale=1.0):
        for t, p in zip(tins, pristine):
            t.copy_(p * scale)
        touts, outs = mfs(8, 0)
        torch.cuda.synchronize()  # torch's stream -> the library's stream
        capi.gradcurv_run(ctx, states, 0, bc, capi.curv_params(prog_min=300.0 * scale, prog_max=2000.0 * scale, threshold=thr, fused=fused), works, outs, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        return touts, outs

    def same_bits(a, b):
        return bool(torch.equal(a.view(torch.int64), b.view(torch.int64)))

    keep = None
    for thr in (None, 0.02):
        fused, _f = run(True, thr)
        again, _a = run(True, thr)
        for l in range(nlev):
            assert same_bits(fused[l], again[l]), f"level {l}: the fused path is not deterministic"
        del again, _a
        passes, _p = run(False, thr)
        for l in range(nlev):
            assert same_bits(fused[l], passes[l]), f"level {l} (threshold {thr}): fused sweep + face fix-up differs from the pass-by-pass kernels"
        del passes, _p
        if thr is None:
            keep = fused
        del fused, _f
    # gradient tool's kernel against the fused sweep's gradient components
    for t, p in zip(tins, pristine):
        t.copy_(p)
    tg, gouts = mfs(4, 0)
    torch.cuda.synchronize()
    capi.grad_run(ctx, states, 0, bc, gouts, 0)
    ctx.sync()
    for l, lv in enumerate(H.levels):
        off8, cs8, _ = mf_layout(lv.boxes, 8, 0)
        off4, cs4, _ = mf_layout(lv.boxes, 4, 0)
        for b in (0, lv.nboxes // 2, lv.nboxes - 1):
            n = int(np.prod(lv.box_shape(b, 0)))
            for c in range(4):
                assert same_bits(keep[l][off8[b] + c * cs8[b]: off8[b] + c * cs8[b] + n], tg[l][off4[b] + c * cs4[b]: off4[b] + c * cs4[b] + n]), (l, b, c)
    # homogeneity: 2 phi
    t2, _o2 = run(True, None, scale=2.0)
    for l, lv in enumerate(H.levels):
        off8, cs8, _ = mf_layout(lv.boxes, 8, 0)
        for b in (0, lv.nboxes // 3, lv.nboxes - 1):
            n = int(np.prod(lv.box_shape(b, 0)))
            for c in range(8):
                a = keep[l][off8[b] + c * cs8[b]: off8[b] + c * cs8[b] + n]
                d = t2[l][off8[b] + c * cs8[b]: off8[b] + c * cs8[b] + n]
                assert same_bits(a * 2.0 if c < 4 else a, d), f"level {l} box {b} comp {c}: not exactly homogeneous"
    # analytic sanity on the finest level: |N| = 1 wherever the gradient is not floored; K finite
    lv = H.levels[-1]
    off8, cs8, _ = mf_layout(lv.boxes, 8, 0)
    b = lv.nboxes // 2
    n = int(np.prod(lv.box_shape(b, 0)))
    N = [keep[-1][off8[b] + c * cs8[b]: off8[b] + c * cs8[b] + n] for c in (4, 5, 6)]
    gm = keep[-1][off8[b] + 3 * cs8[b]: off8[b] + 3 * cs8[b] + n]
    nn = torch.sqrt(N[0] ** 2 + N[1] ** 2 + N[2] ** 2)
    sel = gm > 1e-6
    assert bool(sel.any()) and float((nn[sel] - 1.0).abs().max()) < 1e-12
    K = keep[-1][off8[b] + 7 * cs8[b]: off8[b] + 7 * cs8[b] + n]
    assert bool(torch.isfinite(K).all())
    del keep, t2, _o2, tg, gouts, pristine
    torch.cuda.empty_cache()

    # ---- 5. box filter, 512^3 level of 128^3 boxes, fgr = 4
    import ctypes as C
    from peleanalysis_amd.hierarchy import Level, chop_box
    n = 512
    lv = Level(chop_box((0, 0, 0), (n - 1,) * 3, 128), (0, 0, 0), (n - 1,) * 3, (1, 1, 1), (0, 0, 0), (1, 1, 1))
    dl = capi.DevLevel(ctx, lv)
    w = (C.c_double * 6)()
    ng = ctx.lib.pa_box_filter_weights(4, w)
    assert ng == 2

    def one(ncomp, g, fill):
        _, _, tot = mf_layout(lv.boxes, ncomp, g)
        t = fill(tot)
        return t, capi.DevMF(ctx, dl, ncomp, g, t.data_ptr())

    tin, fin = one(1, ng, lambda tot: 300.0 + 1700.0 * torch.rand(tot, dtype=torch.float64, device=dev))
    outs_f = []
    os.environ["PA_FILTER_EXACT"] = "1"  # the tap-order kernel: the reference's sum, bit for bit (its result is the yardstick below)
    capi.reload_options()
    to, fo = one(1, 0, lambda tot: torch.zeros(tot, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    ctx.check(ctx.lib.pa_fill_boundary(ctx.h, fin.h, 0, 1, ng))
    ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, fin.h, fo.h, 0, 1, ng, w))
    ctx.sync()
    outs_f.append((to, fo))
    os.environ.pop("PA_FILTER_EXACT")
    capi.reload_options()
    # the separable default against the tap-order sum: SURVEY 8(d) metric, <= 1e-12 * Linf
    ts, fs = one(1, 0, lambda tot: torch.zeros(tot, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, fin.h, fs.h, 0, 1, ng, w))
    ctx.sync()
    dsep = float((ts - outs_f[0][0]).abs().max()) / float(outs_f[0][0].abs().max())
    assert 0.0 < dsep <= 1e-12, f"separable box filter differs from the tap-order sum by {dsep:.3e} of Linf (0 would mean the exact kernel ran)"
    del ts, fs
    tin.fill_(1.0)
    to, fo = outs_f[0]
    torch.cuda.synchronize()
    ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, fin.h, fo.h, 0, 1, ng, w))
    ctx.sync()
    off1, cs1, _ = mf_layout(lv.boxes, 1, 0)
    for b in range(lv.nboxes):
        assert bool((to[off1[b]: off1[b] + 128 ** 3] == 1.0).all()), "box filter does not reproduce a constant exactly"
    del outs_f, tin, fin, to, fo

    # ---- 6. marching cubes, the same level: sphere of radius 0.3, 3 coordinate comps + 2 fields, one ghost layer
    g = 130
    off5, cs5, tot5 = mf_layout(lv.boxes, 5, 1)
    t5 = torch.empty(tot5, dtype=torch.float64, device=dev)
    for b in range(lv.nboxes):
        lo = lv.boxes[b, :3] - 1
        xs = [(torch.arange(int(lo[d]), int(lo[d]) + g, device=dev, dtype=torch.float64) + 0.5) / n for d in range(3)]
        Xb, Yb, Zb = xs[0][None, None, :].expand(g, g, g), xs[1][None, :, None].expand(g, g, g), xs[2][:, None, None].expand(g, g, g)
        rb = torch.sqrt((Xb - 0.5) ** 2 + (Yb - 0.5) ** 2 + (Zb - 0.5) ** 2)
        for c, v in enumerate((Xb, Yb, Zb, 300.0 + 1700.0 * 0.5 * (1 + torch.tanh((rb - 0.3) / 0.05)), rb)):
            t5[off5[b] + c * cs5[b]: off5[b] + c * cs5[b] + g ** 3] = v.reshape(-1)
    st5 = capi.DevMF(ctx, dl, 5, 1, t5.data_ptr())
    tmk, mk5 = one(1, 1, lambda tot: torch.zeros(tot, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    ctx.check(ctx.lib.pa_iso_mask_level(ctx.h, mk5.h, 0, None, 2))
    ctx.sync()
    assert bool((tmk[: g ** 3] == 1.0).all())
    loops = np.zeros((lv.nboxes, 6), np.int64)
    for b in range(lv.nboxes):
        loops[b, :3] = np.maximum(lv.boxes[b, :3] - 1, 0)
        loops[b, 3:] = np.minimum(lv.boxes[b, 3:] + 1, n - 1) - 1
    frags = capi.mc_level(ctx, st5, mk5, loops, 3, 1150.0)
    ntot = sum(len(f[2]) for f in frags)
    assert ntot > 800000, ntot  # 4 pi (0.3 * 512)^2 cells cut, about 2 triangles each, + the FAB overlaps
    checked = 0
    offm, csm, _ = mf_layout(lv.boxes, 1, 1)
    for b, (v, k, t) in enumerate(frags):
        if len(t):
            assert t.min() >= 0 and t.max() < len(v), f"FAB {b}: connectivity leaves the FAB's vertex range"
        if len(t) and checked < 3:
            fs, fm, bx = capi.PaFab(), capi.PaFab(), capi.PaBox()
            fs.p, fs.ncomp, fs.nstride = t5.data_ptr() + 8 * int(off5[b]), 5, int(cs5[b])
            fm.p, fm.ncomp, fm.nstride = tmk.data_ptr() + 8 * int(offm[b]), 1, int(csm[b])
            for d in range(3):
                fs.lo[d] = fm.lo[d] = int(lv.boxes[b, d]) - 1
                fs.hi[d] = fm.hi[d] = int(lv.boxes[b, 3 + d]) + 1
                bx.lo[d], bx.hi[d] = int(loops[b, d]), int(loops[b, 3 + d])
            nv, nt = C.c_int64(0), C.c_int64(0)
            ctx.check(ctx.lib.pa_mc_count_fab(ctx.h, bx, fs, fm, 3, 1150.0, C.byref(nv), C.byref(nt)))
            assert (nv.value, nt.value) == (len(v), len(t)), f"FAB {b}: counts differ from the per-FAB entry point"
            dv, dk, dt = capi.DevBuf(ctx, nv.value * 40), capi.DevBuf(ctx, nv.value * 24), capi.DevBuf(ctx, nt.value * 12)
            ctx.check(ctx.lib.pa_mc_emit_fab(ctx.h, bx, fs, fm, 3, 1150.0, dv.ptr, dk.ptr, dt.ptr, nv.value, nt.value))
            assert np.array_equal(dv.to_numpy(np.float64, (nv.value, 5)).view(np.int64), np.ascontiguousarray(v).view(np.int64))
            assert np.array_equal(dk.to_numpy(np.int32, (nv.value, 6)), k) and np.array_equal(dt.to_numpy(np.int32, (nt.value, 3)), t)
            checked += 1
    assert checked == 3
    del t5, st5, tmk, mk5, frags
    torch.cuda.empty_cache()

    # ---- 7. BASELINE config 2: ONE 512^3 level, 10 components through the fused pipeline into recycled outputs: every
    # component's result equals the pass-by-pass kernels' and a second run of the fused pipeline, bit for bit
    lv1 = Level(chop_box((0, 0, 0), (n - 1,) * 3, 128), (0, 0, 0), (n - 1,) * 3, (1, 1, 0), (0, 0, 0), (1, 1, 1))
    dl1 = capi.DevLevel(ctx, lv1)
    nc2 = 10
    off, cs, tot = mf_layout(lv1.boxes, nc2, 2)
    tin2 = torch.zeros(tot, dtype=torch.float64, device=dev)
    bench.fill_level_on_device(torch, lv1, tin2, nc2, 2, off, cs, dev, 777)
    st2 = [capi.DevMF(ctx, dl1, nc2, 2, tin2.data_ptr())]
    _, _, tw = mf_layout(lv1.boxes, 1, 2)
    _, _, to8 = mf_layout(lv1.boxes, 8, 0)
    tw2 = torch.zeros(tw, dtype=torch.float64, device=dev)
    wk2 = [capi.DevMF(ctx, dl1, 1, 2, tw2.data_ptr())]
    touts2 = [torch.zeros(to8, dtype=torch.float64, device=dev) for _ in range(3)]
    outs2 = [[capi.DevMF(ctx, dl1, 8, 0, t.data_ptr())] for t in touts2]
    torch.cuda.synchronize()
    bc2 = capi.bc_from_flags((1, 1, 0))
    for c in range(nc2):
        pm = capi.curv_params(prog_min=200.0, prog_max=4000.0, fused=True)
        capi.gradcurv_run(ctx, st2, c, bc2, pm, wk2, outs2[0], 0)
        capi.gradcurv_run(ctx, st2, c, bc2, pm, wk2, outs2[1], 0)
        capi.gradcurv_run(ctx, st2, c, bc2, capi.curv_params(prog_min=200.0, prog_max=4000.0, fused=False), wk2, outs2[2], 0)
        ctx.sync()
        assert same_bits(touts2[0], touts2[1]), f"config 2, component {c}: not deterministic"
        assert same_bits(touts2[0], touts2[2]), f"config 2, component {c}: fused pipeline differs from the pass-by-pass kernels"
    assert ctx.bc_errors() == 0
    del tin2, tw2, touts2, outs2, st2, wk2
    torch.cuda.empty_cache()

    # ---- 8. BASELINE config 3: filterPlt's ghost fill + box filter, then grad, on the 3-level base-256^3 hierarchy (64^3 boxes,
    # periodic x/y, wall z; fgr 2 / 4 / 8): a constant and (away from walls) a linear field are reproduced across both
    # coarse-fine interfaces, the filter is exactly homogeneous under phi -> 2 phi, two runs agree bit for bit, and the
    # gradient of the filtered linear field is the exact constant
    H3 = nested_hierarchy(256, 3, 64, is_per=(1, 1, 0))
    dl3 = [capi.DevLevel(ctx, lv) for lv in H3.levels]
    ngs = [1, 2, 4]
    ws = []
    for f in (2, 4, 8):
        w_ = (C.c_double * (f + 2))()
        assert ctx.lib.pa_box_filter_weights(f, w_) == f // 2
        ws.append(w_)

    def lin(lv_, b, g_):
        lo = lv_.boxes[b, :3] - g_
        nz_, ny_, nx_ = lv_.box_shape(b, g_)
        dx = lv_.dx
        z = (torch.arange(int(lo[2]), int(lo[2]) + nz_, device=dev, dtype=torch.float64) + 0.5) * dx[2]
        return (1.0 + 3.0 * z)[:, None, None].expand(nz_, ny_, nx_)  # linear in the wall direction only (x / y are periodic)

    def filt(fill):
        tin_, tout_, fi, fo = [], [], [], []
        for l, (lv_, dl_) in enumerate(zip(H3.levels, dl3)):
            off_, cs_, tot_ = mf_layout(lv_.boxes, 1, ngs[l])
            t = torch.full((tot_,), float("nan"), dtype=torch.float64, device=dev)
            for b in range(lv_.nboxes):
                nz_, ny_, nx_ = lv_.box_shape(b, ngs[l])
                g_ = ngs[l]
                v = t[off_[b]: off_[b] + nz_ * ny_ * nx_].view(nz_, ny_, nx_)
                v[g_:-g_, g_:-g_, g_:-g_] = fill(lv_, b)
            _, _, to_ = mf_layout(lv_.boxes, 1, 1)
            o = torch.zeros(to_, dtype=torch.float64, device=dev)
            tin_.append(t); tout_.append(o)
            fi.append(capi.DevMF(ctx, dl_, 1, ngs[l], t.data_ptr())); fo.append(capi.DevMF(ctx, dl_, 1, 1, o.data_ptr()))
        torch.cuda.synchronize()
        for l in range(3):
            ctx.check(ctx.lib.pa_fill_boundary(ctx.h, fi[l].h, 0, 1, ngs[l]))
            if l > 0:
                ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, fi[l].h, fi[l - 1].h, 0, 1, ngs[l], 2, 1))
            ctx.check(ctx.lib.pa_foextrap(ctx.h, fi[l].h, 0, 1, ngs[l]))
            ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, fi[l].h, fo[l].h, 0, 1, ngs[l], ws[l]))
        ctx.sync()
        assert ctx.bc_errors() == 0
        return tin_, tout_, fo

    def valid(lv_, t, b, g_):
        off_, cs_, _ = mf_layout(lv_.boxes, 1, g_)
        nz_, ny_, nx_ = lv_.box_shape(b, g_)
        v = t[off_[b]: off_[b] + nz_ * ny_ * nx_].view(nz_, ny_, nx_)
        return v[g_:nz_ - g_, g_:ny_ - g_, g_:nx_ - g_] if g_ else v

    _, oc, _ = filt(lambda lv_, b: 3.0)
    for l, lv_ in enumerate(H3.levels):
        for b in range(lv_.nboxes):
            assert bool((valid(lv_, oc[l], b, 1) == 3.0).all()), f"config 3: constant not reproduced on level {l} box {b}"
    rnd = [300.0 + 1700.0 * torch.rand((lv_.nboxes, 64, 64, 64), dtype=torch.float64, device=dev) for lv_ in H3.levels]
    lvid = {id(lv_): i for i, lv_ in enumerate(H3.levels)}
    _, o1, _ = filt(lambda lv_, b: rnd[lvid[id(lv_)]][b])
    _, o1b, _ = filt(lambda lv_, b: rnd[lvid[id(lv_)]][b])
    _, o2, _ = filt(lambda lv_, b: 2.0 * rnd[lvid[id(lv_)]][b])
    for l in range(3):
        assert same_bits(o1[l], o1b[l]), f"config 3: filter not deterministic on level {l}"
        assert same_bits(o1[l] * 2.0, o2[l]), f"config 3: filter not exactly homogeneous on level {l}"
    _, ol, fo_l = filt(lambda lv_, b: lin(lv_, b, 0))
    for l, lv_ in enumerate(H3.levels):
        for b in range(lv_.nboxes):
            zlo, zhi = int(lv_.boxes[b, 2]), int(lv_.boxes[b, 5])
            nzd = int(lv_.domhi[2]) + 1
            k0, k1 = max(zlo, ngs[l]), min(zhi, nzd - 1 - ngs[l])  # a filter width away from the walls
            if k0 > k1:
                continue
            got = valid(lv_, ol[l], b, 1)[k0 - zlo: k1 - zlo + 1]
            want = lin(lv_, b, 0)[k0 - zlo: k1 - zlo + 1]
            assert float((got - want).abs().max()) < 1e-13, f"config 3: linear field not reproduced on level {l} box {b}"
    # grad of the filtered linear field (filterPlt + grad of BASELINE config 3): dz = 3 away from the walls, dx = dy = 0
    gts, gos = [], []
    for l, (lv_, dl_) in enumerate(zip(H3.levels, dl3)):
        _, _, tg_ = mf_layout(lv_.boxes, 4, 0)
        t = torch.zeros(tg_, dtype=torch.float64, device=dev)
        gts.append(t); gos.append(capi.DevMF(ctx, dl_, 4, 0, t.data_ptr()))
    torch.cuda.synchronize()
    capi.grad_run(ctx, fo_l, 0, capi.bc_from_flags((1, 1, 0)), gos, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for l, lv_ in enumerate(H3.levels):
        off4, cs4, _ = mf_layout(lv_.boxes, 4, 0)
        for b in range(lv_.nboxes):
            zlo, zhi = int(lv_.boxes[b, 2]), int(lv_.boxes[b, 5])
            nzd = int(lv_.domhi[2]) + 1
            k0, k1 = max(zlo, ngs[l] + 1), min(zhi, nzd - 2 - ngs[l])
            if k0 > k1:
                continue
            g3 = [gts[l][off4[b] + c * cs4[b]: off4[b] + c * cs4[b] + 64 ** 3].view(64, 64, 64)[k0 - zlo: k1 - zlo + 1] for c in range(3)]
            assert float(g3[0].abs().max()) < 1e-9 and float(g3[1].abs().max()) < 1e-9 and float((g3[2] - 3.0).abs().max()) < 1e-9, (l, b)
    # ---- 9. small boxes (AMReX's default max_grid_size = 32): the headline hierarchy's shape at base 256^3 in 32^3 boxes
    # (512 boxes per level) -- the narrow sweep's exact-normal variant + one-layer fix-up against the pass-by-pass kernels
    # and against the first pipeline (PA_FUSED2=0), with and without the threshold clip, bit for bit
    del gts, gos
    torch.cuda.empty_cache()
    Hs = nested_hierarchy(256, 3, 32, is_per=(1, 1, 0))
    dls_s = [capi.DevLevel(ctx, lv_) for lv_ in Hs.levels]

    def mfs_s(ncomp, ng):
        ts, ms = [], []
        for lv_, dl_ in zip(Hs.levels, dls_s):
            _, _, tot_ = mf_layout(lv_.boxes, ncomp, ng)
            t = torch.zeros(tot_, dtype=torch.float64, device=dev)
            ts.append(t); ms.append(capi.DevMF(ctx, dl_, ncomp, ng, t.data_ptr()))
        return ts, ms

    tin_s, st_s = mfs_s(1, 2)
    for li, lv_ in enumerate(Hs.levels):
        off_, cs_, _ = mf_layout(lv_.boxes, 1, 2)
        bench.fill_level_on_device(torch, lv_, tin_s[li], 1, 2, off_, cs_, dev, 777 + li)
    prist_s = [t.clone() for t in tin_s]
    _tw, wk_s = mfs_s(1, 2)

    def run_s(fused, thr):
        for t, p_ in zip(tin_s, prist_s):
            t.copy_(p_)
        to_, o_ = mfs_s(8, 0)
        torch.cuda.synchronize()
        capi.gradcurv_run(ctx, st_s, 0, bc, capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=thr, fused=fused), wk_s, o_, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        return to_, o_, ctx.lib.pa_sweep_kernel_name(ctx.h).decode()

    for thr in (None, 0.05):
        a, _ka, kn = run_s(True, thr)
        assert "march3n" in kn and ("CG=1" in kn or "_levels<" in kn), kn  # the narrow exact-normal sweep did run (level by level or all levels in one launch)
        b_, _kb, _ = run_s(False, thr)
        for l in range(3):
            assert same_bits(a[l], b_[l]), f"32^3 boxes, level {l}, threshold {thr}: narrow exact-normal pipeline differs from the pass-by-pass kernels"
        del b_, _kb
        if thr is not None:
            os.environ["PA_FUSED2"] = "0"
            capi.reload_options()
            c_, _kc, kn1 = run_s(True, thr)
            os.environ.pop("PA_FUSED2")
            capi.reload_options()
            assert "CG=0" in kn1, kn1
            for l in range(3):
                assert same_bits(a[l], c_[l]), f"32^3 boxes, level {l}: the two fused pipelines differ under the threshold clip"
            del c_, _kc
        del a, _ka
    print("fullsize properties OK")


if __name__ == "__main__":
    main()
