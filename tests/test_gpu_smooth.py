"""GPU parity: do_smooth (curvature.cpp:328-406) -- pa_smooth_solve vs the oracle's composite solve.
An iterative solve to a tolerance: GPU and oracle run the same algorithm with different summation
orders, so the comparison is to a stated tolerance: 1e-12 absolute on a field in [0,1] (= north_star's 1e-12 relative
to the field's scale) with both sides iterated to a residual of 1e-14 (cond(I - dt Lap) ~ 25 here), plus the
solver-independent checks (exact discrete eigenmode, residual of the GPU
solution under the ORACLE's operator)."""
import numpy as np
import pytest

from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import MultiFab, Level, chop_box, nested_hierarchy, fill_analytic, field_flame

pytestmark = pytest.mark.gpu


def _solve_gpu(ctx, levels, rhs, dt, bc, tol):
    dls = [capi.DevLevel(ctx, lv) for lv in levels]
    drhs = [capi.DevMF.from_host(ctx, dl, r) for dl, r in zip(dls, rhs)]
    dsol = [capi.DevMF(ctx, dl, 1, 0) for dl in dls]
    it, res = capi.smooth_solve(ctx, drhs, 0, dsol, 0, dt, bc, tol=tol, maxiter=200)
    return [d.download() for d in dsol], it, res


def test_smooth_single_level_eigenmode(ctx):
    n = 32
    lv = Level(chop_box((0, 0, 0), (n - 1,) * 3, 16), (0, 0, 0), (n - 1,) * 3, (1, 1, 1), (0, 0, 0), (1, 1, 1))
    rhs = MultiFab(lv, 1, 0)
    kx, ky, kz = 3, 1, 2
    fill_analytic(rhs, 0, lambda x, y, z: 0.5 + 0.25 * np.sin(2 * np.pi * kx * x) * np.cos(2 * np.pi * ky * y) * np.sin(2 * np.pi * kz * z + 0.3))
    dt, h = 2e-3, 1.0 / n
    sol, it, res = _solve_gpu(ctx, [lv], [rhs], dt, (0, 0, 0), 1e-14)
    assert 0 < it < 60 and res <= 1e-14
    lam = sum((2 - 2 * np.cos(2 * np.pi * k * h)) / h ** 2 for k in (kx, ky, kz))
    for b in range(lv.nboxes):
        assert np.abs(sol[0].valid(b)[0] - (0.5 + (rhs.valid(b)[0] - 0.5) / (1 + dt * lam))).max() < 1e-13


@pytest.mark.parametrize("per", [(1, 1, 0), (0, 0, 0)])
def test_smooth_composite_matches_oracle(ctx, oracle, per):
    H = nested_hierarchy(16, 3, 8, is_per=per)
    rhs = []
    for lv in H.levels:
        m = MultiFab(lv, 1, 0)
        fill_analytic(m, 0, lambda x, y, z: (field_flame(x, y, z, 0) - 300.0) / 1700.0)
        rhs.append(m)
    bc = capi.bc_from_flags(per)
    dt = 5e-4
    want, oit, ores = oracle.smooth_solve(H.levels, rhs, 0, dt, bc, MultiFab, tol=1e-14)
    got, it, res = _solve_gpu(ctx, H.levels, rhs, dt, bc, 1e-14)
    assert 0 < it < 100 and res <= 1e-14 and abs(it - oit) <= 3
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            assert np.abs(got[l].valid(b)[0] - want[l].valid(b)[0]).max() <= 1e-12, (l, b)
    # the GPU solution under the oracle's composite operator
    x = [MultiFab(lv, 1, 1) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            x[l].valid(b)[0] = got[l].valid(b)[0]
    y, mask = oracle.smooth_apply(H.levels, x, dt, bc, MultiFab)
    r = max(float(np.abs((y[l].valid(b)[0] - rhs[l].valid(b)[0]) * mask[l].valid(b)[0]).max()) for l, lv in enumerate(H.levels) for b in range(lv.nboxes))
    assert r <= 1e-12


@pytest.mark.parametrize("per,sym", [((1, 1, 0), (0, 0, 0)), ((0, 1, 0), (1, 0, 1))])
def test_curvature_run_with_smoothing(ctx, oracle, per, sym):
    """the tool path: Progress stays unsmoothed, SmoothedProgress feeds the curvature (idprogvar, :408).  With sym_dir set
    the gradient operators see reflect_odd walls but the smoothing operator stays Neumann (curvature.cpp:348-357)."""
    from util import make_states
    H = nested_hierarchy(16, 3, 8, is_per=per)
    states = make_states(H, 1, 2, field_flame, seed=41)
    bc = capi.bc_from_flags(per, sym)
    dt = 1e-3
    oout = [MultiFab(lv, 18, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oout, 0, MultiFab, do_smooth=True, smoothing_time=dt, smooth_tol=1e-14)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    dout = [capi.DevMF(ctx, dl, 18, 0) for dl in dls]
    capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(fused=False, do_smooth=True, smoothing_time=dt), dout, 0)
    ctx.sync()
    assert ctx.lib.pa_curvature_last_path(ctx.h) == 0
    # the same solve feeding the exact-normal pipeline (fused=True: the smoothed field as a progress source of range [0, 1], which the
    # sweeps' (x - pmin) * invdenom leaves bit for bit): every output, options included, equals the pass-by-pass path's bits
    st4 = make_states(H, 4, 2, field_flame, seed=41)
    dst4 = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, st4)]
    both = []
    for fused in (False, True):
        o = [capi.DevMF(ctx, dl, 18, 0) for dl in dls]
        for m in o:
            m.setval(-7.0)
        capi.curvature_run(ctx, dst4, 0, bc, capi.curv_params(threshold=0.04, fused=fused, do_smooth=True, smoothing_time=dt, do_gauss=True, do_strain=True,
                                                             strain_tensor=True, do_velnormal=True, vel_comp=1), o, 0)
        ctx.sync()
        assert ctx.lib.pa_curvature_last_path(ctx.h) == (1 if fused else 0)
        both.append([m.download() for m in o])
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            assert np.array_equal(both[0][l].valid(b).view(np.int64), both[1][l].valid(b).view(np.int64)), f"do_smooth + options: fast path != pass by pass, level {l} box {b}"
    for l, lv in enumerate(H.levels):
        g = dout[l].download()
        for b in range(lv.nboxes):
            gv, wv = g.valid(b), oout[l].valid(b)
            assert np.array_equal(gv[0].view(np.int64), wv[0].view(np.int64)), "Progress (unsmoothed) must stay bit-identical"
            assert np.abs(gv[17] - wv[17]).max() <= 1e-12, "SmoothedProgress"
            assert not np.array_equal(gv[17], gv[0])
            # Downstream fields.  n = -G / |G| and K = 0.5 div n amplify a difference eps in the smoothed field by 1 / |G|; the
            # tolerance is DERIVED from the measured eps, not chosen.  Cells whose stencil stays inside the box (no ghost cell):
            #   G_d = 0.5 dxinv_d (c[+1] - c[-1])            =>  |dG_d| <= dxinv_d eps,  ||dG||_2 <= eps S,  S = sqrt(sum dxinv_d^2)
            #   n = G / |G|:  |a/|a| - b/|b|| <= 2 |a - b| / max(|a|, |b|)   =>  ||dn||_inf <= 2 eps S / |G|
            #   K = 0.5 sum_d 0.5 dxinv_d (n_d[+1] - n_d[-1])                =>  |dK| <= 0.5 (sum_d dxinv_d) max_neighbours ||dn||_inf
            # (+ a few ulp of the fields' own rounding).  Where |G| <= 1e-14 the reference clamps it: not compared.
            eps = float(np.abs(gv[17] - wv[17]).max())
            dxinv = 1.0 / np.asarray(lv.dx)
            S, S1 = float(np.sqrt((dxinv ** 2).sum())), float(dxinv.sum())
            cs = wv[17]
            Gz, Gy, Gx = [0.5 * dxinv[d] * (np.roll(cs, -1, axis=2 - d) - np.roll(cs, 1, axis=2 - d)) for d in (2, 1, 0)]
            Gm = np.sqrt(Gx ** 2 + Gy ** 2 + Gz ** 2)
            inner = np.zeros(cs.shape, bool)
            inner[1:-1, 1:-1, 1:-1] = True  # np.roll wrapped at the box faces: those cells use ghost data, not bounded here
            ok = inner & (Gm > 1e-6)
            dn_bound = np.where(ok, 2.0 * eps * S / np.maximum(Gm - eps * S, 1e-300) + 8e-16, np.inf)
            for c in (2, 3, 4):
                d = np.abs(gv[c] - wv[c])
                assert np.all(d[ok] <= dn_bound[ok]), (l, b, c, float((d[ok] / dn_bound[ok]).max()))
            nb = dn_bound.copy()  # the bound of a cell's six neighbours and its own
            for ax in range(3):
                nb = np.maximum(nb, np.maximum(np.roll(dn_bound, 1, axis=ax), np.roll(dn_bound, -1, axis=ax)))
            inner2 = np.zeros(cs.shape, bool)
            inner2[2:-2, 2:-2, 2:-2] = True
            okk = inner2 & np.isfinite(nb)
            dk = np.abs(gv[1] - wv[1])
            kb = 0.5 * S1 * nb + 1e-13 * max(1.0, float(np.abs(wv[1]).max()))
            assert np.all(dk[okk] <= kb[okk]), (l, b, float((dk[okk] / kb[okk]).max()))
    # EVERY cell (box faces with their coarse-fine / wall ghost data, the flat parts of the profile), without a tolerance: the
    # downstream fields are a function f of the smoothed field, and f on the HIP path is the reference's f BIT FOR BIT -- the ORACLE's
    # smoothed field handed to the HIP pipeline as a progress variable of range [0, 1] ((c - 0.0) * 1.0 = c exactly) gives the
    # oracle's normals and curvature in every cell.  What separates the two runs above is therefore the solve's eps alone
    # (<= 1e-12, iterative on both sides -- MLMG's in the reference), amplified by the conditioning of f: the bounds above.
    sm = []
    for l, lv in enumerate(H.levels):
        m = MultiFab(lv, 1, 2)
        for b in range(lv.nboxes):
            m.valid(b)[0] = oout[l].valid(b)[17]
        sm.append(m)
    dsm = [capi.DevMF.from_host(ctx, dl, m) for dl, m in zip(dls, sm)]
    for fused in (False, True):
        o = [capi.DevMF(ctx, dl, 18, 0) for dl in dls]
        capi.curvature_run(ctx, dsm, 0, bc, capi.curv_params(prog_min=0.0, prog_max=1.0, fused=fused), o, 0)
        ctx.sync()
        for l, lv in enumerate(H.levels):
            g = o[l].download()
            for b in range(lv.nboxes):
                gv, wv = g.valid(b), oout[l].valid(b)
                assert np.array_equal(gv[0].view(np.int64), wv[17].view(np.int64))
                for c in (1, 2, 3, 4):
                    assert np.array_equal(gv[c].view(np.int64), wv[c].view(np.int64)), f"f(oracle's smoothed field), fused={fused}: level {l} box {b} comp {c}"


def _hier2d(per):
    from peleanalysis_amd.hierarchy import Hierarchy
    per3 = np.array([per[0], per[1], 0])
    l0 = Level(chop_box((0, 0, 0), (31, 31, 0), 16), (0, 0, 0), (31, 31, 0), per3, np.zeros(3), np.ones(3))
    l1 = Level(chop_box((16, 16, 0), (47, 47, 0), 16), (0, 0, 0), (63, 63, 0), per3, np.zeros(3), np.ones(3))
    return Hierarchy([l0, l1], 2)


@pytest.mark.parametrize("per", [(1, 0), (0, 0)])
def test_smooth_2d_composite_matches_oracle(ctx, oracle, per):
    """the AMREX_SPACEDIM == 2 build of do_smooth: one plane of cells per level, refined in x and y only (covered blocks
    of 2 x 2, two fine faces per coarse face in the reflux), z a Neumann wall"""
    H = _hier2d(per)
    rhs = []
    for lv in H.levels:
        m = MultiFab(lv, 1, 0)
        fill_analytic(m, 0, lambda x, y, z: 0.5 * (1.0 + np.tanh((np.hypot(x - 0.5, (y - 0.5) / 0.8) - 0.27) / 0.06)) + 0 * z)
        rhs.append(m)
    bc = capi.bc_from_flags((per[0], per[1], 0))
    dt = 4e-4
    want, oit, ores = oracle.smooth_solve(H.levels, rhs, 0, dt, bc, MultiFab, tol=1e-14)
    got, it, res = _solve_gpu(ctx, H.levels, rhs, dt, bc, 1e-14)
    assert 0 < it < 100 and res <= 1e-14 and abs(it - oit) <= 3
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            assert np.abs(got[l].valid(b)[0] - want[l].valid(b)[0]).max() <= 1e-12, (l, b)
    x = [MultiFab(lv, 1, 1) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            x[l].valid(b)[0] = got[l].valid(b)[0]
    y, mask = oracle.smooth_apply(H.levels, x, dt, bc, MultiFab)
    r = max(float(np.abs((y[l].valid(b)[0] - rhs[l].valid(b)[0]) * mask[l].valid(b)[0]).max()) for l, lv in enumerate(H.levels) for b in range(lv.nboxes))
    assert r <= 1e-12


def test_smooth_2d_single_level_eigenmode(ctx):
    n = 64
    lv = Level(chop_box((0, 0, 0), (n - 1, n - 1, 0), 32), (0, 0, 0), (n - 1, n - 1, 0), (1, 1, 0), (0, 0, 0), (1, 1, 1))
    rhs = MultiFab(lv, 1, 0)
    kx, ky = 3, 5
    fill_analytic(rhs, 0, lambda x, y, z: 0.5 + 0.25 * np.sin(2 * np.pi * kx * x + 0.1) * np.cos(2 * np.pi * ky * y) + 0 * z)
    dt, h = 5e-4, 1.0 / n
    sol, it, res = _solve_gpu(ctx, [lv], [rhs], dt, capi.bc_from_flags((1, 1, 0)), 1e-14)
    assert 0 < it < 60 and res <= 1e-14
    lam = sum((2 - 2 * np.cos(2 * np.pi * k * h)) / h ** 2 for k in (kx, ky))
    for b in range(lv.nboxes):
        assert np.abs(sol[0].valid(b)[0] - (0.5 + (rhs.valid(b)[0] - 0.5) / (1 + dt * lam))).max() < 1e-13


def test_smooth_and_stream_reject_bad_input(ctx):
    """host-side shape checks before any kernel runs: misaligned fine boxes, component ranges, ghost widths"""
    import ctypes as C
    from peleanalysis_amd.hierarchy import Hierarchy
    l0 = Level(chop_box((0, 0, 0), (15, 15, 15), 8), (0, 0, 0), (15, 15, 15), (1, 1, 1), (0, 0, 0), (1, 1, 1))
    l1 = Level(np.array([[9, 8, 8, 18, 23, 23]], dtype=np.int32), (0, 0, 0), (31, 31, 31), (1, 1, 1), (0, 0, 0), (1, 1, 1))  # odd lo: not ratio-aligned
    dls = [capi.DevLevel(ctx, l0), capi.DevLevel(ctx, l1)]
    rhs = [capi.DevMF(ctx, dl, 1, 0) for dl in dls]
    sol = [capi.DevMF(ctx, dl, 1, 0) for dl in dls]
    with pytest.raises(capi.PaError, match="aligned"):
        capi.smooth_solve(ctx, rhs, 0, sol, 0, 1e-3, (0, 0, 0))
    with pytest.raises(capi.PaError, match="component"):
        capi.smooth_solve(ctx, rhs[:1], 3, sol[:1], 0, 1e-3, (0, 0, 0))
    v = [capi.DevMF(ctx, dls[0], 3, 0)]  # no ghost layers
    with pytest.raises(capi.PaError, match="nGrow"):
        capi.stream_trace(ctx, v, 0, np.array([[0.5, 0.5, 0.5]]), 5, 0.01)
    v = [capi.DevMF(ctx, dls[0], 2, 2)]  # too few components
    with pytest.raises(capi.PaError, match="component"):
        capi.stream_trace(ctx, v, 0, np.array([[0.5, 0.5, 0.5]]), 5, 0.01)
    pos, nred = capi.stream_trace(ctx, [capi.DevMF(ctx, dls[0], 3, 2)], 0, np.zeros((0, 3)), 5, 0.01)  # no seeds: nothing to do
    assert pos.shape == (0, 5, 3) and nred == 0


@pytest.mark.parametrize("per,base,box", [((1, 1, 0), 16, 8), ((0, 0, 0), 32, 16), ((1, 1, 0), 12, 6)])  # 6-cell boxes: level 0 cannot be coarsened, more steps on it
def test_smooth_multigrid_preconditioner(ctx, oracle, per, base, box, monkeypatch):
    """a STIFF smoothing step (dt / dx^2 = 100 on the finest level: what smoothing_time = 1e-7 is for a plotfile in physical units):
    BiCGStab with the V-cycle preconditioner (default there; PA_SMOOTH_MG=1) reaches the tolerance in a fraction of the iterations of the
    unpreconditioned solve (PA_SMOOTH_MG=0), both fields agree with each other and with the oracle's unpreconditioned solve to the
    solve's tolerance times the condition number, and the preconditioned field has a small residual under the ORACLE's operator"""
    H = nested_hierarchy(base, 3, box, is_per=per)
    rhs = []
    for lv in H.levels:
        m = MultiFab(lv, 1, 0)
        fill_analytic(m, 0, lambda x, y, z: (field_flame(x, y, z, 0) - 300.0) / 1700.0)
        rhs.append(m)
    bc = capi.bc_from_flags(per)
    nf = base * 4
    dt = 100.0 / nf ** 2
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    drhs = [capi.DevMF.from_host(ctx, dl, r) for dl, r in zip(dls, rhs)]
    out = {}
    for mg in ("0", "1", ""):
        if mg:
            monkeypatch.setenv("PA_SMOOTH_MG", mg)
        else:
            monkeypatch.delenv("PA_SMOOTH_MG")
        capi.reload_options()
        dsol = [capi.DevMF(ctx, dl, 1, 0) for dl in dls]
        it, res = capi.smooth_solve(ctx, drhs, 0, dsol, 0, dt, bc, tol=1e-13, maxiter=2000)
        assert res <= 1e-13
        out[mg] = ([d.download() for d in dsol], it)
    assert out["1"][1] == out[""][1], "dt / dx^2 = 100: the preconditioner is the default"
    # pa_curvature_run iterates to 1e-14 (and accepts 1e-12): the preconditioned recurrence gets there too, in a handful of iterations
    dsol = [capi.DevMF(ctx, dl, 1, 0) for dl in dls]
    it14, res14 = capi.smooth_solve(ctx, drhs, 0, dsol, 0, dt, bc, tol=1e-14, maxiter=100)
    assert res14 <= 1e-14 and it14 <= out["1"][1] + 6, (it14, res14)
    assert out["1"][1] * 3 <= out["0"][1], f"iterations: preconditioned {out['1'][1]}, plain {out['0'][1]}"
    want, oit, ores = oracle.smooth_solve(H.levels, rhs, 0, dt, bc, MultiFab, tol=1e-13, maxiter=2000)
    assert ores <= 1e-13
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            assert np.abs(out["1"][0][l].valid(b)[0] - out["0"][0][l].valid(b)[0]).max() <= 1e-10, (l, b)
            assert np.abs(out["1"][0][l].valid(b)[0] - want[l].valid(b)[0]).max() <= 1e-10, (l, b)
    x = [MultiFab(lv, 1, 1) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            x[l].valid(b)[0] = out["1"][0][l].valid(b)[0]
    y, mask = oracle.smooth_apply(H.levels, x, dt, bc, MultiFab)
    r = max(float(np.abs((y[l].valid(b)[0] - rhs[l].valid(b)[0]) * mask[l].valid(b)[0]).max()) for l, lv in enumerate(H.levels) for b in range(lv.nboxes))
    assert r <= 1e-11


@pytest.mark.parametrize("base", [80, 46])
@pytest.mark.parametrize("dtq", [2.0, 20.0])  # dt / dx^2 on the finest level: the plain iteration / the multigrid-preconditioned one
def test_smooth_wide_boxes_marching_kernels(ctx, oracle, dtq, base, monkeypatch):
    """boxes wider than a wavefront row and taller than a z chunk (every other smoothing test has boxes of <= 16 cells: two rows of 32
    lanes, one chunk): 80-cell boxes = a full 64-cell tile + a partly filled one, two chunks of planes;
    46-cell boxes = one partly filled tile in x and a last tile of rows that is partly outside the box.
    The z-marching stencil kernels (k_smooth_march) give the SAME BITS as the cell-per-thread kernels (PA_SMOOTH_MARCH=0) whatever
    the chunk length, and the field agrees with the oracle's solve"""
    H = nested_hierarchy(base, 2, base, is_per=(1, 0, 0))
    rhs = []
    for lv in H.levels:
        m = MultiFab(lv, 1, 0)
        fill_analytic(m, 0, lambda x, y, z: (field_flame(x, y, z, 0) - 300.0) / 1700.0)
        rhs.append(m)
    bc = capi.bc_from_flags((1, 0, 0))
    dt = dtq / (2.0 * base) ** 2
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    drhs = [capi.DevMF.from_host(ctx, dl, r) for dl, r in zip(dls, rhs)]
    out = {}
    variants = (("0", "64"), ("1", "64"))
    for march, kz in variants:
        monkeypatch.setenv("PA_SMOOTH_MARCH", march)
        capi.reload_options()
        dsol = [capi.DevMF(ctx, dl, 1, 0) for dl in dls]
        it, res = capi.smooth_solve(ctx, drhs, 0, dsol, 0, dt, bc, tol=1e-13, maxiter=500)
        assert res <= 1e-13
        out[(march, kz)] = ([d.download() for d in dsol], it, res)
    monkeypatch.delenv("PA_SMOOTH_MARCH")
    capi.reload_options()
    ref = out[variants[0]]
    for key in variants[1:]:
        assert out[key][1] == ref[1] and out[key][2] == ref[2], (key, out[key][1:], ref[1:])
        for l, lv in enumerate(H.levels):
            for b in range(lv.nboxes):
                assert np.array_equal(out[key][0][l].valid(b)[0], ref[0][l].valid(b)[0]), (key, l, b)
    if base > 64 and dtq > 8:
        return  # (the oracle's unpreconditioned solve of the stiff case takes half a minute at this size: compared on the smaller hierarchy)
    want, oit, ores = oracle.smooth_solve(H.levels, rhs, 0, dt, bc, MultiFab, tol=1e-13, maxiter=2000)
    assert ores <= 1e-13
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            assert np.abs(ref[0][l].valid(b)[0] - want[l].valid(b)[0]).max() <= 1e-10, (l, b)
