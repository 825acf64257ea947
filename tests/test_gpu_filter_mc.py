"""GPU parity: box filter (filterPlt) and marching cubes (isosurface) through the C ABI vs the CPU
oracle.  Derived fields bit-exact (stated tolerance 1e-12 relative); vertex order, edge keys and
triangle connectivity identical."""
import ctypes as C
import os

import numpy as np
import pytest

from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import MultiFab, cell_centers, chop_box, field_flame, nested_hierarchy
from util import assert_filter_parity, assert_valid_bits_equal, make_states, rel_err

pytestmark = pytest.mark.gpu


def _filter_gpu(ctx, H, ins, ncomp, base_fgr, same, interp_type, filter_type=1):
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    din = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, ins)]
    dout = [capi.DevMF(ctx, dl, ncomp, 0) for dl in dls]
    fgr = base_fgr
    for l in range(H.nlev):
        if l > 0 and not same:
            fgr *= 2
        w = (C.c_double * (max(fgr, 3) + 2))()
        ngf = ctx.lib.pa_box_filter_weights(fgr, w) if filter_type == 1 else ctx.lib.pa_filter_weights(filter_type, fgr, w)
        ctx.check(ctx.lib.pa_fill_boundary(ctx.h, din[l].h, 0, ncomp, ngf))
        if l > 0:
            ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, din[l].h, din[l - 1].h, 0, ncomp, ngf, 2, interp_type))
        ctx.check(ctx.lib.pa_foextrap(ctx.h, din[l].h, 0, ncomp, ngf))
        ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, din[l].h, dout[l].h, 0, ncomp, ngf, w))
    ctx.sync()
    assert ctx.bc_errors() == 0
    return [d.download() for d in dout], [d.download() for d in din]


@pytest.mark.parametrize("per,interp_type,same", [((1, 1, 0), 1, False), ((0, 0, 0), 0, False), ((1, 0, 1), 1, True)])
def test_filter_pipeline_matches_oracle(ctx, oracle, filter_mode, per, interp_type, same):
    """filterPlt.cpp:126-219 on a 3-level hierarchy: fgr 2/4/8 (27-, 125-, 729-point box filters),
    FillPatchTwoLevels (cell-conservative linear / piecewise constant) and wall extrapolation"""
    H = nested_hierarchy(32, 3, 16, is_per=per)
    ncomp = 2
    ins = make_states(H, ncomp, 4, field_flame, seed=21)
    o_in = [s.copy() for s in ins]
    o_out = [MultiFab(lv, ncomp, 0) for lv in H.levels]
    info = oracle.filter_pipeline(H.levels, o_in, o_out, ncomp, base_fgr=2, same_fgr_all_levels=same, interp_type=interp_type)
    got, got_in = _filter_gpu(ctx, H, ins, ncomp, 2, same, interp_type)
    for l in range(H.nlev):
        ngf = info[l][1]
        # ghost fill: compare the ngf-deep shell that the filter reads
        for b in range(H.levels[l].nboxes):
            g, w = got_in[l].fab(b), o_in[l].fab(b)
            s = 4 - ngf
            sl = (slice(None), slice(s, g.shape[1] - s), slice(s, g.shape[2] - s), slice(s, g.shape[3] - s))
            assert np.array_equal(g[sl].view(np.int64), w[sl].view(np.int64)), f"ghost fill differs: level {l} box {b}"
        assert_filter_parity(got[l], o_out[l], [(c, c) for c in range(ncomp)], f"filter level {l} (fgr {info[l][0]})", filter_mode)
        for c in range(ncomp):
            assert rel_err(got[l], o_out[l], c, c) <= 1e-12


@pytest.mark.parametrize("ftype", [0, 3, 4, 8])
def test_filter_pipeline_other_filter_types(ctx, oracle, filter_mode, ftype):
    """filter_type 0 / 3 (= 7) / 4 / 8 (filterPlt.cpp:80): the same ghost fill and tap loop with the closed-form weights of
    those PelePhysics types (fgr 2 / 4 / 8 on the three levels enter the weights only: 1-, 3- and 5-point stencils)"""
    H = nested_hierarchy(32, 3, 16, is_per=(1, 0, 0))
    ncomp = 2
    ins = make_states(H, ncomp, 2, field_flame, seed=23)
    o_in = [s.copy() for s in ins]
    o_out = [MultiFab(lv, ncomp, 0) for lv in H.levels]
    oracle.filter_pipeline(H.levels, o_in, o_out, ncomp, base_fgr=2, interp_type=1, filter_type=ftype)
    got, _ = _filter_gpu(ctx, H, ins, ncomp, 2, False, 1, filter_type=ftype)
    for l in range(H.nlev):
        assert_filter_parity(got[l], o_out[l], [(c, c) for c in range(ncomp)], f"filter type {ftype} level {l}", filter_mode)
        if ftype == 0:
            assert_valid_bits_equal(got[l], ins[l], [(c, c) for c in range(ncomp)], "type 0 is the identity")


def test_filter_generic_width_and_fab_entry(ctx, oracle, filter_mode):
    """fgr = 6 (ng = 3, no LDS specialisation) through pa_boxfilter_fab on ragged boxes"""
    from peleanalysis_amd.hierarchy import Level
    lv = Level(chop_box((0, 0, 0), (19, 13, 10), 9), (0, 0, 0), (19, 13, 10), (1, 1, 1), (0, 0, 0), (1, 1, 1))
    rng = np.random.default_rng(5)
    s = MultiFab(lv, 1, 3)
    s.data[:] = rng.standard_normal(s.total)
    ngf, w = oracle.box_filter_weights(6)
    assert ngf == 3 and abs(w.sum() - 1.0) < 1e-15
    oracle.fill_boundary(s, 0, 1, 3)
    oo = MultiFab(lv, 1, 0)
    oracle.lib().orc_apply_filter(C.byref(oracle._mf(s)), C.byref(oracle._mf(oo)), 0, 1, 3, (C.c_double * 7)(*w))
    dl = capi.DevLevel(ctx, lv)
    di = capi.DevMF.from_host(ctx, dl, s)
    do = capi.DevMF(ctx, dl, 1, 0)
    wc = (C.c_double * 7)(*w)
    for b in range(lv.nboxes):
        ctx.check(ctx.lib.pa_boxfilter_fab(ctx.h, capi.box_of(lv, b), di.fab(b), do.fab(b), 0, 1, 3, wc))
    ctx.sync()
    assert_filter_parity(do.download(), oo, [(0, 0)], "pa_boxfilter_fab ng=3", filter_mode)


@pytest.mark.parametrize("fgr", [12, 16])
def test_filter_wide_windows(ctx, oracle, filter_mode, fgr):
    """fgr = 12 / 16 (ng = 6 / 8: 13^3 / 17^3 taps, the widths a 3- or 4-level run with base_fgr = 2 .. 6 reaches on its finest
    level, filterPlt.cpp:132-134): the separable kernel's one-pair-per-thread instantiations (512 and 1024 threads) and the
    generic tap-order kernel, on a level of mixed box widths (40 and 20 cells: both thread counts) through pa_boxfilter_level"""
    from peleanalysis_amd.hierarchy import Level
    ng = fgr // 2
    boxes = np.vstack([chop_box((0, 0, 0), (39, 19, 19), 40), chop_box((40, 0, 0), (59, 19, 19), 20)])
    lv = Level(boxes, (0, 0, 0), (59, 19, 19), (1, 1, 1), (0, 0, 0), (3, 1, 1))
    rng = np.random.default_rng(fgr)
    s = MultiFab(lv, 2, ng)
    s.data[:] = 300.0 + 1700.0 * rng.random(s.total)
    ngf, w = oracle.box_filter_weights(fgr)
    assert ngf == ng
    oracle.fill_boundary(s, 0, 2, ng)
    oo = MultiFab(lv, 2, 0)
    oracle.lib().orc_apply_filter(C.byref(oracle._mf(s)), C.byref(oracle._mf(oo)), 0, 2, ng, (C.c_double * (2 * ng + 1))(*w))
    dl = capi.DevLevel(ctx, lv)
    di = capi.DevMF.from_host(ctx, dl, s)
    do = capi.DevMF(ctx, dl, 2, 0)
    ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, di.h, do.h, 0, 2, ng, (C.c_double * (2 * ng + 1))(*w)))
    ctx.sync()
    assert_filter_parity(do.download(), oo, [(0, 0), (1, 1)], f"fgr {fgr}", filter_mode)


# ------------------------------------------------------------------------------- marching cubes
def _mc_case(n, seed, masked):
    """state FAB over box (-1..n)^3 with 3 coordinate comps + 2 fields; iso field = wrinkled sphere"""
    lo, hi = np.array([-1, -1, -1]), np.array([n, n, n])
    ax = (np.arange(lo[0], hi[0] + 1) + 0.5) / n
    X, Y, Z = ax[None, None, :] + 0 * ax[:, None, None], ax[None, :, None] + 0 * ax[:, None, None], ax[:, None, None] + 0 * ax[None, None, :]
    X, Y, Z = np.broadcast_arrays(X, Y, Z)
    rng = np.random.default_rng(seed)
    r = np.sqrt((X - 0.5) ** 2 + (Y - 0.47) ** 2 + (Z - 0.52) ** 2)
    f = 1000.0 + 900.0 * np.tanh((r - 0.31) / 0.05) + 5.0 * rng.standard_normal(X.shape)
    g = np.sin(3 * X) * np.cos(2 * Y) + Z
    state = np.ascontiguousarray(np.stack([X, Y, Z, f, g]))
    mask = np.ones(X.shape)
    if masked:
        mask[n // 2:, n // 3: 2 * n // 3, : n // 2] = -1.0  # "covered by a finer level"
    # a few exact hits of the iso value and equal neighbours (eps branches of VI_doIt)
    state[3, 3, 4, 5] = 1090.0
    state[3, 7, 7, 7] = state[3, 7, 7, 8]
    return lo, hi, state, mask


@pytest.mark.parametrize("n,masked", [(12, False), (20, True), (33, True)])
def test_marching_cubes_fab_matches_oracle(ctx, oracle, n, masked):
    lo, hi, state, mask = _mc_case(n, 7 + n, masked)
    llo, lhi = lo.copy(), hi - 1
    verts, vkeys, tris = oracle.mc_fab(state, mask, lo, hi, 3, 1090.0, llo, lhi)
    assert len(tris) > 100
    ts, tm = capi.DevBuf.from_numpy(ctx, state), capi.DevBuf.from_numpy(ctx, mask)
    fs, fm, bx = capi.PaFab(), capi.PaFab(), capi.PaBox()
    fs.p, fs.ncomp, fs.nstride = ts.ptr, 5, 0
    fm.p, fm.ncomp, fm.nstride = tm.ptr, 1, 0
    for d in range(3):
        fs.lo[d] = fm.lo[d] = int(lo[d]); fs.hi[d] = fm.hi[d] = int(hi[d])
        bx.lo[d], bx.hi[d] = int(llo[d]), int(lhi[d])
    nv, nt = C.c_int64(0), C.c_int64(0)
    ctx.check(ctx.lib.pa_mc_count_fab(ctx.h, bx, fs, fm, 3, 1090.0, C.byref(nv), C.byref(nt)))
    assert (nv.value, nt.value) == (len(verts), len(tris))
    tv = capi.DevBuf(ctx, nv.value * 5 * 8)
    tk = capi.DevBuf(ctx, nv.value * 6 * 4)
    tt = capi.DevBuf(ctx, nt.value * 3 * 4)
    ctx.check(ctx.lib.pa_mc_emit_fab(ctx.h, bx, fs, fm, 3, 1090.0, tv.ptr, tk.ptr, tt.ptr, nv.value, nt.value))
    assert np.array_equal(tk.to_numpy(np.int32, (nv.value, 6)), vkeys), "edge keys / vertex order differ"
    assert np.array_equal(tt.to_numpy(np.int32, (nt.value, 3)), tris), "triangle connectivity differs"
    assert np.array_equal(tv.to_numpy(np.float64, (nv.value, 5)).view(np.int64), verts.view(np.int64)), "vertex data not bit-identical"
    # wrong buffer sizes are rejected on the host before any launch writes
    assert ctx.lib.pa_mc_emit_fab(ctx.h, bx, fs, fm, 3, 1090.0, tv.ptr, tk.ptr, tt.ptr, nv.value - 1, nt.value) != 0


def test_marching_cubes_empty_and_all_masked(ctx, oracle):
    lo, hi, state, mask = _mc_case(10, 3, False)
    ts = capi.DevBuf.from_numpy(ctx, state)
    fs, fm, bx = capi.PaFab(), capi.PaFab(), capi.PaBox()
    fs.p, fs.ncomp, fs.nstride = ts.ptr, 5, 0
    for d in range(3):
        fs.lo[d] = fm.lo[d] = int(lo[d]); fs.hi[d] = fm.hi[d] = int(hi[d])
        bx.lo[d], bx.hi[d] = int(lo[d]), int(hi[d]) - 1
    nv, nt = C.c_int64(-1), C.c_int64(-1)
    for m, iso in ((np.ones(mask.shape), 1e9), (-np.ones(mask.shape), 1090.0)):  # iso out of range / everything covered
        tm = capi.DevBuf.from_numpy(ctx, m)
        fm.p, fm.ncomp, fm.nstride = tm.ptr, 1, 0
        ctx.check(ctx.lib.pa_mc_count_fab(ctx.h, bx, fs, fm, 3, iso, C.byref(nv), C.byref(nt)))
        assert (nv.value, nt.value) == (0, 0)
        ctx.check(ctx.lib.pa_mc_emit_fab(ctx.h, bx, fs, fm, 3, iso, None, None, None, 0, 0))


@pytest.mark.parametrize("name,ng,cells", [("amr3_wall_z", 1, "slab"), ("amr2_allwalls_ragged", 2, "slab"), ("amr3_sym_x", 1, "slab"), ("amr2_allwalls_ragged", 2, "tiles")])
def test_marching_cubes_level_batched_matches_oracle(ctx, oracle, name, ng, cells, options):
    """pa_iso_mask_level + pa_mc_level over every FAB of every level at once: the fine-covered mask, and per FAB
    the same vertices (bit for bit), edge keys and connectivity as the oracle's per-FAB Polygonise loop"""
    from util import build_config, make_states
    if cells == "tiles":  # the first form of the cell pass (k_mcl_cells<8>): still what FABs wider than 819 cells take
        options(PA_FORCE_FALLBACKS=1)
    H, per, sym, fn = build_config(name)
    fields = make_states(H, 2, 0, fn, seed=11)
    nc = 5
    states = []
    for l, lv in enumerate(H.levels):  # state build of isosurface.cpp:1434-1528 through the oracle's pieces
        st = MultiFab(lv, nc, ng, fill=-666.0)
        for b in range(lv.nboxes):
            f = st.fab(b)
            lo = lv.boxes[b, :3] - ng
            nz, ny, nx = f.shape[1:]
            f[0] = ((np.arange(lo[0], lo[0] + nx) + 0.5) * lv.dx[0] + lv.prob_lo[0])[None, None, :]
            f[1] = ((np.arange(lo[1], lo[1] + ny) + 0.5) * lv.dx[1] + lv.prob_lo[1])[None, :, None]
            f[2] = ((np.arange(lo[2], lo[2] + nz) + 0.5) * lv.dx[2] + lv.prob_lo[2])[:, None, None]
            st.valid(b)[3:5] = fields[l].valid(b)[0:2]
        oracle.fill_boundary(st, 0, nc, ng)
        if l > 0:
            assert oracle.lib().orc_fillpatch_two_levels(C.byref(oracle._mf(st)), C.byref(oracle._mf(states[l - 1])), 0, nc, ng, 2, 0) == 0
        states.append(st)
    allv = np.concatenate([s.valid(b)[3].ravel() for s in states for b in range(s.level.nboxes)])
    iso = float(np.median(allv))
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    ntri_total = 0
    frags = []
    for l, lv in enumerate(H.levels):
        dst = capi.DevMF.from_host(ctx, dls[l], states[l])
        dco = capi.DevMF(ctx, dls[l], 4, ng)  # pa_iso_coords_level: the analytic coordinates of every grown FAB, bit for bit
        ctx.check(ctx.lib.pa_iso_coords_level(ctx.h, dco.h, 1))
        ctx.sync()
        gco = dco.download()
        for b in range(lv.nboxes):
            lo = lv.boxes[b, :3] - ng
            nz, ny, nx = gco.fab(b).shape[1:]
            for d, (n_, sh) in enumerate(((nx, (1, 1, -1)), (ny, (1, -1, 1)), (nz, (-1, 1, 1)))):
                want = np.broadcast_to(((np.arange(lo[d], lo[d] + n_) + 0.5) * lv.dx[d] + lv.prob_lo[d]).reshape(sh), (nz, ny, nx))
                assert np.array_equal(gco.fab(b)[1 + d].view(np.int64), np.ascontiguousarray(want).view(np.int64)), (l, b, d)
        assert ctx.lib.pa_iso_coords_level(ctx.h, dco.h, 2) != 0  # 3 components do not fit behind comp 2
        dmask = capi.DevMF(ctx, dls[l], 1, ng)
        fine = dls[l + 1].h if l + 1 < H.nlev else None
        ctx.check(ctx.lib.pa_iso_mask_level(ctx.h, dmask.h, 0, fine, 2))
        ctx.sync()
        gmask = dmask.download()
        loops = np.zeros((lv.nboxes, 6), np.int64)
        want = []
        for b in range(lv.nboxes):
            lo, hi, mask, llo, lhi = oracle.iso_fab_inputs(H.levels, states, l, b, ng)
            assert np.array_equal(gmask.fab(b)[0], mask), f"{name} level {l} box {b}: fine-covered mask differs"
            loops[b, :3], loops[b, 3:] = llo, lhi
            if b == 1 and lv.nboxes > 2:
                loops[b, 3] = loops[b, 0] - 1  # an empty loop box is skipped
                want.append((np.zeros((0, nc)), np.zeros((0, 6), np.int32), np.zeros((0, 3), np.int32)))
                continue
            want.append(oracle.mc_fab(np.ascontiguousarray(states[l].fab(b)), mask, lo, hi, 3, iso, llo, lhi))
        got = capi.mc_level(ctx, dst, dmask, loops, 3, iso)
        got_fine = capi.mc_level(ctx, dst, dls[l + 1] if l + 1 < H.nlev else None, loops, 3, iso)  # mask evaluated in the cell pass
        for (a1, a2, a3), (b1, b2, b3) in zip(got, got_fine):
            assert np.array_equal(a1.view(np.int64), b1.view(np.int64)) and np.array_equal(a2, b2) and np.array_equal(a3, b3), "pa_mc_level_fine differs"
        for b in range(lv.nboxes):
            (v, k, t), (gv, gk, gt) = want[b], got[b]
            assert (len(gv), len(gt)) == (len(v), len(t)), f"{name} level {l} box {b}: counts differ"
            assert np.array_equal(gk, k) and np.array_equal(gt, t), f"{name} level {l} box {b}: keys / connectivity differ"
            assert np.array_equal(gv.view(np.int64), np.ascontiguousarray(v).view(np.int64)), f"{name} level {l} box {b}: vertex data not bit-identical"
            ntri_total += len(t)
            if len(gt):
                frags.append((gv, gt))
    assert ntri_total > 100
    # the global node / element sets on the device (pa_iso_merge) against the oracle's sequential insertion
    wn, we = oracle.iso_merge(frags, nc)
    gn, ge = capi.iso_merge(ctx, frags, nc)
    assert len(wn) < sum(len(v) for v, _ in frags), "the case has no vertex shared between FABs"
    assert gn.shape == wn.shape and np.array_equal(gn.view(np.int64), np.ascontiguousarray(wn).view(np.int64)), f"{name}: merged nodes differ"
    assert np.array_equal(ge, we), f"{name}: merged elements differ"


@pytest.mark.parametrize("per", [(0, 0, 0), (1, 0, 0)])
def test_marching_squares_level_matches_oracle(ctx, oracle, per):
    """pa_msq_level (AMREX_SPACEDIM == 2: Segmentise, isosurface.cpp:303-406) on a 2-level 2-D hierarchy stored as one
    plane of cells: per FAB the vertices (bit for bit, vertCache order), edge keys and segments of the Python restatement"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
    ng, nc = 1, 4
    l0 = Level(chop_box((0, 0, 0), (15, 15, 0), 8), (0, 0, 0), (15, 15, 0), np.asarray(per), np.zeros(3), np.ones(3))
    l1 = Level(chop_box((8, 8, 0), (23, 23, 0), 8), (0, 0, 0), (31, 31, 0), np.asarray(per), np.zeros(3), np.ones(3))
    H = Hierarchy([l0, l1], 2)
    rng = np.random.default_rng(8)
    states = []
    for l, lv in enumerate(H.levels):
        st = MultiFab(lv, nc, ng, fill=-666.0)
        for b in range(lv.nboxes):
            f = st.fab(b)
            lo = lv.boxes[b, :3] - ng
            nz, ny, nx = f.shape[1:]
            x = ((np.arange(lo[0], lo[0] + nx) + 0.5) * lv.dx[0] + lv.prob_lo[0])[None, None, :]
            y = ((np.arange(lo[1], lo[1] + ny) + 0.5) * lv.dx[1] + lv.prob_lo[1])[None, :, None]
            f[0], f[1] = x, y
            v = st.valid(b)
            xv, yv = x[:, :, ng:-ng], y[:, ng:-ng, :]
            v[2] = 1000.0 + 500.0 * np.sin(2 * np.pi * xv) * np.cos(2 * np.pi * yv) + 200.0 * (yv - 0.5) + 1e-3 * rng.standard_normal(v[2].shape)
            v[3] = xv * yv
        oracle.fill_boundary(st, 0, nc, ng)
        if l > 0:
            assert oracle.lib().orc_fillpatch_two_levels(C.byref(oracle._mf(st)), C.byref(oracle._mf(states[l - 1])), 0, nc, ng, 2, 0) == 0
        states.append(st)
    iso = 1040.0
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    nseg_total = 0
    for l, lv in enumerate(H.levels):
        dst = capi.DevMF.from_host(ctx, dls[l], states[l])
        dmask = capi.DevMF(ctx, dls[l], 1, ng)
        ctx.check(ctx.lib.pa_iso_mask_level(ctx.h, dmask.h, 0, dls[l + 1].h if l + 1 < H.nlev else None, 2))
        ctx.sync()
        gmask = dmask.download()
        loops = np.zeros((lv.nboxes, 6), np.int64)
        want = []
        pg = ng * np.asarray(per)
        for b in range(lv.nboxes):
            lo, hi = lv.boxes[b, :3] - ng, lv.boxes[b, 3:] + ng
            llo = np.maximum(lo, np.asarray(lv.domlo) - pg)
            lhi = np.minimum(hi, np.asarray(lv.domhi) + pg) - 1
            llo[2] = lhi[2] = 0
            loops[b, :3], loops[b, 3:] = llo, lhi
            s2 = np.ascontiguousarray(states[l].fab(b)[:, ng])   # the plane k = 0
            m2 = np.ascontiguousarray(gmask.fab(b)[0, ng])
            want.append(oracle.msq_fab(s2, m2, lo[:2], hi[:2], 2, iso, llo[:2], lhi[:2]))
        if l == 0:
            assert sum((w[1].size > 0) for w in want) > 0 and any((gmask.fab(b)[0, ng] < 0).any() for b in range(lv.nboxes))
        got = capi.mc_level(ctx, dst, dmask, loops, 2, iso, squares=True)
        got_fine = capi.mc_level(ctx, dst, dls[l + 1] if l + 1 < H.nlev else None, loops, 2, iso, squares=True)
        for (a1, a2, a3), (b1, b2, b3) in zip(got, got_fine):
            assert np.array_equal(a1.view(np.int64), b1.view(np.int64)) and np.array_equal(a2, b2) and np.array_equal(a3, b3), "pa_msq_level_fine differs"
        for b in range(lv.nboxes):
            (v, k, sg), (gv, gk, gt) = want[b], got[b]
            assert (len(gv), len(gt)) == (len(v), len(sg)), f"level {l} box {b}: counts differ"
            assert np.array_equal(gk[:, [0, 1, 3, 4]], k) and (gk[:, [2, 5]] == 0).all(), f"level {l} box {b}: edge keys / vertex order differ"
            assert np.array_equal(gt[:, :2], sg) and (gt[:, 2] == -1).all(), f"level {l} box {b}: segments differ"
            assert np.array_equal(gv.view(np.int64), np.ascontiguousarray(v).view(np.int64)), f"level {l} box {b}: vertex data not bit-identical"
            nseg_total += len(sg)
    assert nseg_total > 60


def test_iso_merge_synthetic_clusters(ctx, oracle):
    """pa_iso_merge on hand-made fragments: exact copies and copies 1-3 ulp away in later fragments (in either direction),
    nodes on and next to the faces of the 1e-14 hash cells (the neighbour probes), an element that collapses once its nodes
    merge, the same element arriving from two fragments in two rotations; then a chain a ~ b ~ c with a !~ c, which the
    device path must hand back (code 2) instead of guessing"""
    rng = np.random.default_rng(17)
    nc = 5
    base = np.concatenate([rng.random((400, 3)), 0.25 + 1.0e-14 * rng.integers(0, 50, (60, 3)), 0.5 + 1.0e-14 * rng.integers(0, 4, (60, 3)) + rng.choice([0.0, 1e-16, -1e-16], (60, 3))])
    base = np.unique(base, axis=0)
    rng.shuffle(base)

    def data(p):  # node data is carried along from the FIRST copy: make the copies distinguishable
        return np.concatenate([p, rng.random((len(p), nc - 3))], axis=1)

    frags = []
    n0 = len(base)
    t0 = rng.integers(0, n0, (900, 3)).astype(np.int32)
    frags.append((data(base), t0))
    for rep in range(3):
        pick = rng.choice(n0, 150, replace=False)
        p = base[pick].copy()
        ulps = rng.integers(-3, 4, p.shape)
        for _ in range(3):
            p = np.where(ulps > 0, np.nextafter(p, 2.0), np.where(ulps < 0, np.nextafter(p, -1.0), p))
            ulps = ulps - np.sign(ulps)
        extra = rng.random((40, 3))
        pts = np.concatenate([p, extra])
        rng.shuffle(pts)
        t = rng.integers(0, len(pts), (500, 3)).astype(np.int32)
        frags.append((data(pts), t))
    # the first fragment's elements again, rotated, through a fragment that holds exact copies of its nodes
    frags.append((data(base), np.roll(t0[:200], 1, axis=1)))
    frags.append((np.zeros((0, nc)), np.zeros((0, 3), np.int32)))
    wn, we = oracle.iso_merge(frags, nc)
    got = capi.iso_merge(ctx, frags, nc)
    assert got is not None
    gn, ge = got
    assert len(wn) < sum(len(v) for v, _ in frags) - 300
    assert gn.shape == wn.shape and np.array_equal(gn.view(np.int64), np.ascontiguousarray(wn).view(np.int64))
    assert np.array_equal(ge, we) and len(ge) > 1500
    # not transitive: b is within 1e-15 of a and of c, a and c are 1.4e-15 apart
    a = np.array([0.3, 0.3, 0.3])
    chain = np.stack([a, a + [7e-16, 0, 0], a + [14e-16, 0, 0]])
    assert np.linalg.norm(chain[1] - chain[0]) < 1e-15 and np.linalg.norm(chain[2] - chain[1]) < 1e-15 and np.linalg.norm(chain[2] - chain[0]) > 1e-15
    f2 = [(data(chain), np.array([[0, 1, 2]], np.int32))]
    assert capi.iso_merge(ctx, f2, nc) is None
    wn2, _ = oracle.iso_merge(f2, nc)
    assert len(wn2) == 2  # the sequential rule: c is compared with the kept node a only
    assert capi.iso_merge(ctx, [], nc)[0].shape == (0, nc)


def test_level_entry_points_reject_bad_arguments(ctx):
    """shape / range checks happen on the host before any launch touches memory: every call returns non-zero with a
    message and leaves the output pointers NULL"""
    from peleanalysis_amd.hierarchy import Level, chop_box
    lv = Level(chop_box((0, 0, 0), (15, 15, 15), 8), (0, 0, 0), (15, 15, 15), (0, 0, 0), (0, 0, 0), (1, 1, 1))
    dl = capi.DevLevel(ctx, lv)
    st, m1, m2 = capi.DevMF(ctx, dl, 5, 1), capi.DevMF(ctx, dl, 1, 1), capi.DevMF(ctx, dl, 1, 2)
    loops = (capi.PaBox * lv.nboxes)()
    for b in range(lv.nboxes):
        for d in range(3):
            loops[b].lo[d], loops[b].hi[d] = int(lv.boxes[b, d]), int(lv.boxes[b, 3 + d]) - 1
    nv, nt = (C.c_int64 * lv.nboxes)(), (C.c_int64 * lv.nboxes)()
    pv, pk, pt = C.c_void_p(123), C.c_void_p(123), C.c_void_p(123)

    def call(fn, *a):
        pv.value = pk.value = pt.value = 123
        rc = fn(ctx.h, *a, nv, nt, C.byref(pv), C.byref(pk), C.byref(pt))
        return rc, ctx.lib.pa_last_error(ctx.h).decode()

    rc, msg = call(ctx.lib.pa_mc_level, st.h, m2.h, 0, loops, 3, 1.0)  # ghost widths differ
    assert rc != 0 and "ghost width" in msg and pv.value is None
    rc, msg = call(ctx.lib.pa_mc_level, st.h, m1.h, 1, loops, 3, 1.0)  # mask component
    assert rc != 0 and "component" in msg
    rc, msg = call(ctx.lib.pa_mc_level, st.h, m1.h, 0, loops, 7, 1.0)  # iso component
    assert rc != 0 and "component" in msg
    loops[2].hi[1] = int(lv.boxes[2, 4]) + 1  # base points whose cube leaves the grown FAB
    rc, msg = call(ctx.lib.pa_mc_level, st.h, m1.h, 0, loops, 3, 1.0)
    assert rc != 0 and "loop box" in msg and pv.value is None
    rc, msg = call(ctx.lib.pa_mc_level_fine, st.h, dl.h, 0, loops, 3, 1.0)  # ratio
    assert rc != 0 and "ratio" in msg
    loops[2].hi[1] = int(lv.boxes[2, 4]) - 1
    rc, msg = call(ctx.lib.pa_msq_level, st.h, m1.h, 0, loops, 3, 1.0)  # squares want one plane of base points
    assert rc != 0 and "one plane" in msg
    assert ctx.lib.pa_iso_mask_level(ctx.h, m1.h, 1, None, 2) != 0
    assert ctx.lib.pa_iso_coords_level(ctx.h, m1.h, 0) != 0  # 3 components do not fit
    w = (C.c_double * 5)(0.125, 0.25, 0.25, 0.25, 0.125)
    assert ctx.lib.pa_boxfilter_level2d(ctx.h, m1.h, m1.h, 0, 1, 2, w) != 0 and "ghost" in ctx.lib.pa_last_error(ctx.h).decode()
    # 2-D levels refuse the options the 2-D build of the reference does not have
    out = capi.DevMF(ctx, dl, 18, 0)
    s2 = capi.DevMF(ctx, dl, 4, 2)
    with pytest.raises(capi.PaError, match="2-D"):
        capi.curvature_run(ctx, [s2], 0, capi.bc_from_flags((0, 0, 0)), capi.curv_params(prog_min=0.0, prog_max=1.0, do_gauss=True, spacedim=2), [out], 0)
    # and a well-formed call still works afterwards
    rc, _ = call(ctx.lib.pa_mc_level_fine, st.h, None, 2, loops, 3, 1.0e30)
    assert rc == 0
    if pv.value:
        ctx.lib.pa_device_free(ctx.h, pv)


@pytest.mark.parametrize("name,ng", [("amr3_wall_z", 1), ("amr2_allwalls_ragged", 2), ("amr3_sym_x", 1), ("amr5_wall_z", 1)])  # 5 levels: level by level inside the call
def test_marching_cubes_hierarchy_call_matches_oracle_and_level_calls(ctx, oracle, name, ng):
    """pa_mc_hierarchy_fine (all levels in one call: one count read-back, one pooled output block) against the oracle's per-FAB
    Polygonise loop and against pa_mc_level_fine level by level -- vertices bit for bit, keys and connectivity identical; a
    level whose loop boxes are all empty, and a second call that reuses the cached block"""
    from util import build_config, make_states
    H, per, sym, fn = build_config(name)
    fields = make_states(H, 2, 0, fn, seed=12)
    nc = 5
    states = []
    for l, lv in enumerate(H.levels):
        st = MultiFab(lv, nc, ng, fill=-666.0)
        for b in range(lv.nboxes):
            f = st.fab(b)
            lo = lv.boxes[b, :3] - ng
            nz, ny, nx = f.shape[1:]
            f[0] = ((np.arange(lo[0], lo[0] + nx) + 0.5) * lv.dx[0] + lv.prob_lo[0])[None, None, :]
            f[1] = ((np.arange(lo[1], lo[1] + ny) + 0.5) * lv.dx[1] + lv.prob_lo[1])[None, :, None]
            f[2] = ((np.arange(lo[2], lo[2] + nz) + 0.5) * lv.dx[2] + lv.prob_lo[2])[:, None, None]
            st.valid(b)[3:5] = fields[l].valid(b)[0:2]
        oracle.fill_boundary(st, 0, nc, ng)
        if l > 0:
            assert oracle.lib().orc_fillpatch_two_levels(C.byref(oracle._mf(st)), C.byref(oracle._mf(states[l - 1])), 0, nc, ng, 2, 0) == 0
        states.append(st)
    iso = float(np.median(np.concatenate([s.valid(b)[3].ravel() for s in states for b in range(s.level.nboxes)])))
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    loops, want = [], []
    for l, lv in enumerate(H.levels):
        lp, wl = np.zeros((lv.nboxes, 6), np.int64), []
        for b in range(lv.nboxes):
            lo, hi, mask, llo, lhi = oracle.iso_fab_inputs(H.levels, states, l, b, ng)
            lp[b, :3], lp[b, 3:] = llo, lhi
            wl.append(oracle.mc_fab(np.ascontiguousarray(states[l].fab(b)), mask, lo, hi, 3, iso, llo, lhi))
        loops.append(lp)
        want.append(wl)
    fm = [1] * (H.nlev - 1) + [0]
    for rep in range(2):  # the second call takes the output block from the context's cache
        got = capi.mc_hierarchy(ctx, dst, fm, loops, 3, iso)
        ntri = 0
        for l, lv in enumerate(H.levels):
            lev = capi.mc_level(ctx, dst[l], dls[l + 1] if l + 1 < H.nlev else None, loops[l], 3, iso)
            for b in range(lv.nboxes):
                (v, k, t), (gv, gk, gt), (lv_, lk, lt) = want[l][b], got[l][b], lev[b]
                assert (len(gv), len(gt)) == (len(v), len(t)), f"{name} level {l} box {b}: counts differ"
                assert np.array_equal(gk, k) and np.array_equal(gt, t), f"{name} level {l} box {b}: keys / connectivity differ"
                assert np.array_equal(gv.view(np.int64), np.ascontiguousarray(v).view(np.int64)), f"{name} level {l} box {b}: vertex data not bit-identical"
                assert np.array_equal(gv.view(np.int64), lv_.view(np.int64)) and np.array_equal(gk, lk) and np.array_equal(gt, lt), "hierarchy call != level call"
                ntri += len(t)
        assert ntri > 100
    # the finest level switched off (every loop box empty) and an iso value nothing crosses: null parts, no block
    off = [lp.copy() for lp in loops]
    off[-1][:, 3] = off[-1][:, 0] - 1
    got = capi.mc_hierarchy(ctx, dst, fm, off, 3, iso)
    assert all(len(t) == 0 for (_, _, t) in got[-1]) and sum(len(t) for lev in got[:-1] for (_, _, t) in lev) > 0
    got = capi.mc_hierarchy(ctx, dst, fm, loops, 3, 1.0e30)
    assert all(len(t) == 0 and len(v) == 0 for lev in got for (v, _, t) in lev)


def _ratio4_hierarchy(per):
    """2 levels, refinement ratio 4: base 24^3 in 12^3 boxes; level 1 = coarse cells [6, 15] x [6, 17] x [8, 15] refined (x4) in boxes <= 24"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
    l0 = Level(chop_box((0, 0, 0), (23, 23, 23), 12), (0, 0, 0), (23, 23, 23), per, np.zeros(3), np.ones(3))
    l1 = Level(chop_box((24, 24, 32), (63, 71, 63), 24), (0, 0, 0), (95, 95, 95), per, np.zeros(3), np.ones(3))
    return Hierarchy([l0, l1], 4)


@pytest.mark.parametrize("interp,ng", [(0, 1), (1, 2), (1, 4), (0, 3)])
def test_fillpatch_two_levels_refinement_ratio_4(ctx, oracle, interp, ng):
    """FillPatchTwoLevels with the plotfile's refinement ratio (filterPlt.cpp:193-200, isosurface.cpp:1474-1478,1515-1524), here 4:
    piecewise constant (the parent of 4^3 children) and cell-conservative linear (offsets +-1/8, +-3/8; the limiter's common
    factor with (r - 1) / (2 r) = 3/8) against the oracle, every ghost cell bit for bit"""
    from util import make_states, field_flame
    H = _ratio4_hierarchy((1, 0, 1))
    src = make_states(H, 2, 0, field_flame, seed=9)
    mfs = []
    for l, lv in enumerate(H.levels):
        m = MultiFab(lv, 2, ng, fill=-666.0)
        for b in range(lv.nboxes):
            m.valid(b)[:] = src[l].valid(b)
        mfs.append(m)
    om = [m.copy() for m in mfs]
    for l in range(2):
        oracle.fill_boundary(om[l], 0, 2, ng)
        if l > 0:
            assert oracle.lib().orc_fillpatch_two_levels(C.byref(oracle._mf(om[l])), C.byref(oracle._mf(om[l - 1])), 0, 2, ng, 4, interp) == 0
        oracle.foextrap(om[l], 0, 2, ng)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dm = [capi.DevMF.from_host(ctx, dl, m) for dl, m in zip(dls, mfs)]
    for l in range(2):
        ctx.check(ctx.lib.pa_fill_boundary(ctx.h, dm[l].h, 0, 2, ng))
        if l > 0:
            ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, dm[l].h, dm[l - 1].h, 0, 2, ng, 4, interp))
        ctx.check(ctx.lib.pa_foextrap(ctx.h, dm[l].h, 0, 2, ng))
    ctx.sync()
    assert ctx.bc_errors() == 0
    for l, lv in enumerate(H.levels):
        got = dm[l].download()
        for b in range(lv.nboxes):
            assert np.array_equal(got.fab(b).view(np.int64), om[l].fab(b).view(np.int64)), f"ratio 4 interp {interp} ng {ng}: level {l} box {b} (ghost cells included)"
    assert ctx.lib.pa_fillpatch_two_levels(ctx.h, dm[1].h, dm[0].h, 0, 2, ng, 1, interp) != 0  # ratio 1 is refused
    ctx.lib.pa_last_error(ctx.h)


def test_marching_cubes_refinement_ratio_4(ctx, oracle):
    """the fine-covered mask with the finer level coarsened by 4 (isosurface.cpp:1543) evaluated in the cell pass -- level by
    level and through pa_mc_hierarchy_fine -- and the mask multifab form, against the oracle's per-FAB loop"""
    from util import make_states, field_flame
    H = _ratio4_hierarchy((0, 0, 0))
    fields = make_states(H, 1, 0, field_flame, seed=5)
    ng, nc = 1, 4
    states = []
    for l, lv in enumerate(H.levels):
        st = MultiFab(lv, nc, ng, fill=-666.0)
        for b in range(lv.nboxes):
            f = st.fab(b)
            lo = lv.boxes[b, :3] - ng
            nz, ny, nx = f.shape[1:]
            f[0] = ((np.arange(lo[0], lo[0] + nx) + 0.5) * lv.dx[0] + lv.prob_lo[0])[None, None, :]
            f[1] = ((np.arange(lo[1], lo[1] + ny) + 0.5) * lv.dx[1] + lv.prob_lo[1])[None, :, None]
            f[2] = ((np.arange(lo[2], lo[2] + nz) + 0.5) * lv.dx[2] + lv.prob_lo[2])[:, None, None]
            st.valid(b)[3] = fields[l].valid(b)[0]
        oracle.fill_boundary(st, 0, nc, ng)
        if l > 0:
            assert oracle.lib().orc_fillpatch_two_levels(C.byref(oracle._mf(st)), C.byref(oracle._mf(states[l - 1])), 0, nc, ng, 4, 0) == 0
        states.append(st)
    iso = 1150.0
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    loops, want = [], []
    for l, lv in enumerate(H.levels):
        lp, wl = np.zeros((lv.nboxes, 6), np.int64), []
        for b in range(lv.nboxes):
            lo, hi, mask, llo, lhi = oracle.iso_fab_inputs(H.levels, states, l, b, ng, ratio=4)
            lp[b, :3], lp[b, 3:] = llo, lhi
            wl.append(oracle.mc_fab(np.ascontiguousarray(states[l].fab(b)), mask, lo, hi, 3, iso, llo, lhi))
            if l == 0:  # the mask multifab of pa_iso_mask_level with ratio 4
                pass
        loops.append(lp)
        want.append(wl)
    dmask = capi.DevMF(ctx, dls[0], 1, ng)
    ctx.check(ctx.lib.pa_iso_mask_level(ctx.h, dmask.h, 0, dls[1].h, 4))
    ctx.sync()
    gm = dmask.download()
    for b in range(H.levels[0].nboxes):
        assert np.array_equal(gm.fab(b)[0], oracle.iso_fab_inputs(H.levels, states, 0, b, ng, ratio=4)[2]), f"ratio 4 mask box {b}"

    def check(got, what):
        nt = 0
        for l, lv in enumerate(H.levels):
            for b in range(lv.nboxes):
                (v, k, t), (gv, gk, gt) = want[l][b], got[l][b]
                assert (len(gv), len(gt)) == (len(v), len(t)), f"{what} level {l} box {b}: counts differ"
                assert np.array_equal(gk, k) and np.array_equal(gt, t) and np.array_equal(gv.view(np.int64), np.ascontiguousarray(v).view(np.int64)), f"{what} level {l} box {b}"
                nt += len(t)
        assert nt > 200

    # level by level with ratio 4 (capi.mc_level passes 2: call the C entry directly)
    got = []
    for l, lv in enumerate(H.levels):
        nb = lv.nboxes
        arr = (capi.PaBox * nb)()
        for b in range(nb):
            for d in range(3):
                arr[b].lo[d], arr[b].hi[d] = int(loops[l][b, d]), int(loops[l][b, 3 + d])
        nv, nt = (C.c_int64 * nb)(), (C.c_int64 * nb)()
        pv, pk, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ctx.check(ctx.lib.pa_mc_level_fine(ctx.h, dst[l].h, dls[1].h if l == 0 else None, 4, arr, 3, iso, nv, nt, C.byref(pv), C.byref(pk), C.byref(pt)))
        tv, tt = int(sum(nv[:nb])), int(sum(nt[:nb]))
        V = np.empty((tv, nc)); K = np.empty((tv, 6), np.int32); T = np.empty((tt, 3), np.int32)
        if tv:
            ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, V.ctypes.data_as(C.c_void_p), pv, V.nbytes))
            ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, K.ctypes.data_as(C.c_void_p), pk, K.nbytes))
        if tt:
            ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, T.ctypes.data_as(C.c_void_p), pt, T.nbytes))
        if pv.value:
            ctx.lib.pa_device_free(ctx.h, pv)
        lev, ov, ot = [], 0, 0
        for b in range(nb):
            lev.append((V[ov:ov + nv[b]], K[ov:ov + nv[b]], T[ot:ot + nt[b]]))
            ov += nv[b]; ot += nt[b]
        got.append(lev)
    check(got, "pa_mc_level_fine ratio 4")
    check(capi.mc_hierarchy(ctx, dst, [1, 0], loops, 3, iso, ratio=4), "pa_mc_hierarchy_fine ratio 4")


@pytest.mark.parametrize("parent", ["1", "0"])
def test_fillpatch_not_properly_nested_counts_errors_without_faulting(parent):
    """a fine box whose ghost parents lie outside the coarse level (not properly nested): FillPatchTwoLevels with the
    cell-conservative interpolation must COUNT those ghost cells (pa_bc_errors > 0; the tools turn that into the reference's
    abort) on both forms -- the cached parent list (k_fp_do: advisor finding of round 3, it used to index the coarse multifab
    with an unrelated cell) and the per-ghost-cell kernel -- and never read out of bounds.  Own process: the switch is read once."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import Hierarchy, Level, MultiFab
z, o = np.zeros(3), np.ones(3)
l0 = Level(np.array([[8, 8, 8, 23, 23, 23]], np.int32), (0, 0, 0), (31, 31, 31), (0, 0, 0), z, o)       # the coarse level covers the middle only
l1 = Level(np.array([[16, 16, 16, 47, 47, 47]], np.int32), (0, 0, 0), (63, 63, 63), (0, 0, 0), z, o)   # fine box flush with the coarse level: ghost parents at coarse 7 do not exist
ctx = capi.Context(0)
rng = np.random.default_rng(1)
ms = []
for lv in (l0, l1):
    m = MultiFab(lv, 1, 2)
    m.data[:] = rng.random(m.total)
    ms.append(m)
dls = [capi.DevLevel(ctx, lv) for lv in (l0, l1)]
dm = [capi.DevMF.from_host(ctx, dl, m) for dl, m in zip(dls, ms)]
for it in (1, 0):
    ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, dm[1].h, dm[0].h, 0, 1, 2, 2, it))
    ctx.sync()
    n = ctx.bc_errors()
    assert n > 0, (it, n)
print("NESTED_ERRORS_COUNTED")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, PA_FORCE_FALLBACKS="0" if parent == "1" else "1"))
    assert r.returncode == 0 and "NESTED_ERRORS_COUNTED" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def _iso_loops(H, ng, per):
    """cube base points per FAB as isosurface.cpp:1566-1569: (grown box & domain grown in the periodic directions), high side - 1"""
    loops = []
    for lv in H.levels:
        lp = np.zeros((lv.nboxes, 6), np.int64)
        for b in range(lv.nboxes):
            for d in range(3):
                pg = ng if per[d] else 0
                lp[b, d] = max(int(lv.boxes[b, d]) - ng, int(lv.domlo[d]) - pg)
                lp[b, 3 + d] = min(int(lv.boxes[b, 3 + d]) + ng, int(lv.domhi[d]) + pg) - 1
        loops.append(lp)
    return loops


@pytest.mark.parametrize("case", ["amr3_wall_z", "amr3_sym_x", "periodic", "union0", "union1", "union2", "ratio4"])
@pytest.mark.parametrize("ng", [1, 2])
def test_marching_cubes_with_analytic_coordinates_equals_the_stored_coordinate_state(ctx, oracle, case, ng):
    """pa_mc_hierarchy_xyz (states = fields only, ghost cells by pa_fill_ghosts_hierarchy, vertex coordinates formed from cell
    indices) against pa_mc_hierarchy_fine on the reference-shaped state (three stored coordinate components, FillBoundary +
    FillPatchTwoLevels on them as isosurface.cpp:1458-1478; that path is compared with the oracle's Polygonise above): vertices
    bit for bit, keys and connectivity identical -- nested and non-convex hierarchies, periodic images (quirk Q5: wrapped
    coordinates), coarse-fine ghost cells (coarse cell centres), one or two ghost layers, refinement ratio 2 and 4."""
    from util import build_config, make_states
    from peleanalysis_amd.hierarchy import field_flame, field_trig, union_hierarchy
    ratio = 2
    if case in ("amr3_wall_z", "amr3_sym_x"):
        H, per, _, fn = build_config(case)
    elif case == "periodic":
        from peleanalysis_amd.hierarchy import nested_hierarchy
        H, per, fn = nested_hierarchy(32, 3, 8, is_per=(1, 1, 1)), (1, 1, 1), field_trig
    elif case == "ratio4":
        H, per, fn, ratio = _ratio4_hierarchy((1, 0, 0)), (1, 0, 0), field_flame, 4
    else:
        H = union_hierarchy(7100 + int(case[-1]))
        per, fn = tuple(int(v) for v in H.levels[0].is_per), field_flame
    nf = 2
    fields = make_states(H, nf, 0, fn, seed=3)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    # reference-shaped state on the device, the way tools/src/isosurface.cpp built it up to round 4
    ref, fld = [], []
    for l, (lv, dl) in enumerate(zip(H.levels, dls)):
        hf = MultiFab(lv, nf, ng, fill=-666.0)
        for b in range(lv.nboxes):
            hf.valid(b)[:] = fields[l].valid(b)
        df = capi.DevMF.from_host(ctx, dl, hf)
        st = capi.DevMF(ctx, dl, 3 + nf, ng)
        ctx.check(ctx.lib.pa_iso_coords_level(ctx.h, st.h, 0))
        ctx.check(ctx.lib.pa_mf_copy(ctx.h, df.h, 0, st.h, 3, nf, ng))
        ctx.check(ctx.lib.pa_fill_boundary(ctx.h, st.h, 0, 3 + nf, ng))
        if l > 0:
            ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, st.h, ref[l - 1].h, 0, 3 + nf, ng, ratio, 0))
        ref.append(st)
        fld.append(capi.DevMF.from_host(ctx, dl, hf))
    hm = (C.c_void_p * H.nlev)(*[f.h for f in fld])
    hg = (C.c_int32 * H.nlev)(*([ng] * H.nlev))
    ctx.check(ctx.lib.pa_fill_ghosts_hierarchy(ctx.h, H.nlev, hm, 0, nf, hg, ratio, 0, 0))
    ctx.sync()
    assert ctx.bc_errors() == 0
    # the ghost fill of the hierarchy in three launches == the per-level calls (the field components of the reference-shaped state)
    for l, lv in enumerate(H.levels):
        a, b_ = fld[l].download(), ref[l].download()
        for b in range(lv.nboxes):
            fa, fb = a.fab(b), b_.fab(b)[3:]
            inside = np.ones(fa.shape[1:], bool)  # ghost cells beyond a non-periodic wall are never read: compare the others
            for d, ax in ((0, 2), (1, 1), (2, 0)):
                if not per[d]:
                    idx = np.arange(fa.shape[1 + ax]) + int(lv.boxes[b, d]) - ng
                    m = (idx >= lv.domlo[d]) & (idx <= lv.domhi[d])
                    sh = [1, 1, 1]
                    sh[ax] = -1
                    inside &= m.reshape(sh)
            assert np.array_equal(fa[:, inside].view(np.int64), fb[:, inside].view(np.int64)), f"{case} ng {ng}: ghost fill of level {l} box {b} differs"
    iso = float(np.median(np.concatenate([f.valid(b)[0].ravel() for f in fields for b in range(f.level.nboxes)])))
    loops = _iso_loops(H, ng, per)
    fm = [1] * (H.nlev - 1) + [0]
    want = capi.mc_hierarchy(ctx, ref, fm, loops, 3, iso, ratio=ratio)
    got = capi.mc_hierarchy(ctx, fld, fm, loops, 0, iso, ratio=ratio, xyz=True)
    ntri = 0
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            (v, k, t), (gv, gk, gt) = want[l][b], got[l][b]
            assert (len(gv), len(gt)) == (len(v), len(t)), f"{case} ng {ng} level {l} box {b}: counts differ"
            assert np.array_equal(gk, k) and np.array_equal(gt, t), f"{case} ng {ng} level {l} box {b}: keys / connectivity differ"
            assert np.array_equal(gv.view(np.int64), v.view(np.int64)), f"{case} ng {ng} level {l} box {b}: vertex data (coordinates from indices) not bit-identical"
            ntri += len(t)
    assert ntri > 50
