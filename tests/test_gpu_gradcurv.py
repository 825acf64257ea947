"""GPU parity: HIP grad / curvature path (through the C ABI) vs the CPU oracle, bit for bit.

Tolerance stated by north_star: 1e-12 relative (metric of SURVEY 8d); because kernels and oracle
use the same operation order with contraction off the tests demand 0 ulp (bit equality) and
report the relative error only on failure."""
import numpy as np
import pytest

from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import MultiFab
from util import CONFIGS, assert_valid_bits_equal, build_config, make_states, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-12


def _dev(ctx, H, states):
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dms = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    return dls, dms


@pytest.mark.parametrize("name", list(CONFIGS))
def test_grad_run_matches_oracle(ctx, oracle, name):
    H, per, sym, fn = build_config(name)
    states = make_states(H, 1, 1, fn, seed=3)
    bc = capi.bc_from_flags(per, sym)
    # oracle (reference-shaped multipass)
    ost = [s.copy() for s in states]
    oout = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, ost, 0, bc, oout, 0, multipass=True)
    # HIP
    dls, dst = _dev(ctx, H, states)
    dout = [capi.DevMF(ctx, dl, 4, 0) for dl in dls]
    capi.grad_run(ctx, dst, 0, bc, dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for l in range(H.nlev):
        got = dout[l].download()
        assert_valid_bits_equal(got, oout[l], [(c, c) for c in range(4)], f"{name} level {l}")
        for c in range(4):
            assert rel_err(got, oout[l], c, c) <= TOL


@pytest.mark.parametrize("threshold", [None, 0.05])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_curvature_run_pass_by_pass_matches_oracle(ctx, oracle, name, threshold):
    H, per, sym, fn = build_config(name)
    states = make_states(H, 1, 2, fn, seed=5)
    bc = capi.bc_from_flags(per, sym)
    oout = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oout, 0, MultiFab, threshold=threshold)
    dls, dst = _dev(ctx, H, states)
    dout = [capi.DevMF(ctx, dl, 5, 0) for dl in dls]
    capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(threshold=threshold, fused=False), dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for l in range(H.nlev):
        assert_valid_bits_equal(dout[l].download(), oout[l], [(c, c) for c in range(5)], f"{name} level {l}")


@pytest.mark.parametrize("threshold", [None, 0.05])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_gradcurv_fused_matches_oracle(ctx, oracle, name, threshold):
    """fused headline path == grad tool + curvature tool run separately (oracle), bit for bit"""
    H, per, sym, fn = build_config(name)
    states = make_states(H, 1, 2, fn, seed=7)
    bc = capi.bc_from_flags(per, sym)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=True)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=threshold)
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=threshold, fused=True), work, dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for l in range(H.nlev):
        got = dout[l].download()
        assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"{name} grad level {l}")
        # out: 4..6 normal, 7 K ; oracle: 2..4 normal, 1 K
        assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"{name} curv level {l}")


@pytest.mark.parametrize("pipeline", ["exact", "first"])
@pytest.mark.parametrize("threshold", [None, 0.05])
def test_gradcurv_fused_wide_boxes(ctx, oracle, threshold, pipeline, options):
    """boxes 64 cells wide with a partial last row tile (48 = 3*13 + 9 rows), anisotropic dx, 2 levels, periodic x/y + wall z,
    on the exact-normal pipeline and (PA_FUSED2=0) on the first fused pipeline"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box, field_flame
    if pipeline == "first":
        options(PA_FUSED2=0)
    l0 = Level(chop_box((0, 0, 0), (127, 47, 19), 64), (0, 0, 0), (127, 47, 19), (1, 1, 0), (0, 0, 0), (1, 1, 1))
    l1 = Level(chop_box((64, 24, 10), (191, 71, 29), 64), (0, 0, 0), (255, 95, 39), (1, 1, 0), (0, 0, 0), (1, 1, 1))
    H = Hierarchy([l0, l1], 2)
    assert all(lv.boxes[:, 3].max() - lv.boxes[:, 0].min() + 1 == 128 and ((lv.boxes[:, 3] - lv.boxes[:, 0] + 1) == 64).all() for lv in H.levels)
    states = make_states(H, 1, 2, field_flame, seed=23)
    bc = capi.bc_from_flags((1, 1, 0), (0, 0, 0))
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=True)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=threshold)
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=threshold, fused=True), work, dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for l in range(H.nlev):
        got = dout[l].download()
        assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"wide grad level {l}")
        assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"wide curv level {l}")


@pytest.mark.parametrize("threshold", [None, 0.05, 0.3])
@pytest.mark.parametrize("per,sym", [((1, 1, 0), (0, 0, 0)), ((0, 0, 0), (1, 0, 1)), ((0, 1, 1), (0, 0, 0))])
def test_gradcurv_exact_normal_pipeline(ctx, oracle, per, sym, threshold):
    """boxes wider than 32 cells, no threshold, pure special faces: the exact-normal pipeline (pa_fused.hip: the sweep reads
    the resolved ghost c behind coarse-fine / wall faces from compact face-major arrays; only the curvature of the first
    layer behind such a face is fixed up).  3 levels of 48^3 boxes: 4 row tiles per box (3 x 13 + 9 rows), 3 z segments of
    16 planes (interior and end segments), coarse-fine faces in all three directions, every wall type."""
    from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
    H = nested_hierarchy(96, 3, 48, is_per=per)
    states = make_states(H, 1, 2, field_flame, seed=29)
    bc = capi.bc_from_flags(per, sym)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=False)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=threshold)
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    slow_seen = []
    for rep in range(2):  # the second pass reuses the level's compact arrays
        # with a threshold the sweep clips N and K itself and the one-layer fix-up recomputes the clipped normals it needs
        # (curvature.cpp:549-570; the coarse normals under coarse-fine faces stay clipped: quirk Q2)
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=threshold, fused=True), work, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
        assert kn.endswith("CG=1>") or "_levels<" in kn, kn  # the exact-normal pipeline did run (level by level or all levels in one launch)
        if threshold is not None:
            assert "CLIP" in kn
            nslow = ctx.lib.pa_last_slow_cells(ctx.h)
            assert nslow >= 0
            slow_seen.append(nslow)
            nclip = sum(int((np.abs(oc[l].valid_concat(2)) == 0).sum()) for l in range(H.nlev))
            ncell = sum(lv.ncells for lv in H.levels)
            assert 0.05 * ncell < nclip < 0.98 * ncell  # clipped and unclipped cells both present
        for l in range(H.nlev):
            got = dout[l].download()
            assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"exact grad level {l}")
            assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"exact curv level {l}")


def test_clip_fixup_general_path_is_exercised(ctx, oracle):
    """the clip-aware fix-up's hand-over list (pa_fused.hip: SlowList): a threshold whose iso-surfaces cross the coarse-fine
    faces puts layer-1 cells next to clipped neighbours, which must go through the general path -- and still match the oracle"""
    from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
    H = nested_hierarchy(96, 3, 48, is_per=(1, 1, 0))
    states = make_states(H, 1, 2, field_flame, seed=31)
    bc = capi.bc_from_flags((1, 1, 0), (0, 0, 0))
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    total = 0
    for thr in (0.02, 0.1, 0.25, 0.4):
        oc = [MultiFab(lv, 5, 0) for lv in H.levels]
        oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=thr)
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=thr, fused=True), work, dout, 0)
        ctx.sync()
        n = ctx.lib.pa_last_slow_cells(ctx.h)
        assert n >= 0
        total += n
        for l in range(H.nlev):
            assert_valid_bits_equal(dout[l].download(), oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"thr {thr} level {l}")
    assert total > 0, "no cell ever took the general path: the case does not exercise it"


@pytest.mark.parametrize("per", [(0, 0, 0), (1, 0, 1)])
def test_gradcurv_exact_normal_pipeline_one_short_tiles(ctx, oracle, per):
    """fine boxes of 65 x 13 x 21 cells: the sweep's last full tile ends ONE column / row short of the box (64 t + 1 columns,
    4 t + 1 rows), so its outermost neighbour column / row is the last valid one and the ghost c behind the special face is
    the cell BEYOND it (mode 2 of the CG variant, pa_fused_march3.h) -- the case a randomly drawn hierarchy exposed"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, field_flame
    l0 = Level(np.array([[0, 0, 0, 79, 23, 31]]), (0, 0, 0), (79, 23, 31), per, (0, 0, 0), (1, 1, 1))
    fb = [[16 + 65 * a, 10 + 13 * b, 12 + 21 * c, 16 + 65 * a + 64, 10 + 13 * b + 12, 12 + 21 * c + 20] for c in range(2) for b in range(2) for a in range(2)]
    l1 = Level(np.array(fb), (0, 0, 0), (159, 47, 63), per, (0, 0, 0), (1, 1, 1))
    H = Hierarchy([l0, l1], 2)
    states = make_states(H, 1, 2, field_flame, seed=37)
    bc = capi.bc_from_flags(per, (0, 0, 0))
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=False)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab)
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(fused=True), work, dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
    assert kn.endswith("CG=1>") or "_levels<" in kn, kn  # the exact-normal pipeline did run (level by level or all levels in one launch)
    for l in range(H.nlev):
        got = dout[l].download()
        assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"one-short grad level {l}")
        assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"one-short curv level {l}")


def test_gradcurv_four_levels_many_components(ctx, oracle):
    """BASELINE config 5 in small: 4 levels (PeleLMeX-style nesting), several components (species-like
    fields with different phases/amplitudes) pushed through the fused grad->curvature pipeline one after
    the other into recycled output buffers, each compared with the oracle's grad + curvature tools"""
    from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
    H = nested_hierarchy(16, 4, 8, is_per=(1, 1, 0))
    ncomp = 5
    states = make_states(H, ncomp, 2, field_flame, seed=31)
    bc = capi.bc_from_flags((1, 1, 0), (0, 0, 0))
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    for c in range(ncomp):
        og = [MultiFab(lv, 4, 0) for lv in H.levels]
        oracle.grad_pipeline(H.levels, [s.copy() for s in states], c, bc, og, 0, multipass=True)
        oc = [MultiFab(lv, 5, 0) for lv in H.levels]
        oracle.curvature_pipeline(H.levels, [s.copy() for s in states], c, bc, oc, 0, MultiFab)
        capi.gradcurv_run(ctx, dst, c, bc, capi.curv_params(fused=True), work, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        for l in range(H.nlev):
            got = dout[l].download()
            assert_valid_bits_equal(got, og[l], [(k, k) for k in range(4)], f"comp {c} grad level {l}")
            assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"comp {c} curv level {l}")


@pytest.mark.parametrize("dom,maxbox,per", [
    ((47, 39, 9), (47, 13, 3), (1, 1, 1)),     # 47-wide boxes (64-lane rows with mirrored lanes), 3-cell-thin boxes in z
    ((40, 33, 14), (20, 33, 14), (1, 0, 1)),   # 20-wide boxes (narrow kernel, mirrored columns), 33 rows = 2 full tiles + 1 row
    ((96, 18, 5), (96, 18, 5), (0, 1, 0)),     # one 96-wide box: a full and a half-filled x tile
    ((31, 7, 40), (31, 7, 40), (1, 1, 0)),     # odd extents, fewer rows than a tile, many planes (several z segments at kseg 16)
])
def test_gradcurv_fused_ragged_shapes(ctx, oracle, dom, maxbox, per, monkeypatch):
    """single-level boxes of awkward extents through both sweep kernels (partial tiles in x and y, thin boxes,
    several z segments), periodic and wall boundaries"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, field_trig
    lo, hi = (0, 0, 0), tuple(d - 1 for d in dom)
    boxes = []
    for z0 in range(0, dom[2], maxbox[2]):
        for y0 in range(0, dom[1], maxbox[1]):
            for x0 in range(0, dom[0], maxbox[0]):
                boxes.append([x0, y0, z0, min(x0 + maxbox[0], dom[0]) - 1, min(y0 + maxbox[1], dom[1]) - 1, min(z0 + maxbox[2], dom[2]) - 1])
    H = Hierarchy([Level(np.array(boxes, dtype=np.int32), lo, hi, per, (0, 0, 0), (1, 1, 1))], 2)
    states = make_states(H, 1, 2, field_trig, seed=5)
    bc = capi.bc_from_flags(per, (0, 0, 0))
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=True)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=0.02)
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=0.02, fused=True), work, dout, 0)
    ctx.sync()
    got = dout[0].download()
    assert_valid_bits_equal(got, og[0], [(c, c) for c in range(4)], f"ragged {dom} grad")
    assert_valid_bits_equal(got, oc[0], [(4, 2), (5, 3), (6, 4), (7, 1)], f"ragged {dom} curv")
    # the gradient tool's own kernels (k_grad_march above 32 columns, k_grad_marchn below)
    dls2, dst2 = _dev(ctx, H, states)
    dgr = [capi.DevMF(ctx, dl, 4, 0) for dl in dls2]
    capi.grad_run(ctx, dst2, 0, bc, dgr, 0)
    ctx.sync()
    assert_valid_bits_equal(dgr[0].download(), og[0], [(c, c) for c in range(4)], f"ragged {dom} grad_run")


def test_ghost_fill_matches_oracle(ctx, oracle):
    """FillBoundary (ng=2, edges+corners) and applyBC individually, all ghost cells compared"""
    H, per, sym, fn = build_config("amr3_wall_z")
    states = make_states(H, 2, 2, fn, seed=11)
    bc = capi.bc_from_flags(per, sym)
    dls, dst = _dev(ctx, H, states)
    for l in range(H.nlev):
        oracle.fill_boundary(states[l], 0, 2, 2)
        ctx.check(ctx.lib.pa_fill_boundary(ctx.h, dst[l].h, 0, 2, 2))
    for l in range(H.nlev):
        oracle.apply_bc(states[l], 1, states[l - 1] if l else None, 1, bc)
        ctx.check(ctx.lib.pa_apply_bc(ctx.h, dst[l].h, 1, dst[l - 1].h if l else None, 1, capi._i3(bc), 2, -1))
    ctx.sync()
    for l in range(H.nlev):
        got = dst[l].download()
        a, b = got.data, states[l].data
        same = (a.view(np.int64) == b.view(np.int64)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), f"level {l}: {np.count_nonzero(~same)} cells differ"


def test_per_fab_entry_points(ctx, oracle):
    """pa_*_fab (one MFIter body each) on a single periodic box vs the level kernels' oracle"""
    H, per, sym, fn = build_config("c1_periodic_1lev")
    lv = H.levels[0]
    states = make_states(H, 1, 2, fn, seed=13)
    bc = capi.bc_from_flags(per, sym)
    oracle.fill_boundary(states[0], 0, 1, 2)
    og = MultiFab(lv, 4, 0)
    oracle.grad_fused(states[0], 0, og, 0, True)
    dl = capi.DevLevel(ctx, lv)
    dphi = capi.DevMF.from_host(ctx, dl, states[0])
    dout = capi.DevMF(ctx, dl, 4, 0)
    dxinv = capi._d3(1.0 / lv.dx)
    for b in range(lv.nboxes):
        fp, fo = dphi.fab(b), dout.fab(b)
        ctx.check(ctx.lib.pa_grad_fab(ctx.h, capi.box_of(lv, b), fp, 0, dxinv, fo, 0))
    ctx.sync()
    assert_valid_bits_equal(dout.download(), og, [(c, c) for c in range(4)], "pa_grad_fab")
    # shape checks happen on the host before any launch
    bad = capi.box_of(lv, 0, grow=2)
    assert ctx.lib.pa_grad_fab(ctx.h, bad, dphi.fab(0), 0, dxinv, dout.fab(0), 0) != 0
    assert b"cover" in ctx.lib.pa_last_error(ctx.h)


def _lshape_hierarchy():
    """level 1 = three 8^3-coarse-cell blocks in an L (concave coarse-fine corner): not fusable"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
    l0 = Level(chop_box((0, 0, 0), (31, 31, 31), 16), (0, 0, 0), (31, 31, 31), (1, 0, 1), (0, 0, 0), (1, 1, 1))
    fine = np.array([[16, 16, 16, 31, 31, 47], [32, 16, 16, 47, 31, 47], [16, 32, 16, 31, 47, 47]], dtype=np.int32)
    l1 = Level(fine, (0, 0, 0), (63, 63, 63), (1, 0, 1), (0, 0, 0), (1, 1, 1))
    return Hierarchy([l0, l1], 2)


def test_concave_coarse_fine_corner_falls_back_to_passes(ctx, oracle):
    """general AMR: a level with an L-shaped refined region cannot use the fused sweep (one edge ghost
    cell would need two boundary values); pa_gradcurv_run must take the pass-by-pass kernels and
    still match the oracle (one-sided tangential interpolation stencils are exercised here too)"""
    from util import field_flame
    H = _lshape_hierarchy()
    per, sym = (1, 0, 1), (0, 1, 0)
    states = make_states(H, 1, 2, field_flame, seed=17)
    bc = capi.bc_from_flags(per, sym)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0)
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=0.03)
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=0.03, fused=True), work, dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for l in range(H.nlev):
        got = dout[l].download()
        assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"L-shape grad level {l}")
        assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"L-shape curv level {l}")


@pytest.mark.parametrize("name", ["amr3_wall_z", "amr3_sym_x"])
def test_curvature_options_match_oracle(ctx, oracle, name):
    """do_gaussCurv, do_strain (+getStrainTensor), do_velnormal (curvature.cpp:575-789) vs the oracle"""
    H, per, sym, fn = build_config(name)
    states = make_states(H, 4, 2, fn, seed=23)  # comp 0 = progress source, 1..3 = velocity
    bc = capi.bc_from_flags(per, sym)
    oout = [MultiFab(lv, 17, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oout, 0, MultiFab, threshold=0.05, do_gauss=True, vel_comp=1,
                              do_strain=True, do_velnormal=True, strain_tensor=True)
    dls, dst = _dev(ctx, H, states)
    dout = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
    # the work multifabs of both paths live as long as the level does: a first call on OTHER data leaves them full of stale values
    other = make_states(H, 4, 2, fn, seed=77)
    for m in other:
        m.data[:] = 3.0 * m.data + 11.0
    dst_other = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, other)]
    for fused in (False, True):
        capi.curvature_run(ctx, dst_other, 0, bc, capi.curv_params(fused=fused, do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True, vel_comp=1), dout, 0)
    ctx.sync()
    # fused=False: pass by pass; fused=True: Progress / K / N from the exact-normal pipeline's G-output sweeps + one options pass per level
    for fused in (False, True, False):
        for m in dout:
            m.setval(-7.0)
        P = capi.curv_params(threshold=0.05, fused=fused, do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True, vel_comp=1)
        capi.curvature_run(ctx, dst, 0, bc, P, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        for l in range(H.nlev):
            assert_valid_bits_equal(dout[l].download(), oout[l], [(c, c) for c in range(17)], f"{name} options fused {fused} level {l}")
    # the work multifabs kept with the levels can be released (and come back on the next call)
    freed = sum(int(ctx.lib.pa_level_free_scratch(dl.h)) for dl in dls)
    assert freed > 0 and sum(int(ctx.lib.pa_level_free_scratch(dl.h)) for dl in dls) == 0
    for m in dout:
        m.setval(-7.0)
    capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(threshold=0.05, fused=True, do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True, vel_comp=1), dout, 0)
    ctx.sync()
    for l in range(H.nlev):
        assert_valid_bits_equal(dout[l].download(), oout[l], [(c, c) for c in range(17)], f"{name} options after pa_level_free_scratch, level {l}")
    # too few output components are rejected before any launch
    small = [capi.DevMF(ctx, dl, 6, 0) for dl in dls]
    with pytest.raises(capi.PaError):
        capi.curvature_run(ctx, dst, 0, bc, P, small, 0)


@pytest.mark.parametrize("nlev,base,box", [(5, 8, 8), (6, 8, 4), (5, 40, 40), (5, 64, 64)])
def test_deep_hierarchies_take_the_fused_pipelines(ctx, oracle, nlev, base, box):
    """five and six levels (what a Pele plotfile often holds): the all-levels launches come in chunks of four groups -- until round 5 a
    hierarchy of more than four levels went group by group, and its curvature options pass by pass.  The fused grad -> curvature pass
    and pa_curvature_run with every option (fast path asserted) against the oracle, bit for bit; narrow and wide boxes."""
    from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
    per = (1, 1, 0)
    H = nested_hierarchy(base, nlev, box, is_per=per)
    assert H.nlev == nlev
    states = make_states(H, 4, 2, field_flame, seed=3)
    bc = capi.bc_from_flags(per)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=False)
    oo = [MultiFab(lv, 17, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oo, 0, MultiFab, threshold=0.03, do_gauss=True, vel_comp=1, do_strain=True,
                              do_velnormal=True, strain_tensor=True)
    ou = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, ou, 0, MultiFab)
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    d8 = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(fused=True), work, d8, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
    assert "_levels<" in kn, kn
    for l in range(nlev):
        got = d8[l].download()
        assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"{nlev} levels: grad level {l}")
        assert_valid_bits_equal(got, ou[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"{nlev} levels: curvature level {l}")
    d4 = [capi.DevMF(ctx, dl, 4, 0) for dl in dls]  # the gradient tool's pass (64-row boxes: the all-levels gradient launch, in chunks too)
    capi.grad_run(ctx, dst, 0, bc, d4, 0)
    ctx.sync()
    for l in range(nlev):
        assert_valid_bits_equal(d4[l].download(), og[l], [(c, c) for c in range(4)], f"{nlev} levels: grad_run level {l}")
    d17 = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
    capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(threshold=0.03, fused=True, do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True, vel_comp=1), d17, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0 and ctx.lib.pa_curvature_last_path(ctx.h) in (1, 2)
    for l in range(nlev):
        assert_valid_bits_equal(d17[l].download(), oo[l], [(c, c) for c in range(17)], f"{nlev} levels: options level {l}")


def test_gradcurv_run_comps_equals_component_by_component(ctx, oracle):
    """pa_gradcurv_run_comps (FillBoundary of all components hoisted into one launch, results delivered through the callback
    before the next component overwrites them) == pa_gradcurv_run per component, bit for bit; wide boxes (exact-normal
    pipeline) and narrow boxes (falls back to the per-component path)"""
    from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
    for base, box in ((80, 40), (16, 8)):
        H = nested_hierarchy(base, 3, box, is_per=(1, 1, 0))
        states = make_states(H, 4, 2, field_flame, seed=31)
        bc = capi.bc_from_flags((1, 1, 0))
        dls, dst = _dev(ctx, H, states)
        work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
        dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
        want = {}
        for c in (1, 2, 3):
            capi.gradcurv_run(ctx, dst, c, bc, capi.curv_params(fused=True), work, dout, 0)
            ctx.sync()
            want[c] = [d.download().data.copy() for d in dout]
        dst2 = [capi.DevMF.from_host(ctx, dl, st) for dl, st in zip(dls, states)]  # fresh ghost cells, same levels
        got = {}

        def done(c):
            ctx.sync()
            got[c] = [d.download().data.copy() for d in dout]
        capi.gradcurv_run_comps(ctx, dst2, 1, 3, bc, capi.curv_params(fused=True), work, dout, 0, done)
        ctx.sync()
        assert ctx.bc_errors() == 0 and sorted(got) == [1, 2, 3]
        for c in (1, 2, 3):
            for l in range(H.nlev):
                assert np.array_equal(got[c][l].view(np.int64), want[c][l].view(np.int64)), (base, c, l)


@pytest.mark.parametrize("threshold", [None, 0.1])
@pytest.mark.parametrize("nbatch", [2, 5, 16])
def test_gradcurv_run_comps_batched_equals_component_by_component(ctx, oracle, nbatch, threshold):
    """pa_gradcurv_run_comps2: the boundary kernels of the exact-normal pipeline once per BATCH of components (slot = grid
    dimension: own compact ghost arrays, coarse patches, progress range and 8 output components per slot), the sweeps
    component by component -- every component bit-equal to pa_gradcurv_run on it alone.  5 components with different
    progress ranges (the per-slot range matters), batches that do not divide the count, with and without the threshold clip
    (its hand-over list carries the slot)."""
    from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
    H = nested_hierarchy(80, 3, 40, is_per=(1, 1, 0))
    ncomp = 5
    states = make_states(H, ncomp, 2, field_flame, seed=43)
    for st in states:  # different ranges per component
        for b in range(st.level.nboxes):
            for c in range(ncomp):
                st.fab(b)[c] = st.fab(b)[c] * (1.0 + 0.37 * c) + 11.0 * c
    bc = capi.bc_from_flags((1, 1, 0))
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    params = capi.curv_params(threshold=threshold, fused=True)
    want = {}
    for c in range(ncomp):
        capi.gradcurv_run(ctx, dst, c, bc, params, work, dout, 0)
        ctx.sync()
        want[c] = [d.download() for d in dout]
    dst2 = [capi.DevMF.from_host(ctx, dl, st) for dl, st in zip(dls, states)]  # fresh ghost cells, same levels
    nslot = min(nbatch, ncomp)
    dout2 = [capi.DevMF(ctx, dl, 3 + 8 * nslot, 0) for dl in dls]
    got = {}

    def done(c, oc):
        ctx.sync()
        got[c] = (oc, [d.download() for d in dout2])
    capi.gradcurv_run_comps2(ctx, dst2, 0, ncomp, bc, params, work, dout2, 3, nbatch, done)
    ctx.sync()
    assert ctx.bc_errors() == 0 and sorted(got) == list(range(ncomp))
    kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
    assert kn.endswith("CG=1>") or "_levels<" in kn, kn
    for c in range(ncomp):
        oc, mfs = got[c]
        assert oc == 3 + 8 * (c % nslot)
        for l in range(H.nlev):
            for b in range(H.levels[l].nboxes):
                assert np.array_equal(mfs[l].fab(b)[oc:oc + 8].view(np.int64), want[c][l].fab(b)[0:8].view(np.int64)), (nbatch, c, l, b)


def test_gradcurv_run_comps2_arguments(ctx):
    """pa_gradcurv_run_comps2: `out` must hold 8 components per slot of the batch (a batch larger than the component count is
    cut to it, 0 is taken as 1); a bad component range or a short `out` is refused with a message, nothing is launched"""
    from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
    H = nested_hierarchy(80, 2, 40, is_per=(1, 1, 0))
    states = make_states(H, 3, 2, field_flame, seed=47)
    bc = capi.bc_from_flags((1, 1, 0))
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    params = capi.curv_params(fused=True)
    out16 = [capi.DevMF(ctx, dl, 16, 0) for dl in dls]
    seen = []
    capi.gradcurv_run_comps2(ctx, dst, 0, 2, bc, params, work, out16, 0, 99, lambda c, oc: seen.append((c, oc)))  # 99 -> 2 slots
    capi.gradcurv_run_comps2(ctx, dst, 2, 1, bc, params, work, out16, 8, 0, lambda c, oc: seen.append((c, oc)))   # 0 -> 1 slot, at ocomp 8
    ctx.sync()
    assert seen == [(0, 0), (1, 8), (2, 8)]
    for bad in (lambda: capi.gradcurv_run_comps2(ctx, dst, 0, 3, bc, params, work, out16, 0, 3),    # 3 slots need 24 components
                lambda: capi.gradcurv_run_comps2(ctx, dst, 0, 3, bc, params, work, out16, 9, 1),    # 9 + 8 > 16
                lambda: capi.gradcurv_run_comps2(ctx, dst, 2, 2, bc, params, work, out16, 0, 1),    # components 2, 3 of 3
                lambda: capi.gradcurv_run_comps2(ctx, dst, 0, 0, bc, params, work, out16, 0, 1)):   # no component
        with pytest.raises(capi.PaError):
            bad()


def test_switched_off_paths_still_match(ctx):
    """PA_FORCE_FALLBACKS=1: the paths that exist for inputs the tuned ones do not take -- FillBoundary and the patch gather per
    ghost cell (regions that do not fit a plan), the sweeps group by group (more groups than one launch holds), ghost fills level
    by level -- on the ordinary test hierarchies, through the same oracle comparisons, in a child process (plans are cached per level)"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PA_FORCE_FALLBACKS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "ghost_fill_matches_oracle or exact_normal_pipeline or fused_matches_oracle"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_not_properly_nested_fine_level_is_counted_on_every_path(ctx):
    """a level-2 region that touches the edge of level 1 (no buffer cells): its ghost cells beyond that edge have no coarse
    parent.  The library never aborts -- it counts those ghost cells (pa_bc_errors; the tools turn a non-zero count into the
    reference's abort): in the coarse patches a cell without an owner is a reserved bit pattern, and the chunked face kernels
    count per ghost cell like the per-cell ones (26088 here: three faces of 64 x 64 ghost cells, counted by the preparation of
    phi's ghosts and by the fix-up of n's, plus the ring)"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level
    z, o = np.zeros(3), np.ones(3)
    l0 = Level(np.array([[0, 0, 0, 63, 63, 63]], np.int32), (0, 0, 0), (63, 63, 63), (0, 0, 0), z, o)
    l1 = Level(np.array([[16, 16, 16, 79, 79, 79]], np.int32), (0, 0, 0), (127, 127, 127), (0, 0, 0), z, o)
    l2 = Level(np.array([[32, 32, 32, 95, 95, 95]], np.int32), (0, 0, 0), (255, 255, 255), (0, 0, 0), z, o)  # level-1 cells 16..47: flush with level 1's low sides
    H = Hierarchy([l0, l1, l2], 2)
    rng = np.random.default_rng(5)
    states = []
    for lv in H.levels:
        m = MultiFab(lv, 1, 2)
        m.data[:] = 300.0 + 1700.0 * rng.random(m.total)
        states.append(m)
    bc = capi.bc_from_flags((0, 0, 0))
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    counts = {}
    for sw in ("1", "2"):
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=200.0, prog_max=2100.0, fused=True), work, dout, 0)
        ctx.sync()
        kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
        assert kn.endswith("CG=1>") or "_levels<" in kn, kn
        counts[sw] = ctx.bc_errors()
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=200.0, prog_max=2100.0, fused=False), work, dout, 0)
    ctx.sync()
    assert ctx.bc_errors() > 0  # pass by pass: counted as well (its own number of passes over those cells)
    # three faces of 64 x 64 ghost cells sit beyond level 1, counted once by the prep and once by the fix-up of phi's and n's ghosts
    assert counts["1"] == counts["2"] == 26088, counts


def test_work_multifabs_are_never_read_before_they_are_written():
    """PA_SCRATCH_POISON=1 (round-5 advisor): the work multifabs a level keeps between calls (pa_level_scratch: G, c, n of the
    curvature passes, the Krylov vectors of do_smooth, the smoothed progress source) are filled with NaN every time a call acquires
    them; the curvature-option, smoothing and random-hierarchy tests -- oracle comparisons, bit for bit or to the solve's tolerance --
    run under it in a child process.  A kernel that reads a ghost cell no step of the same call wrote would put a NaN in a result."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PA_SCRATCH_POISON="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_gradcurv.py"), os.path.join(here, "test_gpu_smooth.py"), os.path.join(here, "test_gpu_random.py"),
                        "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", "-k", "curvature_run or options or curvature_options or smoothing or gauss or strain"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(here))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("per,sym", [((0, 0, 0), (0, 0, 0)), ((0, 1, 0), (1, 0, 1)), ((1, 0, 1), (0, 0, 0))])
def test_chunked_face_kernels_odd_origins_small_faces_and_slots(ctx, oracle, per, sym):
    """The chunk records' corner cases (pa_fused.hip, round 6): level-0 boxes cut on ODD indices (block origins on even global
    indices start at -1: cells outside the face are predicated off), extents that are not multiples of anything, faces 3 cells
    wide, wall / reflect-odd / coarse-fine cells in one chunk, a fine level whose faces end in one-sided stencils next to the
    walls -- bit for bit against the oracle, one component at a time and three components in one batch (slot = blockIdx.z)."""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, field_flame

    def cut(lo, hi, cuts):
        edges = [lo[d:d + 1] + [c for c in cuts[d]] + [hi[d] + 1] for d in range(3)]
        out = []
        for kz in range(len(edges[2]) - 1):
            for ky in range(len(edges[1]) - 1):
                for kx in range(len(edges[0]) - 1):
                    out.append([edges[0][kx], edges[1][ky], edges[2][kz], edges[0][kx + 1] - 1, edges[1][ky + 1] - 1, edges[2][kz + 1] - 1])
        return np.array(out, np.int32)
    l0 = Level(cut([0, 0, 0], [45, 37, 29], ([13, 27], [19], [3, 15])), (0, 0, 0), (45, 37, 29), per, np.zeros(3), np.ones(3))  # a 3-plane slab in z, odd cuts in x and y
    # level 1: coarse cells [2, 19] x [8, 17] x [0, 13] -- flush with the z-low wall, two coarse cells from the x-low wall
    l1 = Level(cut([4, 16, 0], [39, 35, 27], ([22], [], [10])), (0, 0, 0), (91, 75, 59), per, np.zeros(3), np.ones(3))
    H = Hierarchy([l0, l1], 2)
    ncomp = 3
    states = make_states(H, ncomp, 2, field_flame, seed=91)
    for st in states:
        for b in range(st.level.nboxes):
            for c in range(ncomp):
                st.fab(b)[c] = st.fab(b)[c] * (1.0 + 0.21 * c) + 7.0 * c
    bc = capi.bc_from_flags(per, sym)
    dls, dst = _dev(ctx, H, states)
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    want = {}
    for c in range(ncomp):
        og = [MultiFab(lv, 4, 0) for lv in H.levels]
        oc = [MultiFab(lv, 5, 0) for lv in H.levels]
        oracle.grad_pipeline(H.levels, [s.copy() for s in states], c, bc, og, 0, multipass=True)
        oracle.curvature_pipeline(H.levels, [s.copy() for s in states], c, bc, oc, 0, MultiFab)
        want[c] = (og, oc)
        dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
        capi.gradcurv_run(ctx, dst, c, bc, capi.curv_params(fused=True), work, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        assert ctx.lib.pa_sweep_kernel_name(ctx.h).decode().find("march3") >= 0
        for l in range(H.nlev):
            got = dout[l].download()
            assert_valid_bits_equal(got, og[l], [(k, k) for k in range(4)], f"comp {c} grad level {l}")
            assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"comp {c} curv level {l}")
    # the same three components as one batch
    dst2 = [capi.DevMF.from_host(ctx, dl, st) for dl, st in zip(dls, states)]
    dout2 = [capi.DevMF(ctx, dl, 8 * ncomp, 0) for dl in dls]
    capi.gradcurv_run_comps2(ctx, dst2, 0, ncomp, bc, capi.curv_params(fused=True), work, dout2, 0, ncomp, None)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for c in range(ncomp):
        og, oc = want[c]
        for l in range(H.nlev):
            got = dout2[l].download()
            assert_valid_bits_equal(got, og[l], [(8 * c + k, k) for k in range(4)], f"batch comp {c} grad level {l}")
            assert_valid_bits_equal(got, oc[l], [(8 * c + 4, 2), (8 * c + 5, 3), (8 * c + 6, 4), (8 * c + 7, 1)], f"batch comp {c} curv level {l}")
    # the gradient tool's applyBC goes through the same kernel (PHIONLY) on a state with ONE ghost layer
    st1 = make_states(H, 1, 1, field_flame, seed=91)
    dst1 = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, st1)]
    og1 = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in st1], 0, bc, og1, 0, multipass=True)
    dgr = [capi.DevMF(ctx, dl, 4, 0) for dl in dls]
    capi.grad_run(ctx, dst1, 0, bc, dgr, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for l in range(H.nlev):
        assert_valid_bits_equal(dgr[l].download(), og1[l], [(k, k) for k in range(4)], f"grad_run level {l}")


@pytest.mark.parametrize("threshold", [None, 0.06])
@pytest.mark.parametrize("per,sym,base,box", [((1, 1, 0), (0, 0, 0), 96, 48), ((0, 0, 0), (1, 0, 1), 80, 40), ((0, 1, 1), (0, 0, 0), 104, 52)])
def test_gaussian_curvature_inside_the_sweep(ctx, oracle, per, sym, base, box, threshold):
    """pa_curvature_run with do_gaussCurv on hierarchies whose boxes are all wider than 32 cells: the G-output sweeps form the
    Gaussian curvature themselves (pa_fused_march3.h GOUT == 2: G's x / y neighbours through LDS rings, z-neighbours in registers,
    k_gauss_curv's operations), k_gauss_cells recomputes the first layer behind the special faces from the stored G -- path 2, all 17
    components bit for bit against the oracle (3 levels: coarse-fine faces in every direction, walls / reflect-odd / periodic,
    tiles of 13 rows (52-row boxes) and of 8 (40 / 48 rows), with and without the threshold clip), and against the pass-by-pass path"""
    from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
    H = nested_hierarchy(base, 3, box, is_per=per)
    states = make_states(H, 4, 2, field_flame, seed=29)
    bc = capi.bc_from_flags(per, sym)
    oo = [MultiFab(lv, 17, 0) for lv in H.levels]
    opts = dict(do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True)
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oo, 0, MultiFab, threshold=threshold, vel_comp=1, **opts)
    dls, dst = _dev(ctx, H, states)
    for fused, want_path in ((True, 2), (False, 0)):
        d17 = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
        for m in d17:
            m.setval(-5.0)
        capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(threshold=threshold, fused=fused, vel_comp=1, **opts), d17, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0 and ctx.lib.pa_curvature_last_path(ctx.h) == want_path
        for l in range(H.nlev):
            assert_valid_bits_equal(d17[l].download(), oo[l], [(c, c) for c in range(17)], f"fused={fused} level {l}")
    # the Gaussian curvature alone (no strain / velocity pass at all)
    og = [MultiFab(lv, 17, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, MultiFab, threshold=threshold, vel_comp=1, do_gauss=True)
    d17 = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
    capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(threshold=threshold, fused=True, vel_comp=1, do_gauss=True), d17, 0)
    ctx.sync()
    assert ctx.lib.pa_curvature_last_path(ctx.h) == 2
    for l in range(H.nlev):
        assert_valid_bits_equal(d17[l].download(), og[l], [(c, c) for c in range(6)], f"gauss only, level {l}")


@pytest.mark.parametrize("per,layout", [((0, 0, 0), 0), ((1, 0, 0), 0), ((0, 0, 0), 1)])
def test_gaussian_curvature_inside_the_sweep_staggered_boxes(ctx, oracle, per, layout):
    """The GOUT == 2 sweeps store G only where something reads it afterwards (pa_sweep_gneed, pa_core.hip): three layers behind box
    faces -- behind an x face only when the face is special or a neighbouring box across it has fix-up cells next to it -- and the coarse
    tiles a patch of the finer level gathers from.  Fine boxes STAGGERED in y and z across their x faces (a face partly covered, partly
    coarse-fine; a neighbour's special y / z plane meeting the face in its interior; the same through the periodic wrap), work multifabs
    poisoned with NaN by the caller's child process (test_work_multifabs_are_never_read_before_they_are_written): all 17 components
    bit for bit against the oracle, path 2."""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, field_flame
    l0 = Level(np.array([[0, 0, 0, 95, 63, 31]], np.int32), (0, 0, 0), (95, 63, 31), per, np.zeros(3), np.ones(3))
    if layout == 1:
        # the x-low face of the first box is covered entirely (two boxes stacked in y), so it is not special, nor is the face of the lower
        # box that looks at it; but the lower box's y-high face is (the upper box is narrower), and its first-layer cell next to the first
        # box differentiates G across that x face, in the middle of the first box's rows
        fine = [[80, 32, 16, 143, 95, 55], [16, 32, 16, 79, 63, 55], [40, 64, 16, 79, 95, 55]]
    elif per[0]:
        fine = [[0, 16, 8, 63, 63, 39], [128, 32, 16, 191, 95, 55], [64, 16, 8, 127, 63, 39]]
    else:
        fine = [[16, 16, 8, 79, 63, 39], [80, 32, 16, 143, 95, 55], [80, 96, 16, 143, 127, 55]]
    l1 = Level(np.array(fine, np.int32), (0, 0, 0), (191, 127, 63), per, np.zeros(3), np.ones(3))
    H = Hierarchy([l0, l1], 2)
    states = make_states(H, 4, 2, field_flame, seed=57)
    bc = capi.bc_from_flags(per, (0, 0, 0))
    opts = dict(do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True)
    for threshold in (None, 0.05):
        oo = [MultiFab(lv, 17, 0) for lv in H.levels]
        oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oo, 0, MultiFab, threshold=threshold, vel_comp=1, **opts)
        dls, dst = _dev(ctx, H, states)
        d17 = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
        capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(threshold=threshold, fused=True, vel_comp=1, **opts), d17, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0 and ctx.lib.pa_curvature_last_path(ctx.h) == 2
        for l in range(H.nlev):
            assert_valid_bits_equal(d17[l].download(), oo[l], [(c, c) for c in range(17)], f"threshold={threshold} level {l}")


@pytest.mark.parametrize("per,sym", [((1, 1, 1), (0, 0, 0)), ((1, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 0, 1))])
def test_gaussian_curvature_inside_the_sweep_single_level(ctx, oracle, per, sym):
    """GOUT == 2 on ONE level (no finer level: no coarse tile stores G in full, no coarser one: walls and periodic sides only) cut into
    boxes of different shapes -- all 17 components bit for bit against the oracle, path 2"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, field_flame
    boxes = [[0, 0, 0, 47, 39, 23], [48, 0, 0, 95, 39, 23], [0, 40, 0, 95, 63, 23], [0, 0, 24, 39, 63, 47], [40, 0, 24, 95, 63, 47]]
    l0 = Level(np.array(boxes, np.int32), (0, 0, 0), (95, 63, 47), per, np.zeros(3), np.ones(3))
    H = Hierarchy([l0], 2)
    states = make_states(H, 4, 2, field_flame, seed=3)
    bc = capi.bc_from_flags(per, sym)
    opts = dict(do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True)
    oo = [MultiFab(lv, 17, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oo, 0, MultiFab, threshold=0.03, vel_comp=1, **opts)
    dls, dst = _dev(ctx, H, states)
    d17 = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
    capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(threshold=0.03, fused=True, vel_comp=1, **opts), d17, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0 and ctx.lib.pa_curvature_last_path(ctx.h) == 2
    assert_valid_bits_equal(d17[0].download(), oo[0], [(c, c) for c in range(17)], "single level")


def test_gaussian_curvature_inside_the_sweep_five_levels(ctx, oracle):
    """five levels of wide boxes: the G-output sweeps go in chunks of four sweep groups per launch, every level but the finest stores G
    in full in the tiles under the next level's coarse patches -- path 2, bit for bit against the oracle"""
    from peleanalysis_amd.hierarchy import nested_hierarchy, field_flame
    H = nested_hierarchy(48, 5, 48, is_per=(0, 1, 0))
    assert H.nlev == 5
    states = make_states(H, 4, 2, field_flame, seed=77)
    bc = capi.bc_from_flags((0, 1, 0), (1, 0, 0))
    opts = dict(do_gauss=True, do_strain=True, do_velnormal=True)
    oo = [MultiFab(lv, 17, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oo, 0, MultiFab, threshold=None, vel_comp=1, **opts)
    dls, dst = _dev(ctx, H, states)
    d17 = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
    capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(threshold=None, fused=True, vel_comp=1, **opts), d17, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0 and ctx.lib.pa_curvature_last_path(ctx.h) == 2
    for l in range(H.nlev):
        assert_valid_bits_equal(d17[l].download(), oo[l], [(c, c) for c in range(8)], f"level {l}")
