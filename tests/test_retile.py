"""Internal re-tiling (pa_level_retile, include/peleanalysis_amd.h).

CPU tier: (1) the re-tiler returns the same cell set in disjoint boxes, leaves thin boxes alone and respects its limits;
(2) THE ORACLE IS BITWISE INVARIANT UNDER RE-TILING -- grad.cpp:158-236, curvature.cpp:283-570 (with and without the
threshold clip) and the filterPlt ghost fill + Filter::apply_filter (filterPlt.cpp:159-219) give the same bits in every
cell whether the level is held in the file's boxes, in the re-tiler's, or in a random re-chop (pieces >= 3 cells thick).
That is the licence for sweeping another tiling than the file's, and a consistency pin on the recalled applyBC /
InterpBndryData / FillPatchTwoLevels restatement: those must test coverage by the LEVEL, never by a box.

GPU tier: the HIP path on the re-tiled level against the oracle on the ORIGINAL BoxArray, compared per original box."""
import numpy as np
import pytest

from peleanalysis_amd.hierarchy import Hierarchy, Level, MultiFab, regrid_copy, retile_level, union_hierarchy, _occupancy
from test_gpu_random import _draw
from util import bits_equal, make_states
from peleanalysis_amd.hierarchy import field_flame, field_trig


def _random_rechop(level: Level, rng, min_thick=3) -> Level:
    """every box cut at random planes into pieces of at least min_thick cells (boxes thinner than 2 * min_thick stay whole
    in that direction): a tiling FINER than the file's"""
    out = []
    for b in level.boxes:
        cuts = []
        for d in range(3):
            lo, hi = int(b[d]), int(b[3 + d])
            edges = [lo]
            while hi + 1 - edges[-1] >= 2 * min_thick and rng.random() < 0.6:
                edges.append(int(rng.integers(edges[-1] + min_thick, hi + 2 - min_thick)))
            edges.append(hi + 1)
            cuts.append(edges)
        for k in range(len(cuts[2]) - 1):
            for j in range(len(cuts[1]) - 1):
                for i in range(len(cuts[0]) - 1):
                    out.append([cuts[0][i], cuts[1][j], cuts[2][k], cuts[0][i + 1] - 1, cuts[1][j + 1] - 1, cuts[2][k + 1] - 1])
    return Level(np.asarray(out, dtype=np.int32), level.domlo, level.domhi, level.is_per, level.prob_lo, level.prob_hi)


def _thin(level, min_thick=3):
    n = level.boxes[:, 3:] - level.boxes[:, :3] + 1
    return (n < min_thick).any(axis=1)


def _draw_h(seed):
    if seed % 2:
        H = union_hierarchy(4000 + seed)
        rng = np.random.default_rng(seed)
        per = tuple(int(x) for x in H.levels[0].is_per)
        sym = tuple(int(x) for x in np.where(np.asarray(per) == 1, 0, rng.integers(0, 2, size=3)))
        return H, per, sym, (field_flame if seed % 4 == 1 else field_trig)
    return _draw(300 + seed)


def _tilings(H, seed):
    """the file's tiling, the re-tiler's (small limits so that the chains are cut), a random re-chop, the re-tiler on the re-chop"""
    rng = np.random.default_rng(50 + seed)
    mx = tuple(int(v) for v in rng.choice([8, 12, 16, 24, 1000], size=3))
    a = Hierarchy([retile_level(lv, mx) for lv in H.levels], 2)
    # a re-chop must not cut boxes that are thin (their interpolation order depends on their thickness): only thick boxes
    b_levels = []
    for lv in H.levels:
        t = _thin(lv)
        thick = Level(lv.boxes[~t], lv.domlo, lv.domhi, lv.is_per, lv.prob_lo, lv.prob_hi)
        rc = _random_rechop(thick, rng).boxes if thick.nboxes else np.zeros((0, 6), np.int32)
        b_levels.append(Level(np.concatenate([rc, lv.boxes[t]]), lv.domlo, lv.domhi, lv.is_per, lv.prob_lo, lv.prob_hi))
    b = Hierarchy(b_levels, 2)
    c = Hierarchy([retile_level(lv, (1000, 1000, 1000)) for lv in b.levels], 2)
    return {"retiled": a, "rechopped": b, "rechopped+retiled": c}, mx


@pytest.mark.parametrize("seed", range(12))
def test_retile_keeps_the_cell_set(seed):
    H, _, _, _ = _draw_h(seed)
    til, mx = _tilings(H, seed)
    for name, T in til.items():
        for l, (lv, tv) in enumerate(zip(H.levels, T.levels)):
            assert tv.ncells == lv.ncells, (name, l)
            occ = np.zeros_like(_occupancy(lv), dtype=np.int32)
            for lo0, lo1, lo2, hi0, hi1, hi2 in tv.boxes - np.concatenate([tv.domlo, tv.domlo]):
                occ[lo2:hi2 + 1, lo1:hi1 + 1, lo0:hi0 + 1] += 1
            assert occ.max() == 1 and np.array_equal(occ.astype(bool), _occupancy(lv)), (name, l, "boxes overlap or the cell set changed")
            # thin boxes of the file come back unchanged, and no new thin box appears
            thin_in = {tuple(b) for b in lv.boxes[_thin(lv)]}
            thin_out = {tuple(b) for b in tv.boxes[_thin(tv)]}
            assert thin_out == thin_in, (name, l)
            if name == "retiled":
                n = tv.boxes[:, 3:] - tv.boxes[:, :3] + 1
                n_in = (lv.boxes[:, 3:] - lv.boxes[:, :3] + 1).max(axis=0)
                assert (n <= np.maximum(np.asarray(mx), n_in)).all(), (name, l, "a merged box exceeds max_size")


def test_retile_regular_tilings():
    """what the bench's secondary entries rely on: a 512^3 level in 32^3 boxes comes back as 64 boxes of 128^3, the three
    levels of the headline hierarchy as 128^3 boxes whatever the file's chop"""
    from peleanalysis_amd.hierarchy import nested_hierarchy
    want = nested_hierarchy(512, 3, 128)
    for box in (32, 64):
        H = nested_hierarchy(512, 3, box)
        for lv, wv in zip(H.levels, want.levels):
            tv = retile_level(lv, (128, 128, 128))
            assert {tuple(b) for b in tv.boxes} == {tuple(b) for b in wv.boxes}


def _dense(mfs, comps):
    """[level] -> float64[ncomp, nz, ny, nx] over the domain (NaN outside the level's cells)"""
    out = []
    for mf in mfs:
        lv = mf.level
        n = lv.domhi - lv.domlo + 1
        d = np.full((len(comps), int(n[2]), int(n[1]), int(n[0])), np.nan)
        for b in range(lv.nboxes):
            lo = lv.boxes[b, :3] - lv.domlo
            hi = lv.boxes[b, 3:] - lv.domlo
            d[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = mf.valid(b)[comps]
        out.append(d)
    return out


def _on_tiling(states, T, ng):
    """the same cell values on another tiling (ghost cells poisoned)"""
    out = []
    for s, tv in zip(states, T.levels):
        m = MultiFab(tv, s.ncomp, ng, fill=np.nan)
        regrid_copy(s, m)
        out.append(m)
    return out


def _oracle_all(oracle, H, states, per, sym, thr):
    bc = oracle.bc_from_flags(per, sym)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=True)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=thr)
    return _dense(og, [0, 1, 2, 3]), _dense(oc, [0, 1, 2, 3, 4])


@pytest.mark.parametrize("seed", range(12))
def test_oracle_is_invariant_under_retiling(oracle, seed):
    H, per, sym, fn = _draw_h(seed)
    thr = None if seed % 3 else 0.04
    states = make_states(H, 1, 2, fn, seed=seed)
    g0, c0 = _oracle_all(oracle, H, states, per, sym, thr)
    til, mx = _tilings(H, seed)
    for name, T in til.items():
        g1, c1 = _oracle_all(oracle, T, _on_tiling(states, T, 2), per, sym, thr)
        for l in range(H.nlev):
            assert bits_equal(g0[l], g1[l]), f"seed {seed} {name} (max {mx}): gradient of level {l} depends on the tiling"
            assert bits_equal(c0[l], c1[l]), f"seed {seed} {name} (max {mx}): curvature / normals of level {l} depend on the tiling"


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("interp", [0, 1])
def test_oracle_filter_is_invariant_under_retiling(oracle, seed, interp):
    H, per, sym, fn = _draw_h(seed)
    fields = make_states(H, 2, 0, fn, seed=seed + 3)
    ngs = [1, 2, 4]

    def run(T):
        ins = []
        for l, (f, tv) in enumerate(zip(fields, T.levels)):
            m = MultiFab(tv, 2, ngs[l], fill=np.nan)
            regrid_copy(f, m)
            ins.append(m)
        outs = [MultiFab(tv, 2, 0) for tv in T.levels]
        oracle.filter_pipeline(T.levels, ins, outs, 2, base_fgr=2, interp_type=interp)
        return _dense(outs, [0, 1])

    f0 = run(H)
    til, mx = _tilings(H, seed)
    for name, T in til.items():
        # the filter's own ghost width must fit the boxes FillPatch reads: keep the tilings whose boxes are not thin
        f1 = run(T)
        for l in range(H.nlev):
            assert bits_equal(f0[l], f1[l]), f"seed {seed} {name} (max {mx}) interp {interp}: filtered level {l} depends on the tiling"


# ------------------------------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_retiled_device_path_matches_oracle_on_the_file_boxes(ctx, oracle, seed):
    """what the tools do: the file's FABs into the re-tiled level, the HIP pipelines there, results back per ORIGINAL box,
    compared bit for bit with the oracle run on the ORIGINAL BoxArray"""
    from peleanalysis_amd import capi
    H, per, sym, fn = _draw_h(seed)
    thr = None if seed % 3 else 0.04
    states = make_states(H, 1, 2, fn, seed=seed)
    bc = capi.bc_from_flags(per, sym)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=True)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=thr)
    rng = np.random.default_rng(seed)
    mx = tuple(int(v) for v in rng.choice([12, 16, 24, 1000], size=3))
    T = Hierarchy([retile_level(lv, mx) for lv in H.levels], 2)
    tst = _on_tiling(states, T, 2)
    for fused in (True, False):
        dls = [capi.DevLevel(ctx, lv) for lv in T.levels]
        dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, tst)]
        work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
        dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=thr, fused=fused), work, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        for l, lv in enumerate(H.levels):
            back = MultiFab(lv, 8, 0, fill=np.nan)
            regrid_copy(dout[l].download(), back)
            for b in range(lv.nboxes):
                for gc, (ref, rc) in {0: (og, 0), 1: (og, 1), 2: (og, 2), 3: (og, 3), 4: (oc, 2), 5: (oc, 3), 6: (oc, 4), 7: (oc, 1)}.items():
                    assert bits_equal(back.valid(b)[gc], ref[l].valid(b)[rc]), \
                        f"seed {seed} max {mx} fused {fused}: level {l} file box {b} comp {gc} differs from the oracle on the file's BoxArray " \
                        f"({lv.nboxes} file boxes -> {T.levels[l].nboxes})"


@pytest.mark.parametrize("mx", [(5, 7, 3), (13, 8, 8), (24, 12, 16), (40, 16, 16)])
def test_small_limits_keep_fine_boxes_aligned_to_the_refinement_ratio(mx):
    """round-5 advisor finding: with an x limit below one 64-cell tile the even split in cells gave boxes like [40..52] out of
    inputs aligned to 4 -- the smoothing solve refuses fine boxes that are not aligned to the refinement ratio, so
    curvature3d do_smooth=1 failed on a re-tiled level that worked with retile=0.  Cuts now fall on even indices."""
    from peleanalysis_amd.hierarchy import retile_level
    rng = np.random.default_rng(mx[0])
    cut = 0
    for trial in range(20):
        # a fine level: a few boxes on a grid of 4 cells (every lo even, every hi odd)
        n = int(rng.integers(2, 5))
        boxes = []
        x0 = 4 * int(rng.integers(0, 8))
        for b in range(n):
            w = 4 * int(rng.integers(2, 30))
            boxes.append([x0, 8 * b, 0, x0 + w - 1, 8 * b + 7, 7])
        lv = Level(np.array(boxes, np.int32), (0, 0, 0), (1023, 255, 63), (0, 0, 0), np.zeros(3), np.ones(3))
        out = retile_level(lv, mx, min_thick=1)
        assert int(((out.boxes[:, 3:] - out.boxes[:, :3] + 1).prod(axis=1)).sum()) == lv.ncells
        assert (out.boxes[:, 0] % 2 == 0).all() and (out.boxes[:, 3] % 2 == 1).all(), (mx, boxes, out.boxes.tolist())
        cut += int(out.nboxes > lv.nboxes)  # (more pieces than the caller's capacity: the level is merged by whole boxes instead)
    assert cut > 0 or mx[0] < 13, "no trial was cut in x"


def _thin_slab_hierarchy():
    """level 1 = thick boxes plus slabs only 2 cells thick in x at the coarse-fine face: their normal interpolant is of
    lower order (orc_apply_bc: NX = min(n + 1, 4)), so they must not be merged"""
    per = (0, 1, 1)
    l0 = Level(np.array([[0, 0, 0, 15, 15, 15]]), (0, 0, 0), (15, 15, 15), per, np.zeros(3), np.ones(3))
    boxes = [[8, 8, 8, 9, 23, 23], [10, 8, 8, 15, 15, 23], [10, 16, 8, 15, 23, 23], [16, 8, 8, 21, 23, 23], [22, 8, 8, 23, 23, 23]]
    l1 = Level(np.array(boxes), (0, 0, 0), (31, 31, 31), per, np.zeros(3), np.ones(3))
    return Hierarchy([l0, l1], 2), per, (0, 0, 0)


def test_thin_boxes_pass_through_and_are_the_reason_for_min_thick(oracle):
    H, per, sym = _thin_slab_hierarchy()
    states = make_states(H, 1, 2, field_flame, seed=5)
    g0, c0 = _oracle_all(oracle, H, states, per, sym, None)
    T = Hierarchy([retile_level(lv, (1000, 1000, 1000), min_thick=3) for lv in H.levels], 2)
    assert T.levels[1].nboxes == 3  # the three thick boxes merged into one, the two slabs untouched
    assert {tuple(b) for b in T.levels[1].boxes[-2:]} == {(8, 8, 8, 9, 23, 23), (22, 8, 8, 23, 23, 23)}
    g1, c1 = _oracle_all(oracle, T, _on_tiling(states, T, 2), per, sym, None)
    assert bits_equal(g0[1], g1[1]) and bits_equal(c0[1], c1[1])
    # negative control: merging the slabs as well (min_thick = 1) changes the bits next to the coarse-fine face -- the test
    # above can tell tilings apart, and the thickness rule is needed
    W = Hierarchy([retile_level(lv, (1000, 1000, 1000), min_thick=1) for lv in H.levels], 2)
    assert W.levels[1].nboxes == 1
    g2, _ = _oracle_all(oracle, W, _on_tiling(states, W, 2), per, sym, None)
    assert not bits_equal(g0[1], g2[1])


def test_retile_limits_policy():
    """pa_hierarchy_retile_limits[_ranks]: 512 x 256 x 256 where every level is made of blocks >= 128 cells thick, else 128^3;
    sharded: the largest of 512 x 256 x 256 / 256^3 / 256 x 256 x 128 / 128^3 that leaves every level at least four boxes per rank"""
    from peleanalysis_amd.hierarchy import nested_hierarchy, retile_hierarchy, tagged_hierarchy
    H = nested_hierarchy(512, 3, 64)
    assert [lv.nboxes for lv in retile_hierarchy(H).levels] == [4, 4, 4]  # 512 x 256 x 256
    assert [lv.nboxes for lv in retile_hierarchy(H, nranks=2).levels] == [8, 8, 8]
    assert [lv.nboxes for lv in retile_hierarchy(H, nranks=4).levels] == [16, 16, 16]
    assert [lv.nboxes for lv in retile_hierarchy(H, nranks=8).levels] == [64, 64, 64]
    for n in (1, 2, 4, 8):
        for lv, tv in zip(H.levels, retile_hierarchy(H, nranks=n).levels):
            assert tv.ncells == lv.ncells and (n == 1 or tv.nboxes >= 4 * n)
    Hi = tagged_hierarchy(256, 3, lambda x, y, z: field_flame(x, y, z, 0), bf=16, max_box=128, base_box=128, frac=(0.08, 0.16), is_per=(1, 1, 0))
    T = retile_hierarchy(Hi)  # a flame sheet of 32-cell blocks: 128^3 limits on every level
    for lv, tv in zip(Hi.levels, T.levels):
        n = tv.boxes[:, 3:] - tv.boxes[:, :3] + 1
        assert tv.ncells == lv.ncells and n.max() <= 128
        # x-runs are cut into whole 64-cell tiles + at most one remainder box of <= 32 cells
        assert set(np.unique(n[:, 0] % 64)) <= {0, 32} and ((n[:, 0] % 64 == 0) | (n[:, 0] == 32)).all()


@pytest.mark.gpu
def test_upload_and_download_of_component_ranges(ctx):
    """pa_mf_upload_comps / pa_mf_download_comps move the named components only (grad3d: inputs up, outputs down)"""
    import ctypes as C
    from peleanalysis_amd import capi
    from peleanalysis_amd.hierarchy import nested_hierarchy
    lv = nested_hierarchy(16, 1, 8).levels[0]
    rng = np.random.default_rng(3)
    host = MultiFab(lv, 5, 1)
    host.data[:] = rng.random(host.total)
    dl = capi.DevLevel(ctx, lv)
    d = capi.DevMF(ctx, dl, 5, 1)
    ctx.check(ctx.lib.pa_mf_setval(ctx.h, d.h, 0, 5, -7.0))
    ctx.check(ctx.lib.pa_mf_upload_comps(ctx.h, d.h, host.data.ctypes.data_as(C.c_void_p), 1, 2))
    got = d.download()
    for b in range(lv.nboxes):
        assert np.array_equal(got.fab(b)[1:3], host.fab(b)[1:3]) and (got.fab(b)[0] == -7.0).all() and (got.fab(b)[3:] == -7.0).all()
    back = MultiFab(lv, 5, 1, fill=9.0)
    ctx.check(ctx.lib.pa_mf_download_comps(ctx.h, d.h, back.data.ctypes.data_as(C.c_void_p), 2, 3))
    for b in range(lv.nboxes):
        assert np.array_equal(back.fab(b)[2], host.fab(b)[2]) and (back.fab(b)[3:] == -7.0).all() and (back.fab(b)[:2] == 9.0).all()
    assert ctx.lib.pa_mf_upload_comps(ctx.h, d.h, host.data.ctypes.data_as(C.c_void_p), 4, 2) != 0  # component range


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("interp", [0, 1])
def test_ghost_fill_of_a_hierarchy_equals_the_per_level_calls(ctx, seed, interp):
    """pa_fill_ghosts_hierarchy (FillBoundary / FillPatchTwoLevels / foextrap of all levels, one launch each) against the sequence
    filterPlt.cpp:159-203 makes per level, every ghost cell bit for bit; ghost widths 1 / 2 / 4 as filterPlt's levels"""
    import ctypes as C
    from peleanalysis_amd import capi
    H, per, sym, fn = _draw_h(seed)
    fields = make_states(H, 2, 0, fn, seed=seed + 9)
    ngs = [1, 2, 4][:H.nlev]
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]

    def upload():
        out = []
        for l, (lv, dl) in enumerate(zip(H.levels, dls)):
            m = MultiFab(lv, 2, ngs[l], fill=-666.0)
            for b in range(lv.nboxes):
                m.valid(b)[:] = fields[l].valid(b)
            out.append(capi.DevMF.from_host(ctx, dl, m))
        return out
    a, b_ = upload(), upload()
    for l in range(H.nlev):
        ctx.check(ctx.lib.pa_fill_boundary(ctx.h, a[l].h, 0, 2, ngs[l]))
        if l > 0:
            ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, a[l].h, a[l - 1].h, 0, 2, ngs[l], 2, interp))
        ctx.check(ctx.lib.pa_foextrap(ctx.h, a[l].h, 0, 2, ngs[l]))
    hm = (C.c_void_p * H.nlev)(*[x.h for x in b_])
    hg = (C.c_int32 * H.nlev)(*ngs)
    ctx.check(ctx.lib.pa_fill_ghosts_hierarchy(ctx.h, H.nlev, hm, 0, 2, hg, 2, interp, 1))
    ctx.sync()
    nbad = ctx.bc_errors()  # improperly nested ghost shells (wide stencils on small random hierarchies) are counted by both paths alike
    for l in range(H.nlev):
        assert np.array_equal(a[l].download().data.view(np.int64), b_[l].download().data.view(np.int64)), f"seed {seed} interp {interp} level {l} (bad cells {nbad})"
