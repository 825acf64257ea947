"""GPU parity on randomly drawn hierarchies (fixed seeds): odd domain extents, uneven box chops, fine regions touching or
crossing periodic faces and walls, two or three levels, every boundary-condition mix, with and without the threshold.
The gradient tool's kernels, the fused sweep + face fix-up and the pass-by-pass kernels against the CPU oracle, bit for
bit -- the point is shapes nobody wrote a dedicated case for."""
import numpy as np
import pytest

from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import Hierarchy, Level, MultiFab, chop_box, field_flame, field_trig
from util import assert_filter_parity, assert_valid_bits_equal, make_states

pytestmark = pytest.mark.gpu


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    n = rng.integers(10, 36, size=3)
    per = rng.integers(0, 2, size=3)
    sym = np.where(per == 1, 0, rng.integers(0, 2, size=3))
    dom_hi = n - 1
    levels = [Level(chop_box((0, 0, 0), dom_hi, int(rng.integers(5, 20))), (0, 0, 0), dom_hi, per, np.zeros(3), np.ones(3))]
    lo, hi = np.zeros(3, dtype=np.int64), dom_hi.astype(np.int64)  # region of the current level in ITS index space
    nlev = int(rng.integers(2, 4))
    for l in range(1, nlev):
        ext = hi - lo + 1
        if np.any(ext < 10):
            break
        # refined region in the coarse index space: at least 2 coarse cells inside the coarse region on every side, so
        # that the fine ghost cells (2 layers = 1 coarse cell) find coarse data; sometimes flush with a periodic face
        clo = lo + rng.integers(2, np.maximum(3, ext // 3))
        chi = hi - rng.integers(2, np.maximum(3, ext // 3))
        if l == 1:
            for d in range(3):
                if per[d] and rng.random() < 0.4:
                    clo[d] = lo[d]  # touches the periodic face: its ghost cells wrap to coarse cells on the far side
        if np.any(chi - clo < 2):
            break
        flo, fhi = 2 * clo, 2 * chi + 1
        size = int(rng.choice([4, 6, 8, 10, 14, 16]))
        domhi_f = 2 * (np.asarray(levels[-1].domhi) + 1) - 1
        levels.append(Level(chop_box(flo, fhi, size), (0, 0, 0), domhi_f, per, np.zeros(3), np.ones(3)))
        lo, hi = flo, fhi
    return Hierarchy(levels, 2), tuple(int(x) for x in per), tuple(int(x) for x in sym), (field_flame if seed % 2 else field_trig)


import os

@pytest.fixture(autouse=True)
def _free_device_objects(monkeypatch):
    """every DevMF / DevLevel a test of this file makes is destroyed after it (the python wrappers have no finaliser): a hunt of a
    thousand cases (PA_RANDOM_*_SEEDS) otherwise ends in `out of device memory` for a 3-MB multifab -- not the card's 288 GB, the
    process's limit on separate allocations (profiles/r05_random_hunt.txt)"""
    made = []
    for cls in (capi.DevMF, capi.DevLevel):
        orig = cls.__init__

        def init(self, *a, _orig=orig, **k):
            _orig(self, *a, **k)
            made.append(self)

        monkeypatch.setattr(cls, "__init__", init)
    yield
    for o in reversed(made):
        if isinstance(o, capi.DevMF):
            o.close()
    for o in reversed(made):
        if isinstance(o, capi.DevLevel):
            o.close()


NSEEDS = int(os.environ.get("PA_RANDOM_SEEDS", "16"))  # PA_RANDOM_SEEDS=200 for a longer hunt


@pytest.mark.parametrize("seed", range(NSEEDS))
def test_random_hierarchy_matches_oracle(ctx, oracle, seed):
    H, per, sym, fn = _draw(seed)
    thr = None if seed % 3 else 0.03
    states = make_states(H, 1, 2, fn, seed=seed)
    bc = capi.bc_from_flags(per, sym)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=True)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=thr)
    tag = f"seed {seed}: {[tuple(lv.domhi + 1) for lv in H.levels]} per {per} sym {sym} boxes {[lv.nboxes for lv in H.levels]}"
    for fused in (True, False):
        dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
        dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
        work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
        dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=thr, fused=fused), work, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0, tag
        for l in range(H.nlev):
            got = dout[l].download()
            assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"{tag} fused {fused} grad level {l}")
            assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"{tag} fused {fused} curv level {l}")
    # the gradient tool's own pipeline (1 ghost layer is enough for it)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    dgr = [capi.DevMF(ctx, dl, 4, 0) for dl in dls]
    capi.grad_run(ctx, dst, 0, bc, dgr, 0)
    ctx.sync()
    for l in range(H.nlev):
        assert_valid_bits_equal(dgr[l].download(), og[l], [(c, c) for c in range(4)], f"{tag} grad_run level {l}")


def _draw_any(seed):
    """seeds 0 .. NSEEDS-1: one rectangular refined region per level (_draw); NSEEDS ..: unions of rectangles (general BoxArrays)"""
    if seed < NSEEDS:
        return _draw(seed)
    from peleanalysis_amd.hierarchy import union_hierarchy
    H = union_hierarchy(7000 + seed - NSEEDS)
    rng = np.random.default_rng(9000 + seed)
    per = tuple(int(x) for x in H.levels[0].is_per)
    sym = tuple(int(x) for x in np.where(np.asarray(per) == 1, 0, rng.integers(0, 2, size=3)))
    return H, per, sym, (field_flame if seed % 2 else field_trig)


NUNION_TOOLS = int(os.environ.get("PA_RANDOM_UNION_TOOL_SEEDS", "8"))


@pytest.mark.parametrize("seed", range(NSEEDS + NUNION_TOOLS))
def test_random_hierarchy_isosurface_and_filter(ctx, oracle, filter_mode, seed):
    """the same random hierarchies through the level-batched marching cubes (mask evaluated from the finer level, periodic
    images included; 1 or 2 ghost layers) and through the box filter with its ghost fill (conservative-linear or
    piecewise-constant; 27 or 125 taps), against the oracle per FAB"""
    import ctypes as C
    H, per, sym, fn = _draw_any(seed)
    rng = np.random.default_rng(77 + seed)
    ng, nc = int(rng.integers(1, 3)), 5
    fields = make_states(H, 2, 0, fn, seed=seed + 1)
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
    allv = np.concatenate([s.valid(b)[3].ravel() for s in states for b in range(s.level.nboxes)])
    iso = float(np.quantile(allv, 0.4))
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    ntri = 0
    frags = []
    for l, lv in enumerate(H.levels):
        dst = capi.DevMF.from_host(ctx, dls[l], states[l])
        loops, want = np.zeros((lv.nboxes, 6), np.int64), []
        for b in range(lv.nboxes):
            lo, hi, mask, llo, lhi = oracle.iso_fab_inputs(H.levels, states, l, b, ng)
            loops[b, :3], loops[b, 3:] = llo, lhi
            want.append(oracle.mc_fab(np.ascontiguousarray(states[l].fab(b)), mask, lo, hi, 3, iso, llo, lhi))
        got = capi.mc_level(ctx, dst, dls[l + 1] if l + 1 < H.nlev else None, loops, 3, iso)
        for b in range(lv.nboxes):
            (v, k, t), (gv, gk, gt) = want[b], got[b]
            assert (len(gv), len(gt)) == (len(v), len(t)), f"seed {seed} level {l} box {b}: counts differ"
            assert np.array_equal(gk, k) and np.array_equal(gt, t), f"seed {seed} level {l} box {b}: keys / connectivity differ"
            assert np.array_equal(gv.view(np.int64), np.ascontiguousarray(v).view(np.int64)), f"seed {seed} level {l} box {b}: vertex data differ"
            ntri += len(t)
            if len(gt):
                frags.append((gv, gt))
    assert ntri > 0
    # the global node / element sets (pa_iso_merge) against the oracle's sequential insertion (ng = 1: no per-FAB trimming);
    # a surface whose nearby vertices do not form transitive clusters is handed back by the device path (None): not seen so far
    if ng == 1 and sum(len(v) for v, _ in frags) < 6000:
        wn, we = oracle.iso_merge(frags, nc)
        got = capi.iso_merge(ctx, frags, nc)
        if got is not None:
            assert got[0].shape == wn.shape and np.array_equal(got[0].view(np.int64), np.ascontiguousarray(wn).view(np.int64)), f"seed {seed}: merged nodes differ"
            assert np.array_equal(got[1], we), f"seed {seed}: merged elements differ"
    # box filter, the same width on every level (fgr 2 or 4: 1 or 2 ghost layers -- the random fine regions keep 2 coarse
    # cells to their level's edge, which is what 2 fine ghost layers + the interpolation stencil need)
    interp = int(rng.integers(0, 2))
    base_fgr = int(rng.choice([2, 4]))
    ngl = [base_fgr // 2] * H.nlev
    ins = [MultiFab(lv, 2, g, fill=0.0) for lv, g in zip(H.levels, ngl)]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            ins[l].valid(b)[:] = fields[l].valid(b)
    oin = [m.copy() for m in ins]
    oouts = [MultiFab(lv, 2, 0) for lv in H.levels]
    info = oracle.filter_pipeline(H.levels, oin, oouts, 2, base_fgr=base_fgr, same_fgr_all_levels=True, interp_type=interp)
    for l, lv in enumerate(H.levels):
        din = capi.DevMF.from_host(ctx, dls[l], ins[l])
        if l == 0:
            dprev = None
        ctx.check(ctx.lib.pa_fill_boundary(ctx.h, din.h, 0, 2, ngl[l]))
        if l > 0:
            ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, din.h, dprev.h, 0, 2, ngl[l], 2, interp))
        ctx.check(ctx.lib.pa_foextrap(ctx.h, din.h, 0, 2, ngl[l]))
        dout = capi.DevMF(ctx, dls[l], 2, 0)
        fgr, ngf = info[l]
        w = (C.c_double * (2 * ngf + 1))()
        assert ctx.lib.pa_box_filter_weights(fgr, w) == ngf == ngl[l]
        ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, din.h, dout.h, 0, 2, ngf, w))
        ctx.sync()
        assert ctx.bc_errors() == 0
        assert_filter_parity(dout.download(), oouts[l], [(0, 0), (1, 1)], f"seed {seed} filter level {l} interp {interp}", filter_mode)
        dprev = din


@pytest.mark.parametrize("seed", range(NSEEDS + NUNION_TOOLS))
def test_random_hierarchy_curvature_options(ctx, oracle, seed):
    """the same hierarchies (and unions of rectangles) through pa_curvature_run with every option on (Gaussian curvature, strain
    rate + tensor, flame-normal velocity; curvature.cpp:575-789) against the oracle, all 17 output components bit for bit"""
    H, per, sym, fn = _draw_any(seed)
    thr = None if seed % 2 else 0.04
    states = make_states(H, 4, 2, fn, seed=seed + 5)  # comp 0 = progress source, 1..3 = velocity
    bc = capi.bc_from_flags(per, sym)
    oout = [MultiFab(lv, 17, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oout, 0, MultiFab, threshold=thr, do_gauss=True, vel_comp=1, do_strain=True,
                              do_velnormal=True, strain_tensor=True)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    dout = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
    for fused in (False, True):  # pass by pass / the exact-normal pipeline's G-output sweeps + one options pass per level
        for m in dout:
            m.setval(-7.0)
        P = capi.curv_params(threshold=thr, fused=fused, do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True, vel_comp=1)
        capi.curvature_run(ctx, dst, 0, bc, P, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        for l in range(H.nlev):
            assert_valid_bits_equal(dout[l].download(), oout[l], [(c, c) for c in range(17)], f"seed {seed} options fused {fused} level {l}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("PA_RANDOM_WIDE_SEEDS", "6"))))
@pytest.mark.parametrize("which", ["all", "gauss", "strain_veln", "veln"])
def test_random_wide_box_curvature_options_fast_path(ctx, oracle, seed, which):
    """pa_curvature_run's fast path on boxes wider than 32 cells (the wide G-output sweep, all levels in one launch; uneven
    chops also leave narrow boxes: both kernels in one pass): every subset of the options the fused options kernel is
    instantiated for, with and without the threshold clip, all written components against the oracle bit for bit; components
    of options that are off stay untouched"""
    H, per, sym, fn = _draw_wide(seed)
    thr = None if seed % 2 else 0.04
    opts = dict(all=dict(do_gauss=True, do_strain=True, strain_tensor=True, do_velnormal=True), gauss=dict(do_gauss=True),
                strain_veln=dict(do_strain=True, do_velnormal=True), veln=dict(do_velnormal=True))[which]
    comps = dict(all=list(range(17)), gauss=[0, 1, 2, 3, 4, 5], strain_veln=[0, 1, 2, 3, 4, 6, 7], veln=[0, 1, 2, 3, 4, 7])[which]
    states = make_states(H, 4, 2, fn, seed=seed + 5)
    bc = capi.bc_from_flags(per, sym)
    oout = [MultiFab(lv, 17, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oout, 0, MultiFab, threshold=thr, vel_comp=1, **opts)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    dout = [capi.DevMF(ctx, dl, 17, 0) for dl in dls]
    for m in dout:
        m.setval(-7.0)
    capi.curvature_run(ctx, dst, 0, bc, capi.curv_params(threshold=thr, fused=True, vel_comp=1, **opts), dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
    assert "_levels<" in kn and ctx.lib.pa_curvature_last_path(ctx.h) in (1, 2), kn  # 2: every box of the draw is wide -- the Gaussian curvature inside the sweeps
    for l in range(H.nlev):
        got = dout[l].download()
        assert_valid_bits_equal(got, oout[l], [(c, c) for c in comps], f"wide seed {seed} options {which} level {l}")
        for c in set(range(17)) - set(comps):
            for b in range(H.levels[l].nboxes):
                assert np.all(got.valid(b)[c] == -7.0), f"wide seed {seed} options {which}: component {c} was written"


def _draw_wide(seed):
    """hierarchies whose boxes are wider than 32 cells (odd widths from uneven chops): the exact-normal pipeline where every
    special face is pure, the first pipeline where a chop leaves a face half covered"""
    rng = np.random.default_rng(5000 + seed)
    n = np.array([int(rng.integers(9, 14)) * 8, int(rng.integers(5, 9)) * 8, int(rng.integers(4, 8)) * 8])
    per = rng.integers(0, 2, size=3)
    sym = np.where(per == 1, 0, rng.integers(0, 2, size=3))
    dom_hi = n - 1
    levels = [Level(chop_box((0, 0, 0), dom_hi, int(rng.integers(36, 60))), (0, 0, 0), dom_hi, per, np.zeros(3), np.ones(3))]
    lo, hi = np.zeros(3, dtype=np.int64), dom_hi.astype(np.int64)
    for l in range(1, 3):
        ext = hi - lo + 1
        clo = lo + rng.integers(2, np.maximum(3, ext // 4))
        chi = hi - rng.integers(2, np.maximum(3, ext // 4))
        clo[0] = min(clo[0], chi[0] - 20)  # at least 42 fine cells in x
        if l == 1 and per[1] and rng.random() < 0.5:
            clo[1] = lo[1]
        if np.any(chi - clo < 6) or clo[0] < lo[0] + 2:
            break
        flo, fhi = 2 * clo, 2 * chi + 1
        domhi_f = 2 * (np.asarray(levels[-1].domhi) + 1) - 1
        levels.append(Level(chop_box(flo, fhi, int(rng.integers(40, 72))), (0, 0, 0), domhi_f, per, np.zeros(3), np.ones(3)))
        lo, hi = flo, fhi
    return Hierarchy(levels, 2), tuple(int(x) for x in per), tuple(int(x) for x in sym), (field_flame if seed % 2 else field_trig)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PA_RANDOM_WIDE_SEEDS", "6"))))
def test_random_wide_box_hierarchy_matches_oracle(ctx, oracle, seed):
    H, per, sym, fn = _draw_wide(seed)
    assert H.nlev >= 2 and min(int((lv.boxes[:, 3] - lv.boxes[:, 0]).min()) + 1 for lv in H.levels) > 16
    states = make_states(H, 1, 2, fn, seed=seed)
    bc = capi.bc_from_flags(per, sym)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=False)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab)
    tag = f"wide seed {seed}: {[tuple(lv.domhi + 1) for lv in H.levels]} per {per} sym {sym} boxes {[(lv.nboxes, int((lv.boxes[:, 3] - lv.boxes[:, 0]).max()) + 1) for lv in H.levels]}"
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(fused=True), work, dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0, tag
    for l in range(H.nlev):
        got = dout[l].download()
        assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"{tag} grad level {l}")
        assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"{tag} curv level {l}")


# ------------------------------------------------------------------------------------------- general BoxArrays
from peleanalysis_amd.hierarchy import union_hierarchy  # noqa: E402

NUNION = int(os.environ.get("PA_RANDOM_UNION_SEEDS", "24"))


def _union_case(seed):
    H = union_hierarchy(7000 + seed)
    rng = np.random.default_rng(9000 + seed)
    per = tuple(int(x) for x in H.levels[0].is_per)
    sym = tuple(int(x) for x in np.where(np.asarray(per) == 1, 0, rng.integers(0, 2, size=3)))
    return H, per, sym, (field_flame if seed % 2 else field_trig)


@pytest.mark.parametrize("seed", range(NUNION))
def test_random_union_hierarchy_matches_oracle(ctx, oracle, seed):
    """fine levels that are UNIONS of rectangles (L / T shapes, disjoint patches, faces partly covered by a neighbour,
    concave coarse-fine corners -- what a Pele plotfile holds): the exact-normal pipeline with its irregular-cell list, the
    pass-by-pass kernels and the gradient tool's pipeline against the oracle, bit for bit, with and without the threshold"""
    H, per, sym, fn = _union_case(seed)
    thr = None if seed % 3 else 0.03
    states = make_states(H, 1, 2, fn, seed=seed)
    bc = capi.bc_from_flags(per, sym)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=True)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=thr)
    tag = f"union seed {seed}: {[tuple(lv.domhi + 1) for lv in H.levels]} per {per} sym {sym} boxes {[lv.nboxes for lv in H.levels]}"
    for fused in (True, False):
        dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
        dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
        work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
        dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=thr, fused=fused), work, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0, tag
        if fused:  # the exact-normal pipeline took it, whatever the shape
            kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
            assert "CG=1" in kn or "_levels<" in kn, (tag, kn)
        for l in range(H.nlev):
            got = dout[l].download()
            assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"{tag} fused {fused} grad level {l}")
            assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"{tag} fused {fused} curv level {l}")
    dgr = [capi.DevMF(ctx, dl, 4, 0) for dl in dls]
    capi.grad_run(ctx, dst, 0, bc, dgr, 0)
    ctx.sync()
    for l in range(H.nlev):
        assert_valid_bits_equal(dgr[l].download(), og[l], [(c, c) for c in range(4)], f"{tag} grad_run level {l}")


def test_union_hierarchies_have_irregular_cells(ctx):
    """the draws above do contain what they are for: cells the sweep + face fix-up cannot get right (and the nested, convex
    hierarchies of the contract configs contain none)"""
    from peleanalysis_amd.hierarchy import nested_hierarchy
    tot = 0
    for seed in range(NUNION):
        H, _, _, _ = _union_case(seed)
        for lv in H.levels[1:]:
            dl = capi.DevLevel(ctx, lv)
            n = ctx.lib.pa_level_irregular_cells(ctx.h, dl.h)
            assert n >= 0
            tot += n
    assert tot > 500, tot
    H = nested_hierarchy(32, 3, 16, is_per=(1, 1, 0))
    for lv in H.levels:
        assert ctx.lib.pa_level_irregular_cells(ctx.h, capi.DevLevel(ctx, lv).h) == 0


def _union_wide_case(seed):
    """unions of rectangles with boxes wider than 32 cells (the wide sweep kernel, several levels per launch): blocks of
    6-12 coarse cells, boxes of 1-4 blocks along x"""
    for attempt in range(50):
        rng = np.random.default_rng(11000 + seed + 1000 * attempt)
        n0 = np.array([int(rng.integers(9, 13)) * 8, int(rng.integers(4, 7)) * 8, int(rng.integers(4, 6)) * 8])
        per = rng.integers(0, 2, size=3)
        H = union_hierarchy(12000 + seed + 1000 * attempt, nlev=3, n0=n0, is_per=per, nrect=(2, 5), block=(8, 13) if seed % 2 else (6, 9), max_blocks=(4, 2, 2),
                            base_box=int(rng.integers(40, 60)))
        if H.nlev >= 2 and all(int((lv.boxes[:, 3] - lv.boxes[:, 0]).max()) + 1 > 32 for lv in H.levels):
            break
    sym = tuple(int(x) for x in np.where(per == 1, 0, rng.integers(0, 2, size=3)))
    return H, tuple(int(x) for x in per), sym, (field_flame if seed % 2 else field_trig)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PA_RANDOM_UNION_WIDE_SEEDS", "8"))))
def test_random_union_wide_box_hierarchy_matches_oracle(ctx, oracle, seed):
    """general BoxArrays with boxes up to 4 blocks wide: the wide sweep kernel (one launch for all levels where the levels
    agree on the tile shape) + irregular-cell list; single component, then two components as one batch (component slots of
    the boundary kernels and of k_curv_general)"""
    H, per, sym, fn = _union_wide_case(seed)
    assert H.nlev >= 2
    thr = None if seed % 2 else 0.04
    states = make_states(H, 2, 2, fn, seed=seed)
    bc = capi.bc_from_flags(per, sym)
    tag = f"union-wide seed {seed}: {[tuple(lv.domhi + 1) for lv in H.levels]} per {per} sym {sym} boxes {[(lv.nboxes, int((lv.boxes[:, 3] - lv.boxes[:, 0]).max()) + 1) for lv in H.levels]}"
    og, oc = [], []
    for c in range(2):
        og.append([MultiFab(lv, 4, 0) for lv in H.levels])
        oc.append([MultiFab(lv, 5, 0) for lv in H.levels])
        oracle.grad_pipeline(H.levels, [s.copy() for s in states], c, bc, og[c], 0, multipass=False)
        oracle.curvature_pipeline(H.levels, [s.copy() for s in states], c, bc, oc[c], 0, MultiFab, threshold=thr)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 16, 0) for dl in dls]
    nirr = sum(ctx.lib.pa_level_irregular_cells(ctx.h, dl.h) for dl in dls)
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=thr, fused=True), work, dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0, tag
    kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
    assert "CG=1" in kn or "_levels<" in kn, (tag, kn)
    for l in range(H.nlev):
        got = dout[l].download()
        assert_valid_bits_equal(got, og[0][l], [(c, c) for c in range(4)], f"{tag} ({nirr} irregular cells) grad level {l}")
        assert_valid_bits_equal(got, oc[0][l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"{tag} ({nirr} irregular cells) curv level {l}")
    capi.gradcurv_run_comps2(ctx, dst, 0, 2, bc, capi.curv_params(threshold=thr, fused=True), work, dout, 0, 2)
    ctx.sync()
    assert ctx.bc_errors() == 0, tag
    for l in range(H.nlev):
        got = dout[l].download()
        for c in range(2):
            assert_valid_bits_equal(got, og[c][l], [(8 * c + q, q) for q in range(4)], f"{tag} batch comp {c} grad level {l}")
            assert_valid_bits_equal(got, oc[c][l], [(8 * c + 4, 2), (8 * c + 5, 3), (8 * c + 6, 4), (8 * c + 7, 1)], f"{tag} batch comp {c} curv level {l}")
