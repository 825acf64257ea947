"""GPU parity against the ORACLE at the geometry the benchmark runs (not only against sibling kernels): 128^3 boxes -- two x
tiles x ten row tiles x two z segments per box in the fused sweep, all levels in one launch (k_gradcurv_march3_levels),
XCD-aware block order; 130^3 marching-cubes FABs; the separable filter's 128-wide strips with fgr 2 / 4 / 8.  The oracle runs
with OpenMP over boxes (a few seconds each).  Reference call sites: grad.cpp:211-236, curvature.cpp:451-567,
isosurface.cpp:1531-1592, filterPlt.cpp:217."""
import ctypes as C
import os

import numpy as np
import pytest

from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import MultiFab, chop_box, field_flame, fill_analytic, nested_hierarchy, Hierarchy, Level
from util import assert_filter_parity, assert_valid_bits_equal

pytestmark = pytest.mark.gpu


def _omp():
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))


@pytest.mark.parametrize("thr", [None, 0.05])
def test_gradcurv_128_boxes_three_levels_matches_oracle(ctx, oracle, thr):
    """3 levels of 256^3 cells in 128^3 boxes (8 per level): the headline's tiling and launch shape, every output of every
    cell against the oracle bit for bit; with the threshold clip too (CG + CLIP sweep, clip-aware fix-up)"""
    _omp()
    H = nested_hierarchy(256, 3, 128, is_per=(1, 1, 0))
    bc = capi.bc_from_flags((1, 1, 0))
    rng = np.random.default_rng(4)
    states = []
    for lv in H.levels:
        s = MultiFab(lv, 1, 2)
        fill_analytic(s, 0, lambda x, y, z: field_flame(x, y, z, 0))
        for b in range(lv.nboxes):
            v = s.valid(b)
            v += 1e-3 * rng.uniform(-1, 1, size=v.shape)
        states.append(s)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=False, omp=True)
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, prog_min=300.0, prog_max=2000.0, threshold=thr, omp=True)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=thr, fused=True), work, dout, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
    assert kn.startswith("k_gradcurv_march3_levels<MTY=13") and "3 levels per launch" in kn, kn
    for l in range(H.nlev):
        got = dout[l].download()
        assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"128^3 boxes thr {thr} grad level {l}")
        assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"128^3 boxes thr {thr} curv level {l}")
    # the gradient tool's sweep (k_grad_march<13>) on the same boxes
    dgr = [capi.DevMF(ctx, dl, 4, 0) for dl in dls]
    capi.grad_run(ctx, dst, 0, bc, dgr, 0)
    ctx.sync()
    for l in range(H.nlev):
        assert_valid_bits_equal(dgr[l].download(), og[l], [(c, c) for c in range(4)], f"128^3 boxes grad_run level {l}")


def test_marching_cubes_130_cubed_fab_matches_oracle(ctx, oracle):
    """level-batched marching cubes on FABs of 130^3 (128^3 boxes + 1 ghost layer, the benchmark's shape): two levels so
    that the finer level's mask and the coarse-fine ghost values are exercised; vertices bit for bit, keys and connectivity
    identical, per FAB"""
    H = nested_hierarchy(256, 2, 128, is_per=(0, 0, 0))
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
        fill_analytic(st, 3, lambda x, y, z: field_flame(x, y, z, 0))
        oracle.fill_boundary(st, 0, nc, ng)
        if l > 0:
            assert oracle.lib().orc_fillpatch_two_levels(C.byref(oracle._mf(st)), C.byref(oracle._mf(states[l - 1])), 0, nc, ng, 2, 0) == 0
        states.append(st)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    ntri = 0
    for l, lv in enumerate(H.levels):
        dst = capi.DevMF.from_host(ctx, dls[l], states[l])
        loops = np.zeros((lv.nboxes, 6), np.int64)
        want = []
        for b in range(lv.nboxes):
            lo, hi, mask, llo, lhi = oracle.iso_fab_inputs(H.levels, states, l, b, ng)
            loops[b, :3], loops[b, 3:] = llo, lhi
            want.append(oracle.mc_fab(np.ascontiguousarray(states[l].fab(b)), mask, lo, hi, 3, 1150.0, llo, lhi) if b < 3 or l > 0 else None)
        got = capi.mc_level(ctx, dst, dls[l + 1] if l + 1 < H.nlev else None, loops, 3, 1150.0)
        for b in range(lv.nboxes):
            if want[b] is None:
                continue
            (v, k, t), (gv, gk, gt) = want[b], got[b]
            assert (len(gv), len(gt)) == (len(v), len(t)), f"level {l} box {b}: counts differ"
            assert np.array_equal(gk, k) and np.array_equal(gt, t), f"level {l} box {b}: keys / connectivity differ"
            assert np.array_equal(gv.view(np.int64), np.ascontiguousarray(v).view(np.int64)), f"level {l} box {b}: vertex data differ"
            ntri += len(t)
    assert ntri > 50000


@pytest.mark.parametrize("fgr", [2, 4, 8])
def test_box_filter_128_boxes_matches_oracle(ctx, oracle, filter_mode, fgr):
    """one 256^3 level of 128^3 boxes through the box filter with fgr 2 / 4 / 8 (ng 1 / 2 / 4): the separable kernel's strips
    over 128-wide boxes (<= 1e-12 of the oracle's tap-order sum) and the tap-order kernels (bit for bit)"""
    _omp()
    ngf = fgr // 2
    lv = Level(chop_box((0, 0, 0), (255, 255, 255), 128), (0, 0, 0), (255, 255, 255), (1, 1, 0), np.zeros(3), np.ones(3))
    H = Hierarchy([lv], 2)
    rng = np.random.default_rng(fgr)
    ins = MultiFab(lv, 1, ngf)
    fill_analytic(ins, 0, lambda x, y, z: field_flame(x, y, z, 0))
    for b in range(lv.nboxes):
        v = ins.valid(b)
        v += 1e-3 * rng.uniform(-1, 1, size=v.shape)
    oin, oout = ins.copy(), MultiFab(lv, 1, 0)
    info = oracle.filter_pipeline(H.levels, [oin], [oout], 1, base_fgr=fgr, same_fgr_all_levels=True, omp=True)
    assert info[0] == (fgr, ngf)
    dl = capi.DevLevel(ctx, lv)
    din = capi.DevMF.from_host(ctx, dl, ins)
    ctx.check(ctx.lib.pa_fill_boundary(ctx.h, din.h, 0, 1, ngf))
    ctx.check(ctx.lib.pa_foextrap(ctx.h, din.h, 0, 1, ngf))
    dout = capi.DevMF(ctx, dl, 1, 0)
    w = (C.c_double * (2 * ngf + 1))()
    assert ctx.lib.pa_box_filter_weights(fgr, w) == ngf
    ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, din.h, dout.h, 0, 1, ngf, w))
    ctx.sync()
    assert_filter_parity(dout.download(), oout, [(0, 0)], f"128^3 boxes fgr {fgr}", filter_mode)


def test_gradcurv_tagged_irregular_hierarchy_matches_oracle(ctx, oracle):
    """an irregular hierarchy built the way the bench's `irregular_amr` is (levels 1-2 = the blocks with the largest |grad T|,
    greedy boxes of 1-4 blocks per side: L-shaped regions, mixed faces, concave corners, boxes of 16..64 cells next to each other,
    wide and narrow sweep groups on one level) at a size the OpenMP oracle does in seconds: every output of every cell bit for
    bit, with the threshold clip, fused and pass by pass"""
    _omp()
    from peleanalysis_amd.hierarchy import tagged_hierarchy
    H = tagged_hierarchy(128, 3, lambda x, y, z: field_flame(x, y, z, 0), bf=8, max_box=64, base_box=64, frac=(0.10, 0.18), is_per=(1, 1, 0))
    assert H.nlev == 3
    widths = [set(int(w) for w in (lv.boxes[:, 3] - lv.boxes[:, 0] + 1)) for lv in H.levels]
    assert any(w <= 32 for w in widths[2]) and any(w > 32 for w in widths[2])  # both sweep groups on the finest level
    bc = capi.bc_from_flags((1, 1, 0))
    rng = np.random.default_rng(8)
    states = []
    for lv in H.levels:
        s = MultiFab(lv, 1, 2)
        fill_analytic(s, 0, lambda x, y, z: field_flame(x, y, z, 0))
        for b in range(lv.nboxes):
            v = s.valid(b)
            v += 1e-3 * rng.uniform(-1, 1, size=v.shape)
        states.append(s)
    thr = 0.04
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=False, omp=True)
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, prog_min=300.0, prog_max=2000.0, threshold=thr, omp=True)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    nirr = [int(ctx.lib.pa_level_irregular_cells(ctx.h, dl.h)) for dl in dls]
    assert nirr[0] == 0 and nirr[1] > 0 and nirr[2] > 0, nirr
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    for fused in (True, False):
        dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=300.0, prog_max=2000.0, threshold=thr, fused=fused), work, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        if fused:
            kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
            assert "_levels<" in kn or "CG=1" in kn, kn  # the exact-normal pipeline (wide + narrow sweep groups), not the pass-by-pass fallback
        for l in range(H.nlev):
            got = dout[l].download()
            assert_valid_bits_equal(got, og[l], [(c, c) for c in range(4)], f"tagged irregular fused {fused} grad level {l}")
            assert_valid_bits_equal(got, oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"tagged irregular fused {fused} curv level {l} ({nirr} irregular cells)")


@pytest.mark.parametrize("shape", ["nested128", "widths"])
def test_x_face_mirror_of_the_sweep_gives_the_same_bits(ctx, oracle, shape, monkeypatch):
    """round 5, third session: the wide exact-normal sweep hands the fix-up what the first cell behind a special X face needs (N_x of the
    first three cells, the y and z terms of K) through face-major arrays (MarchArgs::ncg; PA_NCG=0: the fix-up reads the normals' FABs as
    before).  Both ways bit for bit, with the path asserted -- the parity tests against the oracle run with the default (on).
    "widths": separate fine boxes 36, 66, 68 and 130 cells wide -- high faces whose last tile is 36, 2, 4 and 2 columns wide: the mirror
    takes only tiles that hold three columns, the others keep the old path (the same rule on both sides: ncg_face_ok)"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
    per = (0, 1, 0)
    if shape == "nested128":
        H = nested_hierarchy(128, 3, 64, is_per=(1, 1, 0))
        per = (1, 1, 0)
    else:
        # level 0: 192 x 32 x 32 in one row of boxes; level 1: boxes of the widths under test side by side in y (each its own x extent)
        l0 = Level(chop_box((0, 0, 0), (191, 31, 31), 64), (0, 0, 0), (191, 31, 31), per, np.zeros(3), np.array([6.0, 1.0, 1.0]))
        widths, boxes, y = [36, 66, 68, 130], [], 2
        for w in widths:  # fine boxes: x from 20, 12 rows each with 4 rows of coarse cells between them, z 8 .. 55 (even corners: ratio 2)
            boxes.append([20, y, 8, 20 + w - 1, y + 11, 55])
            y += 16
        l1 = Level(np.array(boxes, np.int32), (0, 0, 0), (383, 63, 63), per, np.zeros(3), np.array([6.0, 1.0, 1.0]))
        H = Hierarchy([l0, l1], 2)
    bc = capi.bc_from_flags(per)
    rng = np.random.default_rng(9)
    states = []
    for lv in H.levels:
        s = MultiFab(lv, 1, 2)
        fill_analytic(s, 0, lambda x, y, z: field_flame(x / (6.0 if shape == "widths" else 1.0), y, z, 0))
        for b in range(lv.nboxes):
            v = s.valid(b)
            v += 1e-3 * rng.uniform(-1, 1, size=v.shape)
        states.append(s)
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, states)]
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    outs = {}
    for v in ("0", "1"):
        monkeypatch.setenv("PA_NCG", v)
        capi.reload_options()
        dout = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
        for m in dout:
            m.setval(-3.0)
        capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(prog_min=300.0, prog_max=2000.0, fused=True), work, dout, 0)
        ctx.sync()
        assert ctx.bc_errors() == 0
        kn = ctx.lib.pa_sweep_kernel_name(ctx.h).decode()
        assert ("x faces mirrored" in kn) == (v == "1"), kn
        outs[v] = [m.download() for m in dout]
    monkeypatch.delenv("PA_NCG")
    capi.reload_options()
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            assert np.array_equal(outs["0"][l].valid(b).view(np.int64), outs["1"][l].valid(b).view(np.int64)), (shape, l, b)
    _omp()
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=False, omp=True)
    oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, prog_min=300.0, prog_max=2000.0, omp=True)
    for l in range(H.nlev):
        assert_valid_bits_equal(outs["1"][l], og[l], [(c, c) for c in range(4)], f"{shape} grad level {l}")
        assert_valid_bits_equal(outs["1"][l], oc[l], [(4, 2), (5, 3), (6, 4), (7, 1)], f"{shape} curv level {l}")
