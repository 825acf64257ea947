"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the oracle).
CPU tier: the oracle still reproduces them bit for bit.  GPU tier: the HIP path reproduces them."""
import os

import numpy as np
import pytest

from peleanalysis_amd.hierarchy import Hierarchy, Level, MultiFab

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    d = np.load(os.path.join(GOLD, name))
    levels = []
    for l in range(int(d["nlev"])):
        dom = d[f"dom{l}"]
        levels.append(Level(d[f"boxes{l}"], dom[0:3], dom[3:6], dom[6:9], (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
    return d, Hierarchy(levels, 2)


def _mf_from(lv, arr, ng, ncomp=None):
    """arr: [nboxes][(ncomp)][nz][ny][nx] valid data"""
    if arr.ndim == 4:
        arr = arr[:, None]
    mf = MultiFab(lv, arr.shape[1] if ncomp is None else ncomp, ng)
    for b in range(lv.nboxes):
        mf.valid(b)[:arr.shape[1]] = arr[b]
    return mf


def _same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(b).view(np.int64))


def _per(H):
    return tuple(int(x) for x in H.levels[0].is_per)


# ------------------------------------------------------------------------------------ CPU tier
def test_oracle_reproduces_gradcurv_fixture(oracle):
    d, H = _load("gradcurv_amr3.npz")
    bc = oracle.bc_from_flags(_per(H))
    st = [_mf_from(lv, d[f"in{l}"], 2) for l, lv in enumerate(H.levels)]
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, [s.copy() for s in st], 0, bc, og, 0)
    pm = oracle.curvature_pipeline(H.levels, [s.copy() for s in st], 0, bc, oc, 0, MultiFab, threshold=float(d["threshold"]))
    assert _same(np.array(pm), d["prog_minmax"])
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            assert _same(og[l].valid(b), d[f"grad{l}"][b]) and _same(oc[l].valid(b), d[f"curv{l}"][b])


def test_oracle_reproduces_filter_fixture(oracle):
    d, H = _load("filter_amr2.npz")
    ins = [_mf_from(lv, d[f"in{l}"], 2) for l, lv in enumerate(H.levels)]
    outs = [MultiFab(lv, 1, 0) for lv in H.levels]
    oracle.filter_pipeline(H.levels, ins, outs, 1, base_fgr=2, interp_type=1)
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            assert _same(outs[l].valid(b)[0], d[f"out{l}"][b])


def test_oracle_reproduces_iso_fixture(oracle):
    d, H = _load("iso_amr2.npz")
    fields = [_mf_from(lv, d[f"in{l}"], 0) for l, lv in enumerate(H.levels)]
    nodes, elts = oracle.isosurface_pipeline(H.levels, fields, [0, 1], 0, float(d["isoval"]), MultiFab)
    assert len(d["elts"]) > 100
    assert _same(nodes, d["nodes"]) and np.array_equal(elts, d["elts"])


# ------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_hip_reproduces_gradcurv_fixture(ctx, fused):
    from peleanalysis_amd import capi
    d, H = _load("gradcurv_amr3.npz")
    bc = capi.bc_from_flags(_per(H))
    st = [_mf_from(lv, d[f"in{l}"], 2) for l, lv in enumerate(H.levels)]
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    dst = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, st)]
    work = [capi.DevMF(ctx, dl, 1, 2) for dl in dls]
    out = [capi.DevMF(ctx, dl, 8, 0) for dl in dls]
    capi.gradcurv_run(ctx, dst, 0, bc, capi.curv_params(threshold=float(d["threshold"]), fused=fused), work, out, 0)
    ctx.sync()
    assert ctx.bc_errors() == 0
    for l, lv in enumerate(H.levels):
        got = out[l].download()
        for b in range(lv.nboxes):
            g = got.valid(b)
            assert _same(g[0:4], d[f"grad{l}"][b])
            assert _same(g[4:7], d[f"curv{l}"][b][2:5]) and _same(g[7], d[f"curv{l}"][b][1])


@pytest.mark.gpu
def test_hip_reproduces_filter_fixture(ctx, filter_mode):
    import ctypes as C
    from peleanalysis_amd import capi
    d, H = _load("filter_amr2.npz")
    ins = [_mf_from(lv, d[f"in{l}"], 2) for l, lv in enumerate(H.levels)]
    dls = [capi.DevLevel(ctx, lv) for lv in H.levels]
    din = [capi.DevMF.from_host(ctx, dl, s) for dl, s in zip(dls, ins)]
    dout = [capi.DevMF(ctx, dl, 1, 0) for dl in dls]
    fgr = 2
    for l in range(H.nlev):
        if l > 0:
            fgr *= 2
        w = (C.c_double * (fgr + 2))()
        ngf = ctx.lib.pa_box_filter_weights(fgr, w)
        ctx.check(ctx.lib.pa_fill_boundary(ctx.h, din[l].h, 0, 1, ngf))
        if l > 0:
            ctx.check(ctx.lib.pa_fillpatch_two_levels(ctx.h, din[l].h, din[l - 1].h, 0, 1, ngf, 2, 1))
        ctx.check(ctx.lib.pa_foextrap(ctx.h, din[l].h, 0, 1, ngf))
        ctx.check(ctx.lib.pa_boxfilter_level(ctx.h, din[l].h, dout[l].h, 0, 1, ngf, w))
    ctx.sync()
    for l, lv in enumerate(H.levels):
        got = dout[l].download()
        scale = max(float(np.abs(d[f"out{l}"][b]).max()) for b in range(lv.nboxes))
        for b in range(lv.nboxes):
            if filter_mode == "exact":
                assert _same(got.valid(b)[0], d[f"out{l}"][b])
            else:  # separable default: SURVEY 8(d) metric
                assert float(np.abs(got.valid(b)[0] - d[f"out{l}"][b]).max()) <= 1e-12 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("merge", ["device", "host"])
def test_hip_reproduces_iso_fixture(tmp_path, merge):
    """isosurface.cpp:1434-1728 + 1657-1726 on the HIP path against the committed fixture, no live oracle: the fixture's two
    components go into a plotfile, the drop-in binary (state build, pa_mc_hierarchy_*, node / element sets on the device or --
    PA_ISO_HOST_MERGE=1 -- the sequential host merge) writes its MEF; nodes bit for bit, elements identical (1-based in the file)."""
    import subprocess
    from peleanalysis_amd.plotfile import read_mef, write_plotfile
    d, H = _load("iso_amr2.npz")
    mfs = [_mf_from(lv, d[f"in{l}"], 0) for l, lv in enumerate(H.levels)]
    p = str(tmp_path / "plt00007")
    write_plotfile(p, H, mfs, ["temp", "x_velocity"], time=0.5, level_steps=[7] * H.nlev)
    exe = os.path.join(os.path.dirname(GOLD), "..", "tools", "bin", "isosurface3d.ex")
    env = dict(os.environ, PA_ISO_HOST_MERGE="1") if merge == "host" else None
    out = subprocess.run([exe, "infile=" + p, "isoCompName=temp", "isoVal=%r" % float(d["isoval"]), "comps=0 1", "outfile_base=" + str(tmp_path / "surf")],
                         cwd=tmp_path, capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    _, names, nodes, faces = read_mef(str(tmp_path / "surf.mef"))
    assert names == ["X", "Y", "Z", "temp", "x_velocity"]
    assert _same(nodes, d["nodes"]), "node data not bit-identical to the fixture"
    assert np.array_equal(faces, d["elts"] + 1), "connectivity differs from the fixture"
