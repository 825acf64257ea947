"""Plotfile format: python writer/reader round trip, and the C++ reader/writer of the tool drivers
through template3d.ex (the drop-in for Src/template.cpp; no GPU needed).  GPU tier: the grad /
curvature / filterPlt / isosurface drivers end to end against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from peleanalysis_amd.hierarchy import MultiFab, field_flame, nested_hierarchy
from peleanalysis_amd.plotfile import read_mef, read_plotfile, write_plotfile
from util import make_states

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin")


def _build_tools():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s"])


def _synth(tmp_path, nlev=3, base=16, box=8, ncomp=3, per=(1, 1, 0), names=("temp", "x_velocity", "density")):
    H = nested_hierarchy(base, nlev, box, is_per=per)
    mfs = make_states(H, ncomp, 0, field_flame, seed=77)
    p = str(tmp_path / "plt00005")
    write_plotfile(p, H, mfs, list(names)[:ncomp], time=0.125, level_steps=[5] * nlev)
    return p, H, mfs


def test_python_plotfile_roundtrip(tmp_path):
    p, H, mfs = _synth(tmp_path)
    r = read_plotfile(p, is_per=(1, 1, 0))
    assert r.names == ["temp", "x_velocity", "density"] and r.time == 0.125 and r.hier.nlev == 3
    for l in range(3):
        assert np.array_equal(r.hier.levels[l].boxes, H.levels[l].boxes)
        assert np.array_equal(r.hier.levels[l].domhi, H.levels[l].domhi)
        assert np.array_equal(r.mfs[l].data.view(np.int64), mfs[l].data.view(np.int64))
    hdr = open(os.path.join(p, "Header")).read().split("\n")
    assert hdr[0] == "HyperCLaw-V1.1" and hdr[1] == "3"
    assert open(os.path.join(p, "Level_0", "Cell_H")).read().startswith("1\n1\n3\n0\n(8 0\n((0,0,0) (7,7,7) (0,0,0))")


def test_cpp_template_tool_roundtrip(tmp_path):
    """template3d.ex infile=<plt>: C++ reader -> copy -> C++ writer; python reads the result back"""
    _build_tools()
    p, H, mfs = _synth(tmp_path)
    out = subprocess.run([os.path.join(BIN, "template3d.ex"), "infile=" + p, "is_per=1 1 0"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "Writing new data to plt00005_temp" in out.stdout
    r = read_plotfile(str(tmp_path / "plt00005_temp"))
    assert r.names == ["temp", "x_velocity", "density"] and r.time == 0.0 and r.level_steps == [0, 0, 0]
    for l in range(3):
        assert np.array_equal(r.hier.levels[l].boxes, H.levels[l].boxes)
        assert np.array_equal(r.mfs[l].data.view(np.int64), mfs[l].data.view(np.int64))
    # finestLevel clamps; a missing required key aborts with the ParmParse message
    out = subprocess.run([os.path.join(BIN, "template3d.ex"), "infile=" + p, "finestLevel=0"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0 and read_plotfile(str(tmp_path / "plt00005_temp")).hier.nlev == 1
    # the second run over the first one's output: the old plotfile was renamed at WRITE time and kept, as AMReX's
    # UtilCreateCleanDirectory does -- no level of the three-level run survives in the new one, the old one is whole
    assert sorted(os.listdir(tmp_path / "plt00005_temp")) == ["Header", "Level_0"]
    olds = [f for f in os.listdir(tmp_path) if ".old." in f]
    assert len(olds) == 1 and sorted(os.listdir(tmp_path / olds[0])) == ["Header", "Level_0", "Level_1", "Level_2"]
    # remove_old_output=1: the renamed directory goes away once the new plotfile is complete
    out = subprocess.run([os.path.join(BIN, "template3d.ex"), "infile=" + p, "remove_old_output=1"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0 and [f for f in os.listdir(tmp_path) if ".old." in f] == olds
    assert read_plotfile(str(tmp_path / "plt00005_temp")).hier.nlev == 3
    out = subprocess.run([os.path.join(BIN, "template3d.ex"), "finestLevel=0"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode != 0 and "infile" in out.stderr


def test_old_output_is_only_touched_after_validation(tmp_path):
    """Round-5 advisor finding on pa::OldOutput: (1) a run that aborts on its inputs (missing plotfile, bad Header) must leave an
    earlier output exactly as it was; (2) an existing directory at the output path that is NOT a plotfile (no Header + Level_0)
    is never renamed -- the tool writes into it."""
    _build_tools()
    p, H, mfs = _synth(tmp_path)
    exe = os.path.join(BIN, "template3d.ex")
    assert subprocess.run([exe, "infile=" + p], cwd=tmp_path, capture_output=True, text=True).returncode == 0
    first = _tree_bytes(str(tmp_path / "plt00005_temp"))
    # (1) same output name, input gone: abort, nothing renamed, nothing removed
    bad = tmp_path / "elsewhere"
    bad.mkdir()
    os.symlink(tmp_path / "plt00005_temp", bad / "plt00005_temp")
    out = subprocess.run([exe, "infile=" + str(bad / "plt00005")], cwd=bad, capture_output=True, text=True)
    assert out.returncode != 0 and "Unable to open plotfile Header" in out.stderr
    assert _tree_bytes(str(tmp_path / "plt00005_temp")) == first
    assert not [f for f in os.listdir(tmp_path) + os.listdir(bad) if ".old." in f]
    # a Header that does not parse (2-D file handed to the 3-D build)
    hdr = open(os.path.join(p, "Header")).read().split("\n")
    p2 = tmp_path / "plt2d"
    os.makedirs(p2)
    hdr[2 + 3] = "2"  # HyperCLaw-V1.1, ncomp, 3 names, then the dimension
    (p2 / "Header").write_text("\n".join(hdr))
    (tmp_path / "plt2d_temp").mkdir()
    (tmp_path / "plt2d_temp" / "Header").write_text("an earlier output")
    (tmp_path / "plt2d_temp" / "Level_0").mkdir()
    out = subprocess.run([exe, "infile=" + str(p2)], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode != 0
    assert (tmp_path / "plt2d_temp" / "Header").read_text() == "an earlier output" and not [f for f in os.listdir(tmp_path) if ".old." in f]
    # (2) a directory that is not a plotfile: written into, never renamed
    d = tmp_path / "notplt"
    d.mkdir()
    os.symlink(p, d / "plt00005")
    (d / "plt00005_temp").mkdir()
    (d / "plt00005_temp" / "my_notes.txt").write_text("keep me")
    out = subprocess.run([exe, "infile=plt00005"], cwd=d, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert (d / "plt00005_temp" / "my_notes.txt").read_text() == "keep me" and not [f for f in os.listdir(d) if ".old." in f]
    got = _tree_bytes(str(d / "plt00005_temp"))
    got.pop("my_notes.txt")
    assert got == first


def test_cpp_template_tool_large_level_with_and_without_huge_pages(tmp_path):
    """a level big enough (80^3 x 3 components = 12 MB per multifab, 4 MB per FAB buffer) for the tools' host copies to take the 2-MiB-aligned,
    MADV_HUGEPAGE blocks (tools/common/pa_plotfile.h DefaultInitAlloc; the other CPU-tier cases stay below its 4-MiB threshold):
    template3d writes the same bytes with PA_HOST_THP=1 (default) and 0, re-tiled or not"""
    _build_tools()
    p, H, mfs = _synth(tmp_path, nlev=2, base=80, box=80)
    outs = {}
    for thp in ("1", "0"):
        for retile in ("0", "1"):
            out = subprocess.run([os.path.join(BIN, "template3d.ex"), "infile=" + p, "is_per=1 1 0", "retile=" + retile], cwd=tmp_path, capture_output=True,
                                 text=True, env=dict(os.environ, PA_HOST_THP=thp))
            assert out.returncode == 0, out.stderr
            d = tmp_path / "plt00005_temp"
            outs[(thp, retile)] = {os.path.relpath(os.path.join(r, f), d): open(os.path.join(r, f), "rb").read() for r, _, fs in os.walk(d) for f in fs}
    ref = outs[("1", "0")]
    assert len(ref) >= 5 and sum(len(v) for v in ref.values()) > 20_000_000
    for k, v in outs.items():
        assert v == ref, k
    r = read_plotfile(str(tmp_path / "plt00005_temp"))
    for l in range(2):
        assert np.array_equal(r.mfs[l].data.view(np.int64), mfs[l].data.view(np.int64))


def test_cpp_template_tool_retiled_io_is_byte_identical(tmp_path):
    """the host half of the tools' internal re-tiling without a GPU: template3d.ex retile=1 reads the file's FABs into merged boxes
    (pa_level_retile; small limits through PA_RETILE_MAX so that merged boxes span several file boxes AND file boxes span several
    merged ones) and writes the file's BoxArray back -- every byte, the per-FAB minima / maxima of Cell_H included, as retile=0"""
    _build_tools()
    p, H, mfs = _synth(tmp_path, nlev=3, base=32, box=8)
    ref = None
    for name, args, env in (("file", [], None), ("default", ["retile=1"], None), ("small", ["retile=1"], {"PA_RETILE_MAX": "24 12 16"}),
                            ("tiny", ["retile=1"], {"PA_RETILE_MAX": "5 7 3"})):
        d = tmp_path / name
        d.mkdir()
        out = subprocess.run([os.path.join(BIN, "template3d.ex"), "infile=" + p, "is_per=1 1 0"] + args, cwd=d, capture_output=True, text=True,
                             env=None if env is None else dict(os.environ, **env))
        assert out.returncode == 0, out.stderr
        got = {}
        for root, _, files in os.walk(d / "plt00005_temp"):
            for f in files:
                got[os.path.relpath(os.path.join(root, f), d)] = open(os.path.join(root, f), "rb").read()
        assert len(got) >= 7
        if ref is None:
            ref = got
        else:
            assert got.keys() == ref.keys()
            for k in ref:
                assert got[k] == ref[k], f"template3d {name}: {k} differs"


def test_cpp_mef_to_dat_tool(tmp_path):
    """surfMEFtoDAT3d.ex (host only): the MEF layout written by the python writer is what the consumer parses"""
    _build_tools()
    rng = np.random.default_rng(2)
    nodes = rng.random((7, 5))
    faces = np.array([[1, 2, 3], [3, 4, 5], [5, 6, 7]], dtype=np.int32)
    f = str(tmp_path / "s.mef")
    with open(f, "wb") as fh:  # isosurface.cpp:2097-2134
        fh.write(b"0.25\nX Y Z temp rho\n3 3\n")
        fh.write(b"FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))((0,0,0) (6,0,0) (0,0,0)) 5\n")
        fh.write(nodes.astype("<f8").tobytes())
        fh.write(faces.astype("<i4").tobytes())
    assert read_mef(f)[2].shape == (7, 5)
    out = subprocess.run([os.path.join(BIN, "surfMEFtoDAT3d.ex"), "infile=" + f], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    txt = open(str(tmp_path / "s.dat")).read().split("\n")
    assert txt[0] == "VARIABLES = X Y Z temp rho" and txt[1] == 'ZONE T="0.25" N=7 E=3 F=FEPOINT ET=TRIANGLE'
    got = np.array([[float(t) for t in ln.split()] for ln in txt[2:9]])
    assert np.abs(got - nodes).max() < 1e-5
    assert [[int(t) for t in ln.split()] for ln in txt[9:12]] == faces.tolist()


def _write_mef(path, nodes, faces1, names=b"X Y Z"):
    with open(path, "wb") as fh:  # isosurface.cpp:2097-2134
        fh.write(b"0.5\n" + names + b"\n%d 3\n" % len(faces1))
        fh.write(b"FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))((0,0,0) (%d,0,0) (0,0,0)) %d\n" % (len(nodes) - 1, nodes.shape[1]))
        fh.write(np.ascontiguousarray(nodes, "<f8").tobytes())
        fh.write(np.ascontiguousarray(faces1, "<i4").tobytes())


def _icosphere(nsub=2):
    """closed, consistently oriented triangulation of the unit sphere (subdivided octahedron)"""
    v = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    v = [np.array(p, float) for p in v]
    for _ in range(nsub):
        mid, nf = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p))
                mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        f = nf
    return np.array(v), np.array(f, dtype=np.int32)


def test_cpp_iso_mef_tool(tmp_path, oracle):
    """isoMEF3d.ex (host only; isoMEF.cpp): contour lines of a node variable on a MEF surface -> Tecplot line zones in
    ./out.dat.  Known answers on a sphere: the contour of z is ONE closed line on the plane z = isoVal whose points lie
    between the sphere and the chords of its facets; the contour of x*y at 0.2 is two closed lines; out.dat equals the
    Python restatement (oracle.iso_mef) line for line."""
    _build_tools()
    xyz, faces = _icosphere(3)
    nodes = np.column_stack([xyz, xyz[:, 2] + 0.013, xyz[:, 0] * xyz[:, 1]])
    f = str(tmp_path / "sphere.mef")
    _write_mef(f, nodes, faces + 1, names=b"X Y Z height xy")
    for comp, val, nlines in ((3, 0.35, 1), (4, 0.2, 2), (3, 5.0, 0)):
        out = subprocess.run([os.path.join(BIN, "isoMEF3d.ex"), "infile=" + f, f"isoComp={comp}", f"isoVal={val}"], cwd=tmp_path, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        nseg, lines = oracle.iso_mef(nodes, faces, comp, val)
        assert f"Found {nseg} segments" in out.stdout and f"number of contours {nlines}" in out.stderr and len(lines) == nlines
        txt = open(tmp_path / "out.dat").read().split("\n")
        assert txt[0] == "VARIABLES = X Y Z height xy"
        want = []
        for ln in lines:
            want.append(f"ZONE ZONETYPE=FELINESEG DATAPACKING=POINT N={len(ln)} E={len(ln) - 1}")
            want += [" ".join("%g" % v for v in p) + " " for p in ln]
            want += [f"{c} {c + 1}" for c in range(1, len(ln))]
        assert txt[1:-1] == want and txt[-1] == ""
        for ln in lines:
            P = np.array(ln)
            assert np.allclose(P[0], P[-1], atol=0)                                  # closed
            assert np.abs(P[:, comp] - val).max() < 1e-12                            # on the iso value
            r = np.linalg.norm(P[:, :3], axis=1)
            assert r.max() <= 1.0 + 1e-12 and r.min() > 0.98                         # on the facets' edges, just inside the sphere
        if comp == 3 and nlines == 1:
            P = np.array(lines[0])
            length = np.linalg.norm(np.diff(P[:, :3], axis=0), axis=1).sum()
            assert abs(length / (2 * np.pi * np.sqrt(1 - (0.35 - 0.013) ** 2)) - 1) < 0.01
    bad = subprocess.run([os.path.join(BIN, "isoMEF3d.ex"), "infile=" + f, "isoComp=9", "isoVal=0"], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "isoComp" in bad.stderr


def test_cpp_check_iso_tool(tmp_path):
    """checkIso3d.ex (host only; checkIso.cpp:127-149): a consistently oriented tetrahedron passes; with one face
    flipped the reference's assertion still cannot fire (direction-blind comparator: quirk kept), strict=1 reports it"""
    _build_tools()
    nodes = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
    good = np.array([[1, 3, 2], [1, 2, 4], [2, 3, 4], [3, 1, 4]], np.int32)  # outward normals
    f = str(tmp_path / "tet.mef")
    _write_mef(f, nodes, good)
    for extra in ([], ["strict=1"]):
        out = subprocess.run([os.path.join(BIN, "checkIso3d.ex"), "isoFile=" + f] + extra, cwd=tmp_path, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert "Found 6 edges (nElts * 3 = 12)" in out.stdout and "All shared edges are consistently numbered." in out.stdout
    bad = good.copy()
    bad[2] = bad[2][::-1]
    _write_mef(f, nodes, bad)
    out = subprocess.run([os.path.join(BIN, "checkIso3d.ex"), "isoFile=" + f], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0 and "Found 6 edges" in out.stdout
    out = subprocess.run([os.path.join(BIN, "checkIso3d.ex"), "isoFile=" + f, "strict=1"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 2 and "traversed twice in the same direction" in out.stderr
    out = subprocess.run([os.path.join(BIN, "checkIso3d.ex")], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode != 0 and "isoFile" in out.stderr


def test_single_precision_plotfile_is_read_like_amrdata(tmp_path):
    """FABio::FAB_NATIVE_32 plotfiles (IEEE floats; what AmrLevel-based codes such as PeleC write by default): both readers
    widen to double on read, as AmrData does; the C++ one through template3d.ex, which writes doubles back"""
    _build_tools()
    p, H, mfs = _synth(tmp_path)
    p32 = str(tmp_path / "plt32")
    write_plotfile(p32, H, mfs, ["temp", "x_velocity", "density"], time=0.125, level_steps=[5, 5, 5], precision=32)
    assert b"(4, (4 3 2 1))" in open(os.path.join(p32, "Level_0", "Cell_D_00000"), "rb").read(200)
    r = read_plotfile(p32)
    for l in range(3):
        want = mfs[l].data.astype(np.float32).astype(np.float64)
        assert np.array_equal(r.mfs[l].data.view(np.int64), want.view(np.int64))
    out = subprocess.run([os.path.join(BIN, "template3d.ex"), "infile=" + p32, "is_per=1 1 0"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    back = read_plotfile(str(tmp_path / "plt32_temp"))
    for l in range(3):
        want = mfs[l].data.astype(np.float32).astype(np.float64)
        assert np.array_equal(back.mfs[l].data.view(np.int64), want.view(np.int64))
    # an unknown RealDescriptor is refused, not misread
    raw = open(os.path.join(p32, "Level_0", "Cell_D_00000"), "rb").read().replace(b"(4, (4 3 2 1))", b"(4, (1 2 3 4))")
    open(os.path.join(p32, "Level_0", "Cell_D_00000"), "wb").write(raw)
    out = subprocess.run([os.path.join(BIN, "template3d.ex"), "infile=" + p32], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode != 0 and "RealDescriptor" in out.stderr


# ------------------------------------------------------------------------------------ GPU tier
@pytest.mark.gpu
def test_grad_tool_end_to_end(tmp_path, oracle):
    """grad3d.ex infile=plt gradVar=temp is_per="1 1 0" Aux_Variables=density  ->  <root>_gt"""
    if not os.path.exists(os.path.join(BIN, "grad3d.ex")):
        _build_tools()
    p, H, mfs = _synth(tmp_path)
    out = subprocess.run([os.path.join(BIN, "grad3d.ex"), "infile=" + p, "gradVar=temp", "is_per=1 1 0", "Aux_Variables=density"],
                         cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    r = read_plotfile(str(tmp_path / "plt00005_gt"))
    assert r.names == ["temp", "density", "temp_gx", "temp_gy", "temp_gz", "||gradtemp||"]
    bc = oracle.bc_from_flags((1, 1, 0))
    st = []
    for l, lv in enumerate(H.levels):
        s = MultiFab(lv, 1, 1)
        for b in range(lv.nboxes):
            s.valid(b)[0] = mfs[l].valid(b)[0]
        st.append(s)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, st, 0, bc, og, 0)
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            v = r.mfs[l].valid(b)
            assert np.array_equal(v[0].view(np.int64), mfs[l].valid(b)[0].view(np.int64))
            assert np.array_equal(v[1].view(np.int64), mfs[l].valid(b)[2].view(np.int64))
            assert np.array_equal(np.ascontiguousarray(v[2:6]).view(np.int64), np.ascontiguousarray(og[l].valid(b)).view(np.int64))
    bad = subprocess.run([os.path.join(BIN, "grad3d.ex"), "infile=" + p, "gradVar=nope"], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "Cannot find nope" in bad.stderr


def _run(tool, args, cwd, timeout=None, env=None):
    if not os.path.exists(os.path.join(BIN, tool)):
        _build_tools()
    out = subprocess.run([os.path.join(BIN, tool)] + args, cwd=cwd, capture_output=True, text=True, timeout=timeout,
                         env=None if env is None else dict(os.environ, **env))
    assert out.returncode == 0, out.stderr + out.stdout
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [1, 0])
def test_curvature_tool_end_to_end(tmp_path, oracle, fused):
    p, H, mfs = _synth(tmp_path)
    _run("curvature3d.ex", ["infile=" + p, "progressName=temp", "is_per=1 1 0", "Aux_Variables=density", "threshold_prog=1",
                            "threshold_value=0.01", "fused=%d" % fused], tmp_path)
    r = read_plotfile(str(tmp_path / "plt00005_K"))
    assert r.names == ["temp", "density", "Progress", "SmoothedProgress", "MeanCurvature_temp", "FlameNormalX_temp", "FlameNormalY_temp",
                       "FlameNormalZ_temp", "GaussianCurvature_temp"]
    st = []
    for l, lv in enumerate(H.levels):
        s = MultiFab(lv, 1, 2)
        for b in range(lv.nboxes):
            s.valid(b)[0] = mfs[l].valid(b)[0]
        st.append(s)
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, st, 0, oracle.bc_from_flags((1, 1, 0)), oc, 0, MultiFab, threshold=0.01)
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            v, w = r.mfs[l].valid(b), oc[l].valid(b)
            same = lambda a, c: np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(c).view(np.int64))
            assert same(v[2], w[0]) and same(v[4], w[1]) and same(v[5:8], w[2:5])
            assert np.all(v[3] == 0.0) and np.all(v[8] == 0.0)  # quirk Q1: defined as 0.0


@pytest.mark.gpu
def test_filter_tool_end_to_end(tmp_path, oracle):
    p, H, mfs = _synth(tmp_path, nlev=2, base=16, box=16, per=(0, 0, 0))
    _run("filterPlt3d.ex", ["infile=" + p, "max_grid_size=8", "variables=temp density"], tmp_path)
    rsep = read_plotfile(str(tmp_path / "plt00005_filtered"))  # the separable default, compared below to 1e-12 * Linf
    _run("filterPlt3d.ex", ["infile=" + p, "max_grid_size=8", "variables=temp density", "exact_filter=1"], tmp_path)
    r = read_plotfile(str(tmp_path / "plt00005_filtered"))
    assert r.names == ["temp", "density"] and r.time == 0.125
    # oracle on the re-chopped BoxArray (BoxArray::maxSize(8))
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
    levels, ins = [], []
    for l, lv in enumerate(H.levels):
        ba = np.vstack([chop_box(bx[:3], bx[3:], 8) for bx in lv.boxes])
        nl = Level(ba, lv.domlo, lv.domhi, (0, 0, 0), lv.prob_lo, lv.prob_hi)
        assert np.array_equal(r.hier.levels[l].boxes, ba)
        s = MultiFab(nl, 2, 2)
        for b in range(nl.nboxes):
            for ob in range(lv.nboxes):  # locate the parent file box
                if np.all(ba[b, :3] >= lv.boxes[ob, :3]) and np.all(ba[b, 3:] <= lv.boxes[ob, 3:]):
                    o = ba[b, :3] - lv.boxes[ob, :3]
                    nz, ny, nx = nl.box_shape(b)
                    s.valid(b)[:] = mfs[l].valid(ob)[[0, 2], o[2]:o[2] + nz, o[1]:o[1] + ny, o[0]:o[0] + nx]
        levels.append(nl)
        ins.append(s)
    outs = [MultiFab(lv, 2, 0) for lv in levels]
    oracle.filter_pipeline(levels, [s.copy() for s in ins], outs, 2, base_fgr=2, interp_type=1)
    for l, lv in enumerate(levels):
        for b in range(lv.nboxes):
            assert np.array_equal(np.ascontiguousarray(r.mfs[l].valid(b)).view(np.int64), np.ascontiguousarray(outs[l].valid(b)).view(np.int64))
        for c in range(2):
            scale = max(float(np.abs(outs[l].valid(b)[c]).max()) for b in range(lv.nboxes))
            for b in range(lv.nboxes):
                assert float(np.abs(rsep.mfs[l].valid(b)[c] - outs[l].valid(b)[c]).max()) <= 1e-12 * scale
    # filter_type=4 (the 5-point approximation of the box filter; an odd base_fgr is fine there), and the types that stay refused
    _run("filterPlt3d.ex", ["infile=" + p, "max_grid_size=8", "variables=temp density", "filter_type=4", "base_fgr=3", "exact_filter=1"], tmp_path)
    r4 = read_plotfile(str(tmp_path / "plt00005_filtered"))
    outs4 = [MultiFab(lv, 2, 0) for lv in levels]
    oracle.filter_pipeline(levels, [s.copy() for s in ins], outs4, 2, base_fgr=3, interp_type=1, filter_type=4)
    for l, lv in enumerate(levels):
        for b in range(lv.nboxes):
            assert np.array_equal(np.ascontiguousarray(r4.mfs[l].valid(b)).view(np.int64), np.ascontiguousarray(outs4[l].valid(b)).view(np.int64))
            assert not np.array_equal(r4.mfs[l].valid(b), outs[l].valid(b))
    bad = subprocess.run([os.path.join(BIN, "filterPlt3d.ex"), "infile=" + p, "filter_type=5"], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "filter_type 5 is not available" in bad.stderr
    # filter_type=2 (Gaussian) is REFUSED unless asked for: PelePhysics' weights could not be checked.  With the opt-in the tool uses
    # the textbook kernel and says so; the comparison below is of two implementations of that SAME unverified formula (the library's
    # kernels against the oracle's loops) -- a consistency check of the filter kernels at wide stencils, not a parity reference.
    bad = subprocess.run([os.path.join(BIN, "filterPlt3d.ex"), "infile=" + p, "filter_type=2"], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "refused by default" in bad.stderr and "allow_unverified_gaussian=1" in bad.stderr
    g = _run("filterPlt3d.ex", ["infile=" + p, "max_grid_size=8", "variables=temp density", "filter_type=2", "exact_filter=1", "allow_unverified_gaussian=1"], tmp_path)
    assert "UNVERIFIED against PelePhysics" in g.stdout
    rg = read_plotfile(str(tmp_path / "plt00005_filtered"))
    insg = []
    for s_, lv in zip(ins, levels):  # ngrow 3 on level 0 (fgr 2), 5 on level 1 (fgr 4)
        m = MultiFab(lv, 2, 5)
        for b in range(lv.nboxes):
            m.valid(b)[:] = s_.valid(b)
        insg.append(m)
    outsg = [MultiFab(lv, 2, 0) for lv in levels]
    info = oracle.filter_pipeline(levels, insg, outsg, 2, base_fgr=2, interp_type=1, filter_type=2)
    assert info == [(2, 3), (4, 5)]
    for l, lv in enumerate(levels):
        for b in range(lv.nboxes):
            assert np.array_equal(np.ascontiguousarray(rg.mfs[l].valid(b)).view(np.int64), np.ascontiguousarray(outsg[l].valid(b)).view(np.int64)), f"gaussian level {l} box {b}"
    bad = subprocess.run([os.path.join(BIN, "filterPlt3d.ex"), "infile=" + p, "base_fgr=3"], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "even" in bad.stderr


@pytest.mark.gpu
def test_isosurface_tool_end_to_end(tmp_path, oracle):
    """node ids, node data and triangle connectivity of the MEF file identical to the oracle's"""
    p, H, mfs = _synth(tmp_path, nlev=3, base=16, box=8, per=(0, 0, 0))
    _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 2", "computeArea=1"], tmp_path)
    label, names, nodes, faces = read_mef(p + "_temp_1150.mef")
    assert label == "0.125" and names == ["X", "Y", "Z", "temp", "density"]
    fields = [MultiFab(lv, 3, 0, mfs[l].data.copy()) for l, lv in enumerate(H.levels)]
    onodes, oelts = oracle.isosurface_pipeline(H.levels, fields, [0, 2], 0, 1150.0, MultiFab)
    assert len(oelts) > 200
    assert np.array_equal(faces, oelts + 1), "connectivity (1-based) differs"
    assert np.array_equal(nodes.view(np.int64), onodes.view(np.int64)), "node data not bit-identical"
    # the run above built the node / element sets on the device (pa_iso_merge); the sequential host merge writes the same bytes
    host = subprocess.run([os.path.join(BIN, "isosurface3d.ex"), "infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 2", "outfile_base=" + str(tmp_path / "hostmerge")],
                          cwd=tmp_path, capture_output=True, text=True, env=dict(os.environ, PA_ISO_HOST_MERGE="1"))
    assert host.returncode == 0, host.stderr
    assert open(tmp_path / "hostmerge.mef", "rb").read() == open(p + "_temp_1150.mef", "rb").read()
    # ... and so does the round-4 form of the state (three stored coordinate components, FillBoundary + FillPatch on them, a ghost fill
    # per level: PA_ISO_XYZ=0); the default forms the coordinates from cell indices (pa_mc_hierarchy_xyz)
    stored = subprocess.run([os.path.join(BIN, "isosurface3d.ex"), "infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 2", "outfile_base=" + str(tmp_path / "stored")],
                            cwd=tmp_path, capture_output=True, text=True, env=dict(os.environ, PA_ISO_XYZ="0"))
    assert stored.returncode == 0, stored.stderr
    assert open(tmp_path / "stored.mef", "rb").read() == open(p + "_temp_1150.mef", "rb").read()
    # single-level surface: closed and consistently oriented (the invariant checkIso.cpp is after, checked for real)
    (tmp_path / "one").mkdir()
    p1, H1, _ = _synth(tmp_path / "one", nlev=1, base=32, box=16, per=(0, 0, 0))
    _run("isosurface3d.ex", ["infile=" + p1, "isoCompName=temp", "isoVal=1150", "comps=0"], tmp_path)
    out = _run("checkIso3d.ex", ["isoFile=" + p1 + "_temp_1150.mef", "strict=1"], tmp_path)
    assert "All shared edges are consistently numbered." in out.stdout


@pytest.mark.gpu
def test_tools_on_a_five_level_plotfile(tmp_path, oracle):
    """more levels than one batched launch of the library takes (PA_MAXB = 4: the all-levels launches go in chunks, marching cubes level
    by level): grad3d, curvature3d and isosurface3d against the oracle, bit for bit"""
    p, H, mfs = _synth(tmp_path, nlev=5, base=16, box=8, per=(1, 1, 0))
    same = lambda a, c: np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(c).view(np.int64))
    _run("grad3d.ex", ["infile=" + p, "gradVar=temp", "is_per=1 1 0"], tmp_path)
    r = read_plotfile(str(tmp_path / "plt00005_gt"))
    assert r.hier.nlev == 5
    st1, st2 = [], []
    for l, lv in enumerate(H.levels):
        for ng, dst in ((1, st1), (2, st2)):
            s = MultiFab(lv, 1, ng)
            for b in range(lv.nboxes):
                s.valid(b)[0] = mfs[l].valid(b)[0]
            dst.append(s)
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, st1, 0, oracle.bc_from_flags((1, 1, 0)), og, 0)
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            assert same(r.mfs[l].valid(b)[1:5], og[l].valid(b)), ("grad", l, b)
    _run("curvature3d.ex", ["infile=" + p, "progressName=temp", "is_per=1 1 0"], tmp_path)
    r = read_plotfile(str(tmp_path / "plt00005_K"))
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, st2, 0, oracle.bc_from_flags((1, 1, 0)), oc, 0, MultiFab)
    i0 = r.names.index("Progress")
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            v, w = r.mfs[l].valid(b), oc[l].valid(b)
            assert same(v[i0], w[0]) and same(v[i0 + 2], w[1]) and same(v[i0 + 3:i0 + 6], w[2:5]), ("curvature", l, b)
    _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 2", "is_per=1 1 0"], tmp_path)
    label, names, nodes, faces = read_mef(p + "_temp_1150.mef")
    fields = [MultiFab(lv, 3, 0, mfs[l].data.copy()) for l, lv in enumerate(H.levels)]
    onodes, oelts = oracle.isosurface_pipeline(H.levels, fields, [0, 2], 0, 1150.0, MultiFab)
    assert len(oelts) > 100 and np.array_equal(faces, oelts + 1) and np.array_equal(nodes.view(np.int64), onodes.view(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("per", [(1, 1, 0), (1, 1, 1)])
def test_isosurface_tool_periodic(tmp_path, oracle, per):
    """is_per != 0 (isosurface.cpp:1395, 1437, 1469, 1550-1560): periodic ghost fill of coordinates and fields, periodic
    images of the fine-covered mask, loop box over the periodically grown domain -- MEF identical to the oracle's.  The
    fine level touches the low x and the high z face of the domain and the surface crosses every face."""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box, field_trig
    zero, one = np.zeros(3), np.ones(3)
    l0 = Level(chop_box((0, 0, 0), (15, 15, 15), 8), (0, 0, 0), (15, 15, 15), np.asarray(per), zero, one)
    l1 = Level(chop_box((0, 8, 16), (15, 23, 31), 8), (0, 0, 0), (31, 31, 31), np.asarray(per), zero, one)
    H = Hierarchy([l0, l1], 2)
    mfs = make_states(H, 3, 0, field_trig, seed=5)
    p = str(tmp_path / "plt00007")
    write_plotfile(p, H, mfs, ["temp", "x_velocity", "density"], time=0.5, level_steps=[7, 7])
    _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1050", "comps=0 2", "is_per=" + " ".join(str(x) for x in per)], tmp_path)
    _, _, nodes, faces = read_mef(p + "_temp_1050.mef")
    fields = [MultiFab(lv, 3, 0, mfs[l].data.copy()) for l, lv in enumerate(H.levels)]
    onodes, oelts = oracle.isosurface_pipeline(H.levels, fields, [0, 2], 0, 1050.0, MultiFab)
    assert len(oelts) > 500
    span = onodes[oelts][:, :, 0].max(axis=1) - onodes[oelts][:, :, 0].min(axis=1)
    assert (span > 2.0 / 32).any(), "no stretched element behind the periodic x face: the quirk is not exercised"
    assert np.array_equal(faces, oelts + 1), "connectivity (1-based) differs"
    assert np.array_equal(nodes.view(np.int64), onodes.view(np.int64)), "node data not bit-identical"


@pytest.mark.gpu
def test_isosurface_tool_distance_function(tmp_path, oracle):
    """build_distance_function=1 (isosurface.cpp:1361-1381, 1595-1655, 1731-1748): the "distance" plotfile
    and the (unmasked) surface identical to the oracle's -- whose make_level_set3 is pinned to the reference
    build; nGrow = 1, 2, 4 on the three levels, so the element trimming of :1657-1682 is exercised too"""
    p, H, mfs = _synth(tmp_path, nlev=3, base=16, box=8, per=(0, 0, 0))
    out = _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 2", "build_distance_function=1", "outfile=" + str(tmp_path / "dist")],
               tmp_path)
    assert "dmax: 0.0625" in out.stdout
    fields = [MultiFab(lv, 3, 0, mfs[l].data.copy()) for l, lv in enumerate(H.levels)]
    onodes, oelts, odist = oracle.isosurface_pipeline(H.levels, fields, [0, 2], 0, 1150.0, MultiFab, build_distance=True)
    assert [d.ng for d in odist] == [1, 2, 4]
    r = read_plotfile(str(tmp_path / "dist"))
    assert r.names == ["distance"] and r.time == 0.0 and r.level_steps == [5, 5, 5]  # isosurface.cpp:1417,1747: the local `Real time = 0`
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            got, want = np.ascontiguousarray(r.mfs[l].valid(b)[0]), np.ascontiguousarray(odist[l].valid(b)[0])
            assert np.array_equal(got.view(np.int64), want.view(np.int64)), (l, b)
        v = r.mfs[l].valid_concat(0)
        assert np.abs(v).max() == 0.0625 and (np.abs(v) < 0.0625).any()
    label, names, nodes, faces = read_mef(p + "_temp_1150.mef")
    assert np.array_equal(faces, oelts + 1) and np.array_equal(nodes.view(np.int64), onodes.view(np.int64))
    # XDMF output (isosurface.cpp:2135-2229): same surface, 0-based connectivity, xyz then one array per component
    _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 2", "build_distance_function=1", "outfile=" + str(tmp_path / "d2"),
                            "surfFormat=XDMF", "outfile_base=" + str(tmp_path / "xs")], tmp_path)
    xmf = open(str(tmp_path / "xs.xmf")).read()
    raw = open(str(tmp_path / "xs.mesh"), "rb").read()
    nn, ne = len(onodes), len(oelts)
    assert f'NumberOfElements="{ne}"' in xmf and f'Seek="{12 * ne}" Dimensions="{3 * nn}"' in xmf and 'Value="1150"' in xmf and 'Attribute Name="density"' in xmf
    assert len(raw) == 12 * ne + 8 * nn * 5
    assert np.array_equal(np.frombuffer(raw[:12 * ne], "<i4").reshape(ne, 3), oelts)
    xyz = np.frombuffer(raw[12 * ne:12 * ne + 24 * nn], "<f8").reshape(nn, 3)
    assert np.array_equal(xyz.view(np.int64), np.ascontiguousarray(onodes[:, :3]).view(np.int64))
    for c in range(2):
        a = np.frombuffer(raw[12 * ne + 24 * nn + 8 * nn * c:12 * ne + 24 * nn + 8 * nn * (c + 1)], "<f8")
        assert np.array_equal(a.view(np.int64), np.ascontiguousarray(onodes[:, 3 + c]).view(np.int64))
    # plain surface with two ghost layers: trimming leaves the one-ghost-layer surface
    _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 2", "nGrow=2", "outfile_base=" + str(tmp_path / "g2")], tmp_path)
    _, _, nodes2, faces2 = read_mef(str(tmp_path / "g2.mef"))
    on2, oe2 = oracle.isosurface_pipeline(H.levels, fields, [0, 2], 0, 1150.0, MultiFab, ngrow=2)
    assert np.array_equal(faces2, oe2 + 1) and np.array_equal(nodes2.view(np.int64), on2.view(np.int64))


@pytest.mark.gpu
def test_curvature_tool_options(tmp_path, oracle):
    """do_gaussCurv=1 do_strain=1 getStrainTensor=1 do_velnormal=1: names (curvature.cpp:796-831) and values"""
    p, H, mfs = _synth(tmp_path, nlev=2, ncomp=4, names=("temp", "x_velocity", "y_velocity", "z_velocity"))
    _run("curvature3d.ex", ["infile=" + p, "is_per=1 1 0", "do_gaussCurv=1", "do_strain=1", "getStrainTensor=1", "do_velnormal=1"], tmp_path)
    r = read_plotfile(str(tmp_path / "plt00005_K"))
    assert r.names[:4] == ["temp", "x_velocity", "y_velocity", "z_velocity"]
    assert r.names[4:12] == ["Progress", "SmoothedProgress", "MeanCurvature_temp", "FlameNormalX_temp", "FlameNormalY_temp", "FlameNormalZ_temp",
                             "GaussianCurvature_temp", "StrainRate_temp"]
    assert r.names[12:21] == ["ROST_dUxdx", "ROST_dUxdy", "ROST_dUxdz", "ROST_dUydx", "ROST_dUydy", "ROST_dUydz", "ROST_dUzdx", "ROST_dUzdy", "ROST_dUzdz"]
    assert r.names[21] == "VelFlameNormal" and len(r.names) == 22
    st = [MultiFab(lv, 4, 2) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            st[l].valid(b)[:] = mfs[l].valid(b)
    oc = [MultiFab(lv, 17, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, st, 0, oracle.bc_from_flags((1, 1, 0)), oc, 0, MultiFab, do_gauss=True, vel_comp=1, do_strain=True,
                              do_velnormal=True, strain_tensor=True)
    same = lambda a, c: np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(c).view(np.int64))
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            v, w = r.mfs[l].valid(b), oc[l].valid(b)
            assert same(v[6], w[1]) and same(v[7:10], w[2:5]) and same(v[10], w[5]) and same(v[11], w[6]) and same(v[12:21], w[8:17]) and same(v[21], w[7])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [1e-3, 2.5e-2])
def test_curvature_tool_do_smooth(tmp_path, oracle, dt):
    """do_smooth=1 smoothing_time=<dt> (curvature.cpp:328-406): SmoothedProgress from the composite implicit solve
    (to the oracle's solve within 1e-10), Progress untouched, curvature computed from the smoothed field.  dt = 2.5e-2 is
    dt / dx^2 = 100 on the finest level: the tool's solve then runs with the multigrid preconditioner (pa_smooth.hip), the
    oracle's stays the plain iteration"""
    p, H, mfs = _synth(tmp_path, nlev=3, ncomp=1, names=("temp",))
    out = _run("curvature3d.ex", ["infile=" + p, "is_per=1 1 0", "do_smooth=1", f"smoothing_time={dt}"], tmp_path)
    its = int(out.stdout.split("Smoothing solve")[1].split(":")[1].split("iterations")[0])
    assert 0 < its < (40 if dt > 1e-2 else 100), out.stdout
    r = read_plotfile(str(tmp_path / "plt00005_K"))
    assert r.names == ["temp", "Progress", "SmoothedProgress", "MeanCurvature_temp", "FlameNormalX_temp", "FlameNormalY_temp", "FlameNormalZ_temp",
                       "GaussianCurvature_temp"]
    st = [MultiFab(lv, 1, 2) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            st[l].valid(b)[:] = mfs[l].valid(b)
    oc = [MultiFab(lv, 18, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, st, 0, oracle.bc_from_flags((1, 1, 0)), oc, 0, MultiFab, do_smooth=True, smoothing_time=dt, smooth_tol=1e-13, smooth_maxiter=3000)
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            v, w = r.mfs[l].valid(b), oc[l].valid(b)
            assert np.array_equal(np.ascontiguousarray(v[1]).view(np.int64), np.ascontiguousarray(w[0]).view(np.int64))
            assert np.abs(v[2] - w[17]).max() <= 1e-10 and np.abs(v[2] - v[1]).max() > 1e-4
    # downstream of the solve: the tool's curvature of the ORACLE's smoothed field (a plotfile whose "temp" is that field, range
    # [0, 1] given: (c - 0.0) * 1.0 = c) is the oracle's curvature / normals bit for bit in every cell -- what separates the run
    # above from the oracle is the solve's tolerance alone, amplified by n = G / |G| (bounds: tests/test_gpu_smooth.py)
    sm = [MultiFab(lv, 1, 0) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            sm[l].valid(b)[0] = oc[l].valid(b)[17]
    ps = str(tmp_path / "plt00009")
    write_plotfile(ps, H, sm, ["temp"], time=0.125, level_steps=[5] * H.nlev)
    _run("curvature3d.ex", ["infile=" + ps, "is_per=1 1 0", "useFileMinMax=0", "progMin=0", "progMax=1"], tmp_path)
    rs = read_plotfile(str(tmp_path / "plt00009_K"))
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            v, w = rs.mfs[l].valid(b), oc[l].valid(b)
            for vc, wc in ((3, 1), (4, 2), (5, 3), (6, 4)):
                assert np.array_equal(np.ascontiguousarray(v[vc]).view(np.int64), np.ascontiguousarray(w[wc]).view(np.int64)), (l, b, vc)


@pytest.mark.gpu
def test_partstream_tool_end_to_end(tmp_path, oracle):
    """partStream3d.ex (partStream.cpp / StreamPC.cpp): seeds from the isosurface tool's MEF nodes and from a rake;
    the Tecplot zones (seed order, forward then backward line) equal the oracle's lines to the 6 digits written"""
    p, H, mfs = _synth(tmp_path, nlev=3, ncomp=4, names=("temp", "x_velocity", "y_velocity", "z_velocity"), per=(0, 0, 0))
    _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0", "outfile_base=" + str(tmp_path / "surf")], tmp_path)
    _, _, nodes, _ = read_mef(str(tmp_path / "surf.mef"))
    fields = [MultiFab(lv, 4, 0, mfs[l].data.copy()) for l, lv in enumerate(H.levels)]
    v = oracle.stream_field(H.levels, fields, (1, 2, 3), MultiFab, ngrow=3)
    dt = 0.2 * float(H.levels[-1].dx[0])

    def zones(path):
        txt = open(path).read().split("\n")
        assert txt[0].strip() == "VARIABLES = X Y Z"
        out, cur = [], None
        for ln in txt[1:]:
            if ln.startswith("ZONE"):
                cur = []
                out.append(cur)
            elif ln.strip():
                cur.append([float(t) for t in ln.split()])
        return np.array(out)

    for args, seeds in ((["isoFile=" + str(tmp_path / "surf.mef")], nodes[:, :3]),
                        (["seedRakeNum=5", "seedRakeL=0.3 0.4 0.45", "seedRakeR=0.7 0.6 0.55"],
                         np.array([[0.3 + (i / 4.0) * (0.7 - 0.3), 0.4 + (i / 4.0) * (0.6 - 0.4), 0.45 + (i / 4.0) * (0.55 - 0.45)] for i in range(5)]))):
        _run("partStream3d.ex", ["infile=" + p, "Nsteps=30", "hRK=0.2"] + args, tmp_path)
        z = zones(str(tmp_path / "tec.dat" / "str_00000.dat"))
        want, _ = oracle.stream_trace(H.levels, v, seeds, 30, dt)
        assert z.shape == want.shape and len(seeds) >= 5
        assert np.abs(z - want).max() <= 5e-6 * max(1.0, np.abs(want).max())  # ostream default precision: 6 significant digits
        # ngpus=<n>: the lines dealt to n ranks (here sharing the GPU), the redistribution flag of every step reduced over the
        # ranks: the file is byte-identical (3 ranks: uneven shares; 7 ranks > 5 rake seeds: ranks without a line)
        ref_bytes = open(tmp_path / "tec.dat" / "str_00000.dat", "rb").read()
        for n in (3, 7):
            out = _run("partStream3d.ex", ["infile=" + p, "Nsteps=30", "hRK=0.2", f"ngpus={n}", "gpu_share=1"] + args, tmp_path, timeout=300)
            assert f"Lines dealt to {n} GPUs" in out.stdout
            assert open(tmp_path / "tec.dat" / "str_00000.dat", "rb").read() == ref_bytes, f"partStream ngpus={n} differs from the single-GPU file"
    bad = subprocess.run([os.path.join(BIN, "partStream3d.ex"), "infile=" + p], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "Assertion" in bad.stderr


@pytest.mark.gpu
def test_partstream_one_seed_per_cell(tmp_path, oracle):
    """oneSeedPerCell (partStream.cpp:21-61): a seed at the centre of every cell, not covered by the finer level, of the grids
    that contain cell (0, 50, 107) of their level -- here one 8 x 32 x 32 grid of level 0 (a finer level covers part of
    it) and one grid of level 1.  Lines equal the oracle's for the seeds generated here by the same rule."""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box, fill_analytic
    plo, phi = np.zeros(3), np.asarray((0.125, 1.0, 2.0))
    l0 = Level(chop_box((0, 0, 0), (7, 63, 127), 32), np.zeros(3, dtype=np.int64), np.asarray((7, 63, 127)), np.zeros(3, dtype=np.int64), plo, phi)
    # level 1 refines coarse cells [0..7] x [40..55] x [100..111]  ->  fine [0..15] x [80..111] x [200..223] (contains (0, 50, 107)? no: j = 50 < 80)
    l1 = Level(chop_box((0, 80, 200), (15, 111, 223), 16), np.zeros(3, dtype=np.int64), np.asarray((15, 127, 255)), np.zeros(3, dtype=np.int64), plo, phi)
    H = Hierarchy([l0, l1], 2)
    mfs = []
    for lv in H.levels:
        m = MultiFab(lv, 3, 0)
        fill_analytic(m, 0, lambda x, y, z: 0.3 + 0.0 * x + 0.2 * np.sin(3 * y))
        fill_analytic(m, 1, lambda x, y, z: 0.1 * np.cos(2 * z) + 0.0 * x)
        fill_analytic(m, 2, lambda x, y, z: -0.2 + 0.1 * x + 0.05 * y)
        mfs.append(m)
    p = str(tmp_path / "pltv")
    write_plotfile(p, H, mfs, ["x_velocity", "y_velocity", "z_velocity"], time=0.0, level_steps=[0, 0])
    seeds = []
    for l, lv in enumerate(H.levels):
        dx = lv.dx
        for b in range(lv.nboxes):
            lo, hi = lv.boxes[b, :3], lv.boxes[b, 3:]
            if not (lo[0] <= 0 <= hi[0] and lo[1] <= 50 <= hi[1] and lo[2] <= 107 <= hi[2]):
                continue
            for k in range(lo[2], hi[2] + 1):
                for j in range(lo[1], hi[1] + 1):
                    for i in range(lo[0], hi[0] + 1):
                        if l == 0 and 0 <= i <= 7 and 40 <= j <= 55 and 100 <= k <= 111:
                            continue  # covered by level 1
                        seeds.append([(i + 0.5) * dx[0], (j + 0.5) * dx[1], (k + 0.5) * dx[2]])
    seeds = np.asarray(seeds)
    assert len(seeds) == 8 * 32 * 32 - 8 * 16 * 12  # only the level-0 grid [0..7] x [32..63] x [96..127] holds the tagged cell
    _run("partStream3d.ex", ["infile=" + p, "oneSeedPerCell=1", "Nsteps=4", "hRK=0.25", "nGrow=2"], tmp_path)
    txt = open(tmp_path / "tec.dat" / "str_00000.dat").read().split("\n")
    z = np.array([[float(t) for t in ln.split()] for ln in txt[1:] if ln.strip() and not ln.startswith("ZONE")]).reshape(-1, 4, 3)
    v = oracle.stream_field(H.levels, [MultiFab(lv, 3, 0, m.data.copy()) for lv, m in zip(H.levels, mfs)], (0, 1, 2), MultiFab, ngrow=2)
    want, _ = oracle.stream_trace(H.levels, v, seeds, 4, 0.25 * float(H.levels[-1].dx[0]))
    assert z.shape == want.shape
    assert np.abs(z - want).max() <= 5e-6 * max(1.0, np.abs(want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("per", [(0, 0), (1, 0)])
def test_isosurface2d_tool_end_to_end(tmp_path, oracle, per):
    """isosurface2d.ex (the AMREX_SPACEDIM == 2 build: Segmentise, two nodes per element, MakeCLines + "Integral:"):
    2-D plotfile -> MEF identical to the Python restatement (node ids, node data bit for bit, sorted segments), the same
    number of contour lines and the same line integrals"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
    per3 = np.array([per[0], per[1], 0])
    l0 = Level(chop_box((0, 0, 0), (31, 31, 0), 8), (0, 0, 0), (31, 31, 0), per3, np.zeros(3), np.ones(3))
    l1 = Level(chop_box((16, 16, 0), (47, 47, 0), 8), (0, 0, 0), (63, 63, 0), per3, np.zeros(3), np.ones(3))
    H = Hierarchy([l0, l1], 2)

    def fn(x, y, z, c):
        r = np.sqrt((x - 0.5) ** 2 + ((y - 0.5) / 0.8) ** 2) + 0 * z
        base = 300.0 + 850.0 * (1.0 + np.tanh((r - 0.27 - 0.03 * np.sin(5 * np.arctan2(y - 0.5, x - 0.5))) / 0.06))
        return base + 400.0 * np.sin(2 * np.pi * x) * (c == 0) if per[0] else (1.0 + 0.2 * c) * base

    mfs = make_states(H, 2, 0, fn, seed=9)
    p = str(tmp_path / "plt2d")
    write_plotfile(p, H, mfs, ["temp", "density"], time=0.25, level_steps=[3, 3], dim=2)
    assert open(os.path.join(p, "Header")).read().split("\n")[4] == "2"
    out = _run("isosurface2d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 1", "is_per=%d %d" % per], tmp_path)
    label, names, nodes, faces = read_mef(p + "_temp_1150.mef")
    assert label == "0.25" and names == ["X", "Y", "temp", "density"] and faces.shape[1] == 2
    fields = [MultiFab(lv, 2, 0, mfs[l].data.copy()) for l, lv in enumerate(H.levels)]
    onodes, oelts = oracle.isosurface2d_pipeline(H.levels, fields, [0, 1], 0, 1150.0, MultiFab)
    assert len(oelts) > 100
    assert np.array_equal(faces, oelts + 1), "segments (1-based) differ"
    assert np.array_equal(nodes.view(np.int64), onodes.view(np.int64)), "node data not bit-identical"
    lines = oracle.make_clines(oelts)
    assert "number of contour lines: %d" % len(lines) in out.stderr
    got = [[float(t) for t in ln.split()[1:]] for ln in out.stdout.splitlines() if ln.startswith("Integral:")]
    assert len(got) == len(lines)
    for g, ln in zip(got, lines):
        integ = np.zeros(2)
        for a, b in ln:
            p0, p1 = onodes[a], onodes[b]
            length = np.sqrt((p1[0] - p0[0]) ** 2 + (p1[1] - p0[1]) ** 2)
            nrm = np.array([(p0[1] - p1[1]) / length, (p1[0] - p0[0]) / length]) if length > 0 else np.zeros(2)
            integ += nrm * 0.5 * (p0[2] + p1[2]) * length
        assert np.allclose(g, integ, rtol=2e-5, atol=1e-9), (g, integ)
    # XDMF (Polyline / XY): 0-based segments, then (x, y) per node, then one array per component
    _run("isosurface2d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 1", "is_per=%d %d" % per, "surfFormat=XDMF",
                             "outfile_base=" + str(tmp_path / "x2")], tmp_path)
    xmf, raw = open(str(tmp_path / "x2.xmf")).read(), open(str(tmp_path / "x2.mesh"), "rb").read()
    nn, ne = len(onodes), len(oelts)
    assert f'TopologyType="Polyline" NodesPerElement="2" NumberOfElements="{ne}"' in xmf and 'GeometryType="XY"' in xmf and f'Seek="{8 * ne}" Dimensions="{2 * nn}"' in xmf
    assert len(raw) == 8 * ne + 8 * nn * 4
    assert np.array_equal(np.frombuffer(raw[:8 * ne], "<i4").reshape(ne, 2), oelts)
    xy = np.frombuffer(raw[8 * ne:8 * ne + 16 * nn], "<f8").reshape(nn, 2)
    assert np.array_equal(xy.view(np.int64), np.ascontiguousarray(onodes[:, :2]).view(np.int64))
    # a 3-D tool refuses the 2-D plotfile, the 2-D tool a 3-D one; the distance function aborts as in the reference
    bad = subprocess.run([os.path.join(BIN, "isosurface3d.ex"), "infile=" + p, "isoCompName=temp"], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "3-D" in bad.stderr
    bad = subprocess.run([os.path.join(BIN, "isosurface2d.ex"), "infile=" + p, "isoCompName=temp", "build_distance_function=1"], cwd=tmp_path,
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "not worked out for 2D" in bad.stderr


def _hier2d(per):
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
    per3 = np.array([per[0], per[1], 0])
    l0 = Level(chop_box((0, 0, 0), (31, 31, 0), 16), (0, 0, 0), (31, 31, 0), per3, np.zeros(3), np.ones(3))
    l1 = Level(chop_box((16, 16, 0), (47, 47, 0), 16), (0, 0, 0), (63, 63, 0), per3, np.zeros(3), np.ones(3))
    return Hierarchy([l0, l1], 2)


def _flame2d(x, y, z, c):
    r = np.sqrt((x - 0.5) ** 2 + ((y - 0.5) / 0.8) ** 2) + 0 * z
    return (1.0 + 0.2 * c) * (300.0 + 850.0 * (1.0 + np.tanh((r - 0.27 - 0.03 * np.sin(5 * np.arctan2(y - 0.5, x - 0.5))) / 0.06)))


def test_python_plotfile_roundtrip_2d(tmp_path):
    H = _hier2d((1, 0))
    mfs = make_states(H, 2, 0, _flame2d, seed=3)
    p = str(tmp_path / "p2")
    write_plotfile(p, H, mfs, ["temp", "density"], time=0.5, level_steps=[1, 1], dim=2)
    assert open(os.path.join(p, "Level_1", "Cell_H")).read().split("\n")[5] == "((16,16) (31,31) (0,0))"
    r = read_plotfile(p, is_per=(1, 0, 0))
    assert r.names == ["temp", "density"] and r.hier.nlev == 2
    for l in range(2):
        assert np.array_equal(r.hier.levels[l].boxes, H.levels[l].boxes)
        assert np.array_equal(r.mfs[l].data.view(np.int64), mfs[l].data.view(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("per", [(1, 1), (1, 0)])
def test_grad2d_and_curvature2d_tools(tmp_path, oracle, per):
    """grad2d.ex / curvature2d.ex (the AMREX_SPACEDIM == 2 builds): 2-D plotfile in, 2-D plotfile out; values identical to
    the oracle run on the same hierarchy stored as one plane of cells with a Neumann wall in z -- its divergence taken over
    x and y only and not halved -- and, for the interior of the periodic single-level case, to a direct numpy 2-D stencil"""
    from peleanalysis_amd import capi
    H = _hier2d(per)
    mfs = make_states(H, 2, 0, _flame2d, seed=4)
    p = str(tmp_path / "plt2")
    write_plotfile(p, H, mfs, ["temp", "density"], time=0.5, level_steps=[1, 1], dim=2)
    per3 = (per[0], per[1], 0)
    bc = capi.bc_from_flags(per3, (0, 0, 0))
    _run("grad2d.ex", ["infile=" + p, "gradVar=temp", "Aux_Variables=density", "is_per=%d %d" % per], tmp_path)
    r = read_plotfile(str(tmp_path / "plt2_gt"))
    assert r.names == ["temp", "density", "temp_gx", "temp_gy", "||gradtemp||"] and r.time == 0.0
    assert open(str(tmp_path / "plt2_gt" / "Header")).read().split("\n")[7] == "2"
    ost = [MultiFab(lv, 1, 1) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            ost[l].valid(b)[0] = mfs[l].valid(b)[0]
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oracle.grad_pipeline(H.levels, ost, 0, bc, og, 0, multipass=True)
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            got, want = r.mfs[l].valid(b), og[l].valid(b)
            assert np.array_equal(got[0].view(np.int64), mfs[l].valid(b)[0].view(np.int64))
            assert np.array_equal(got[1].view(np.int64), mfs[l].valid(b)[1].view(np.int64))
            for gc, wc in ((2, 0), (3, 1), (4, 3)):
                assert np.array_equal(np.ascontiguousarray(got[gc]).view(np.int64), np.ascontiguousarray(want[wc]).view(np.int64)), (l, b, gc)
            assert (want[2] == 0.0).all()  # the z derivative of the plane is an exact zero
    # direct 2-D stencil, level 0 (covers the whole periodic domain), interior of the x direction everywhere
    g0 = np.zeros((32, 32))
    for b in range(H.levels[0].nboxes):
        bx = H.levels[0].boxes[b]
        g0[bx[1]:bx[4] + 1, bx[0]:bx[3] + 1] = mfs[0].valid(b)[0, 0]
    dxinv = 1.0 / (1.0 / 32.0)
    fl = -(dxinv * (g0[:, 1:-1] - g0[:, :-2]))
    fh = -(dxinv * (g0[:, 2:] - g0[:, 1:-1]))
    gx = -(0.5 * (fl + fh))
    got0 = np.zeros((32, 32))
    for b in range(H.levels[0].nboxes):
        bx = H.levels[0].boxes[b]
        got0[bx[1]:bx[4] + 1, bx[0]:bx[3] + 1] = r.mfs[0].valid(b)[2, 0]
    assert np.array_equal(got0[:, 1:-1].view(np.int64), gx.view(np.int64))

    _run("curvature2d.ex", ["infile=" + p, "progressName=temp", "Aux_Variables=density", "is_per=%d %d" % per, "threshold_prog=1", "threshold_value=0.01"], tmp_path)
    k = read_plotfile(str(tmp_path / "plt2_K"))
    assert k.names == ["temp", "density", "Progress", "SmoothedProgress", "MeanCurvature_temp", "FlameNormalX_temp", "FlameNormalY_temp"]
    ost = [MultiFab(lv, 1, 2) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            ost[l].valid(b)[0] = mfs[l].valid(b)[0]
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, ost, 0, bc, oc, 0, MultiFab, threshold=0.01, spacedim=2)
    nz = 0
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            got, want = k.mfs[l].valid(b), oc[l].valid(b)
            for gc, wc in ((2, 0), (4, 1), (5, 2), (6, 3)):
                assert np.array_equal(np.ascontiguousarray(got[gc]).view(np.int64), np.ascontiguousarray(want[wc]).view(np.int64)), (l, b, gc)
            nz += int((got[4] != 0).sum())
    assert nz > 200
    # do_smooth in the 2-D build: SmoothedProgress from the composite solve on the planes, curvature from the smoothed field
    _run("curvature2d.ex", ["infile=" + p, "progressName=temp", "is_per=%d %d" % per, "do_smooth=1", "smoothing_time=4e-4", "outfile=" + str(tmp_path / "plt2_KS")], tmp_path)
    ks = read_plotfile(str(tmp_path / "plt2_KS"))
    assert ks.names == ["temp", "Progress", "SmoothedProgress", "MeanCurvature_temp", "FlameNormalX_temp", "FlameNormalY_temp"]
    ocs = [MultiFab(lv, 18, 0) for lv in H.levels]  # comp 17 = SmoothedProgress
    oracle.curvature_pipeline(H.levels, [o.copy() for o in ost], 0, bc, ocs, 0, MultiFab, do_smooth=True, smoothing_time=4e-4, smooth_tol=1e-14, spacedim=2)
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            got, want = ks.mfs[l].valid(b), ocs[l].valid(b)
            assert np.array_equal(np.ascontiguousarray(got[1]).view(np.int64), np.ascontiguousarray(want[0]).view(np.int64))
            assert np.abs(got[2] - want[17]).max() <= 1e-12 and not np.array_equal(got[2], got[1])
    # downstream: the 2-D tool's curvature of the ORACLE's smoothed field is the oracle's, bit for bit (see test_curvature_tool_do_smooth)
    sm2 = [MultiFab(lv, 1, 0) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            sm2[l].valid(b)[0] = ocs[l].valid(b)[17]
    ps2 = str(tmp_path / "plt2_sm")
    write_plotfile(ps2, H, sm2, ["temp"], time=0.5, level_steps=[1] * H.nlev, dim=2)
    _run("curvature2d.ex", ["infile=" + ps2, "progressName=temp", "is_per=%d %d" % per, "useFileMinMax=0", "progMin=0", "progMax=1"], tmp_path)
    kt = read_plotfile(ps2 + "_K")
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            got, want = kt.mfs[l].valid(b), ocs[l].valid(b)
            for gc, wc in ((3, 1), (4, 2), (5, 3)):
                assert np.array_equal(np.ascontiguousarray(got[gc]).view(np.int64), np.ascontiguousarray(want[wc]).view(np.int64)), (l, b, gc)
    bad = subprocess.run([os.path.join(BIN, "curvature2d.ex"), "infile=" + p, "progressName=temp", "do_gaussCurv=1"], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "2-D build" in bad.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("per,interp", [((1, 0), 1), ((0, 0), 0)])
def test_filterplt2d_tool(tmp_path, oracle, per, interp):
    """filterPlt2d.ex (the AMREX_SPACEDIM == 2 build): 9 taps on level 0, 25 on level 1; identical to the oracle's ghost
    fill on the one-plane hierarchy followed by a numpy restatement of the 2-D tap loop"""
    H = _hier2d(per)
    mfs = make_states(H, 2, 0, _flame2d, seed=6)
    p = str(tmp_path / "pf2")
    write_plotfile(p, H, mfs, ["temp", "density"], time=0.75, level_steps=[2, 2], dim=2)
    _run("filterPlt2d.ex", ["infile=" + p, "base_fgr=2", "max_grid_size=16", "interp_type=%d" % interp, "is_per=%d %d" % per], tmp_path)
    r = read_plotfile(str(tmp_path / "pf2_filtered"))
    assert r.names == ["temp", "density"] and r.time == 0.75
    ins = [MultiFab(lv, 2, 2, fill=0.0) for lv in H.levels]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            ins[l].valid(b)[:] = mfs[l].valid(b)
    outs = [MultiFab(lv, 2, 0) for lv in H.levels]
    info = oracle.filter_pipeline(H.levels, ins, outs, 2, base_fgr=2, interp_type=interp, spacedim=2)
    assert info == [(2, 1), (4, 2)]
    for l, lv in enumerate(H.levels):
        assert np.array_equal(r.hier.levels[l].boxes, lv.boxes)
        for b in range(lv.nboxes):
            assert np.array_equal(np.ascontiguousarray(r.mfs[l].valid(b)).view(np.int64), np.ascontiguousarray(outs[l].valid(b)).view(np.int64)), (l, b)
    # a constant stays a constant exactly (dyadic box weights)
    cm = make_states(H, 1, 0, lambda x, y, z, c: 7.0 + 0 * x + 0 * y + 0 * z)
    pc = str(tmp_path / "pc2")
    write_plotfile(pc, H, cm, ["c"], dim=2)
    _run("filterPlt2d.ex", ["infile=" + pc, "is_per=%d %d" % per], tmp_path)
    rc = read_plotfile(str(tmp_path / "pc2_filtered"))
    assert all((rc.mfs[l].data == 7.0).all() for l in range(2))


@pytest.mark.gpu
def test_curvature2d_tool_options(tmp_path, oracle):
    """curvature2d.ex do_strain=1 getStrainTensor=1 do_velnormal=1: names (no Gaussian curvature, 2 x 2 tensor) and values
    identical to the oracle on the one-plane hierarchy with a zero third velocity component"""
    from peleanalysis_amd import capi
    per = (1, 0)
    H = _hier2d(per)

    def fn(x, y, z, c):
        if c == 0:
            return _flame2d(x, y, z, 0)
        return (np.sin(2 * np.pi * x) * (1 + y) if c == 1 else np.cos(2 * np.pi * x) * y * y) + 0 * z

    mfs = make_states(H, 3, 0, fn, seed=12)
    p = str(tmp_path / "pv2")
    write_plotfile(p, H, mfs, ["temp", "x_velocity", "y_velocity"], time=0.5, level_steps=[1, 1], dim=2)
    _run("curvature2d.ex", ["infile=" + p, "progressName=temp", "is_per=1 0", "do_strain=1", "getStrainTensor=1", "do_velnormal=1", "threshold_prog=1",
                            "threshold_value=0.02"], tmp_path)
    k = read_plotfile(str(tmp_path / "pv2_K"))
    assert k.names == ["temp", "x_velocity", "y_velocity", "Progress", "SmoothedProgress", "MeanCurvature_temp", "FlameNormalX_temp", "FlameNormalY_temp",
                       "StrainRate_temp", "ROST_dUxdx", "ROST_dUxdy", "ROST_dUydx", "ROST_dUydy", "VelFlameNormal"]
    bc = capi.bc_from_flags((1, 0, 0), (0, 0, 0))
    ost = [MultiFab(lv, 4, 2) for lv in H.levels]  # [temp, u, v, 0]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            ost[l].valid(b)[0:3] = mfs[l].valid(b)
            ost[l].valid(b)[3] = 0.0
    oc = [MultiFab(lv, 17, 0) for lv in H.levels]
    oracle.curvature_pipeline(H.levels, ost, 0, bc, oc, 0, MultiFab, threshold=0.02, spacedim=2, vel_comp=1, do_strain=True, strain_tensor=True,
                              do_velnormal=True)
    pairs = [(3, 0), (5, 1), (6, 2), (7, 3), (8, 6), (9, 8), (10, 9), (11, 11), (12, 12), (13, 7)]
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            got, want = k.mfs[l].valid(b), oc[l].valid(b)
            for gc, wc in pairs:
                assert np.array_equal(np.ascontiguousarray(got[gc]).view(np.int64), np.ascontiguousarray(want[wc]).view(np.int64)), (l, b, gc, wc)
            assert (want[10] == 0.0).all() and (want[13] == 0.0).all() and (want[14] == 0.0).all()  # d/dz and dw/d. vanish exactly
    assert any((k.mfs[1].valid(b)[8] != 0).any() for b in range(H.levels[1].nboxes))


@pytest.mark.gpu
def test_tool_options_level_limits_ranges_inputs_file(tmp_path, oracle):
    """keys the other tests leave at their defaults: finestLevel / max_filter_level (fewer levels than the file holds),
    useFileMinMax=0 with progMin / progMax, same_fgr_all_levels + interp_type=0 + base_fgr=4, sComp / nComp, and the
    `exe inputs_file key=value` form (later definitions win)"""
    from peleanalysis_amd import capi
    p, H, mfs = _synth(tmp_path, nlev=3, base=16, box=8, per=(1, 1, 0))
    H2 = type(H)(H.levels[:2], 2)
    bc = capi.bc_from_flags((1, 1, 0), (0, 0, 0))
    # grad: finestLevel=1 from an inputs file, overridden outfile on the command line
    inp = tmp_path / "inputs.grad"
    inp.write_text("# sample inputs (Src/InputsSamples/inputs.grad)\ninfile = %s\ngradVar = temp\nfinestLevel = 1\nis_per = 1 1 0\noutfile = ignored\n" % p)
    _run("grad3d.ex", [str(inp), "outfile=" + str(tmp_path / "g1")], tmp_path)
    r = read_plotfile(str(tmp_path / "g1"))
    assert r.hier.nlev == 2 and r.names == ["temp", "temp_gx", "temp_gy", "temp_gz", "||gradtemp||"]
    ost = [MultiFab(lv, 1, 1) for lv in H2.levels]
    for l, lv in enumerate(H2.levels):
        for b in range(lv.nboxes):
            ost[l].valid(b)[0] = mfs[l].valid(b)[0]
    og = [MultiFab(lv, 4, 0) for lv in H2.levels]
    oracle.grad_pipeline(H2.levels, ost, 0, bc, og, 0, multipass=True)
    for l, lv in enumerate(H2.levels):
        for b in range(lv.nboxes):
            assert np.array_equal(np.ascontiguousarray(r.mfs[l].valid(b)[1:5]).view(np.int64), np.ascontiguousarray(og[l].valid(b)[0:4]).view(np.int64))
    # curvature: explicit progress range, two levels; bench_json=1 adds one machine-readable line
    import json
    out = _run("curvature3d.ex", ["infile=" + p, "progressName=temp", "is_per=1 1 0", "finestLevel=1", "useFileMinMax=0", "progMin=250", "progMax=2100",
                                  "outfile=" + str(tmp_path / "k1"), "bench_json=1"], tmp_path)
    js = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{"tool"')][-1])
    assert js["tool"] == "curvature3d" and js["cells"] == 2 * 16 ** 3 and set(js["phases_s"]) >= {"read", "upload", "compute", "write"}
    k = read_plotfile(str(tmp_path / "k1"))
    assert k.hier.nlev == 2
    ost = [MultiFab(lv, 1, 2) for lv in H2.levels]
    for l, lv in enumerate(H2.levels):
        for b in range(lv.nboxes):
            ost[l].valid(b)[0] = mfs[l].valid(b)[0]
    oc = [MultiFab(lv, 5, 0) for lv in H2.levels]
    oracle.curvature_pipeline(H2.levels, ost, 0, bc, oc, 0, MultiFab, prog_min=250.0, prog_max=2100.0)
    for l, lv in enumerate(H2.levels):
        for b in range(lv.nboxes):
            got, want = k.mfs[l].valid(b), oc[l].valid(b)
            for gc, wc in ((1, 0), (3, 1), (4, 2), (5, 3), (6, 4)):
                assert np.array_equal(np.ascontiguousarray(got[gc]).view(np.int64), np.ascontiguousarray(want[wc]).view(np.int64)), (l, b, gc)
    # filterPlt: two of three levels, the same 125-tap filter on both, piecewise-constant interpolation
    _run("filterPlt3d.ex", ["infile=" + p, "max_filter_level=1", "same_fgr_all_levels=1", "base_fgr=4", "interp_type=0", "max_grid_size=8", "is_per=1 1 0",
                            "variables=temp", "exact_filter=1"], tmp_path)
    f = read_plotfile(str(tmp_path / "plt00005_filtered"))
    assert f.hier.nlev == 2 and f.names == ["temp"]
    ins = [MultiFab(lv, 1, 2, fill=0.0) for lv in H2.levels]
    for l, lv in enumerate(H2.levels):
        for b in range(lv.nboxes):
            ins[l].valid(b)[0] = mfs[l].valid(b)[0]
    outs = [MultiFab(lv, 1, 0) for lv in H2.levels]
    assert oracle.filter_pipeline(H2.levels, ins, outs, 1, base_fgr=4, same_fgr_all_levels=True, interp_type=0) == [(4, 2), (4, 2)]
    for l, lv in enumerate(H2.levels):
        for b in range(lv.nboxes):
            assert np.array_equal(np.ascontiguousarray(f.mfs[l].valid(b)).view(np.int64), np.ascontiguousarray(outs[l].valid(b)).view(np.int64)), (l, b)
    # isosurface: sComp / nComp instead of comps, two levels
    _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "sComp=0", "nComp=2", "finestLevel=1", "outfile_base=" + str(tmp_path / "s1")],
         tmp_path)
    _, names, nodes, faces = read_mef(str(tmp_path / "s1.mef"))
    assert names == ["X", "Y", "Z", "temp", "x_velocity"]
    fields = [MultiFab(lv, 3, 0, mfs[l].data.copy()) for l, lv in enumerate(H2.levels)]
    lv_np = [Level_np for Level_np in H2.levels]
    import copy
    nonper = []
    for lv in lv_np:  # the isosurface tool defaults to is_per = 0 0 0
        q = copy.copy(lv)
        q.is_per = np.zeros(3, dtype=lv.is_per.dtype if hasattr(lv.is_per, "dtype") else int)
        nonper.append(q)
    onodes, oelts = oracle.isosurface_pipeline(nonper, [MultiFab(q, 3, 0, fields[l].data.copy()) for l, q in enumerate(nonper)], [0, 1], 0, 1150.0, MultiFab)
    assert np.array_equal(faces, oelts + 1) and np.array_equal(nodes.view(np.int64), onodes.view(np.int64))


def _tree_bytes(path):
    """every file of a plotfile directory, by relative name"""
    out = {}
    for root, _, files in os.walk(path):
        for f in files:
            p = os.path.join(root, f)
            out[os.path.relpath(p, path)] = open(p, "rb").read()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args,suffix", [
    ("grad3d.ex", ["gradVar=temp", "is_per=1 1 0", "Aux_Variables=density"], "_gt"),
    ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0", "Aux_Variables=density"], "_K"),
    ("curvature3d.ex", ["progressName=temp", "is_per=0 1 1", "sym_dir=1 0 0", "fused=0", "threshold_prog=1", "threshold_value=0.02", "do_gaussCurv=1",
                        "do_strain=1", "do_velnormal=1"], "_K"),
    ("curvature3d.ex", ["progressName=temp", "is_per=0 1 1", "sym_dir=1 0 0", "threshold_prog=1", "threshold_value=0.02", "do_gaussCurv=1",
                        "do_strain=1", "do_velnormal=1"], "_K"),
    ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0", "do_smooth=1", "smoothing_time=1e-3"], "_K"),
    ("filterPlt3d.ex", ["max_grid_size=8", "is_per=1 1 0"], "_filtered"),
    ("filterPlt3d.ex", ["max_grid_size=8", "interp_type=0", "base_fgr=4", "same_fgr_all_levels=1"], "_filtered"),
    ("isosurface3d.ex", ["isoCompName=temp", "isoVal=1150", "comps=0 4", "outfile_base=surf"], "surf.mef"),
    ("isosurface3d.ex", ["isoCompName=temp", "isoVal=1150", "comps=0 1", "outfile_base=surf", "is_per=1 1 0", "nGrow=2", "surfFormat=XDMF"], "surf.mesh"),
    ("isosurface3d.ex", ["isoCompName=temp", "isoVal=1150", "comps=0", "outfile_base=surf", "build_distance_function=1", "outfile=dist", "nGrow=2"], "DIST"),
])
def test_tools_multi_gpu_outputs_are_byte_identical(tmp_path, tool, args, suffix):
    """ngpus=<n>: the boxes of every level dealt to n ranks (host threads, one HIP context each; here they share the one GPU
    and exchange through the in-process transport).  Same pipelines, cross-rank ghost fills inside the library: every
    output file must be byte-identical to the single-GPU run."""
    p, H, mfs = _synth(tmp_path, nlev=3, base=16, box=8, ncomp=5, names=("temp", "x_velocity", "y_velocity", "z_velocity", "density"))
    ref = None
    # do_smooth: the DISTRIBUTED composite solve (the default on ngpus > 1) sums its dot products in another order, so it equals
    # the one-GPU field to the solve's tolerance (checked below); the replicated form reproduces the one-GPU bits
    smooth = "do_smooth=1" in args
    for n in (1, 2, 4):
        d = tmp_path / f"n{n}"
        d.mkdir()
        # retile=0: the boxes of the FILE are dealt to the ranks (with the internal re-tiling these 16^3 .. 64^3 levels are one box each)
        out = _run(tool, ["infile=" + p] + args + [f"ngpus={n}", "gpu_share=1", "retile=0"], d, env={"PA_SMOOTH_REPLICATED": "1"} if smooth else None)
        if n > 1:
            assert f"distributed over {n} GPUs" in out.stdout
        if suffix.startswith("surf"):  # isosurface: the surface file(s) in the run directory
            got = {f: open(d / f, "rb").read() for f in os.listdir(d) if f.startswith("surf")}
            assert suffix in got and len(got[suffix]) > 10000
        elif suffix == "DIST":  # the distance plotfile (every rank computes the grids of its own FABs) + the surface
            got = _tree_bytes(str(d / "dist"))
            got.update({f: open(d / f, "rb").read() for f in os.listdir(d) if f.startswith("surf")})
            assert len(got) >= 6 and "surf.mef" in got
        else:
            got = _tree_bytes(str(d / ("plt00005" + suffix)))
            assert len(got) >= 5
        if ref is None:
            ref = got
        else:
            assert got.keys() == ref.keys()
            for k in ref:
                assert got[k] == ref[k], f"{tool} ngpus={n}: {k} differs from the single-GPU output"
    if smooth:
        one = read_plotfile(str(tmp_path / "n1" / ("plt00005" + suffix)))
        for n in (2, 4):
            d = tmp_path / f"dist{n}"
            d.mkdir()
            _run(tool, ["infile=" + p] + args + [f"ngpus={n}", "gpu_share=1", "retile=0"], d)
            r = read_plotfile(str(d / ("plt00005" + suffix)))
            assert r.names == one.names
            isp, ipr = r.names.index("SmoothedProgress"), r.names.index("Progress")
            for l in range(len(one.mfs)):
                for b in range(one.hier.levels[l].nboxes):
                    assert np.array_equal(r.mfs[l].valid(b)[ipr], one.mfs[l].valid(b)[ipr])
                    assert np.abs(r.mfs[l].valid(b)[isp] - one.mfs[l].valid(b)[isp]).max() <= 1e-12, f"distributed smoothing solve on {n} ranks, level {l} box {b}"
                    assert np.isfinite(r.mfs[l].valid(b)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args,suffix", [
    ("grad3d.ex", ["gradVar=temp", "is_per=1 1 0", "Aux_Variables=density"], "_gt"),
    ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0", "Aux_Variables=density"], "_K"),
    ("curvature3d.ex", ["progressName=temp", "is_per=0 1 1", "sym_dir=1 0 0", "threshold_prog=1", "threshold_value=0.02", "do_gaussCurv=1", "do_strain=1",
                        "do_velnormal=1"], "_K"),
    ("filterPlt3d.ex", ["max_grid_size=8", "is_per=1 1 0", "exact_filter=1"], "_filtered"),
    ("filterPlt3d.ex", ["max_grid_size=16", "interp_type=0", "base_fgr=4", "same_fgr_all_levels=1", "exact_filter=1"], "_filtered"),
])
def test_tools_retiled_outputs_are_byte_identical(tmp_path, tool, args, suffix):
    """retile=1 (the default): the tool holds and sweeps the level on merged boxes (pa_level_retile) and writes the file's
    BoxArray back -- every byte of the output, the per-FAB minima / maxima of Cell_H included, must equal the run on the
    file's own boxes (retile=0).  Default limits (one box per level here), small limits (PA_RETILE_MAX: several merged boxes
    per level, chains cut), and the small limits sharded over two ranks."""
    p, H, mfs = _synth(tmp_path, nlev=3, base=32, box=8, ncomp=5, names=("temp", "x_velocity", "y_velocity", "z_velocity", "density"))
    ref = None
    for name, extra, env in (("file", ["retile=0"], None), ("default", [], None), ("small", ["retile=1"], {"PA_RETILE_MAX": "16 24 16"}),
                             ("small2", ["ngpus=2", "gpu_share=1"], {"PA_RETILE_MAX": "16 24 16"})):
        d = tmp_path / name
        d.mkdir()
        _run(tool, ["infile=" + p] + args + extra, d, env=env)
        got = _tree_bytes(str(d / ("plt00005" + suffix)))
        assert len(got) >= 5
        if ref is None:
            ref = got
            continue
        assert got.keys() == ref.keys()
        for k in ref:
            assert got[k] == ref[k], f"{tool} {name}: {k} differs from the run on the file's boxes"


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args,suffix", [
    ("grad3d.ex", ["gradVar=temp", "is_per=1 1 0"], "_gt"),
    ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0"], "_K"),
    ("filterPlt3d.ex", ["max_grid_size=8", "is_per=1 1 0"], "_filtered"),
])
def test_tools_run_again_over_their_own_output(tmp_path, tool, args, suffix):
    """A tool run a second time in the same directory: as AMReX's UtilCreateCleanDirectory does inside WriteMultiLevelPlotfile
    (grad.cpp:256, curvature.cpp:843, filterPlt.cpp:52), the old plotfile is renamed to <name>.old.<unique> at write time and
    KEPT (pa::OldOutput) -- the second run's files are the first run's bytes, stale files of the old directory are not in the new
    one, the renamed directory is whole; remove_old_output=1 removes it once the new plotfile is complete; a run that aborts on its
    inputs touches nothing; an output path that contains the input aborts before anything is renamed."""
    p, H, mfs = _synth(tmp_path, nlev=3, base=16, box=8, ncomp=3)
    d = tmp_path / "run"
    d.mkdir()
    _run(tool, ["infile=" + p] + args, d)
    outdir = d / ("plt00005" + suffix)
    first = _tree_bytes(str(outdir))
    assert len(first) >= 5
    (outdir / "stale_file_of_the_first_run").write_bytes(b"x" * 100)
    _run(tool, ["infile=" + p] + args, d)
    assert _tree_bytes(str(outdir)) == first
    olds = [f for f in os.listdir(d) if ".old." in f]
    kept = _tree_bytes(str(d / olds[0]))
    assert len(olds) == 1 and kept.pop("stale_file_of_the_first_run") == b"x" * 100 and kept == first
    _run(tool, ["infile=" + p] + args + ["remove_old_output=1"], d)
    assert _tree_bytes(str(outdir)) == first and [f for f in os.listdir(d) if ".old." in f] == olds
    # an unknown variable aborts after the Header is read: the output of the earlier run stays as it is
    bad = {"grad3d.ex": "gradVar=nope", "curvature3d.ex": "progressName=nope", "filterPlt3d.ex": "variables=nope"}[tool]
    out = subprocess.run([os.path.join(BIN, tool), "infile=" + p] + args + [bad], cwd=d, capture_output=True, text=True)
    assert out.returncode != 0
    assert _tree_bytes(str(outdir)) == first and [f for f in os.listdir(d) if ".old." in f] == olds
    if tool != "filterPlt3d.ex":  # outfile= is a key of grad / curvature only
        out = subprocess.run([os.path.join(BIN, tool), "infile=" + p] + args + ["outfile=" + str(tmp_path)], cwd=d, capture_output=True, text=True)
        assert out.returncode != 0 and "contains the input plotfile" in out.stderr
        assert os.path.isdir(p) and _tree_bytes(str(outdir)) == first and not [f for f in os.listdir(tmp_path.parent) if ".old." in f]


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args,suffix", [
    ("grad3d.ex", ["gradVar=temp", "is_per=1 1 0"], "_gt"),
    ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0", "fused=0"], "_K"),
    ("curvature3d.ex", ["progressName=temp", "is_per=1 1 0"], "_K"),
    ("filterPlt3d.ex", ["max_grid_size=8", "is_per=1 1 0"], "_filtered"),
    ("isosurface3d.ex", ["isoCompName=temp", "isoVal=1150", "comps=0 1", "outfile_base=surf"], "surf.mef"),
])
def test_tools_multi_gpu_sparse_levels(tmp_path, tool, args, suffix):
    """More ranks than boxes: every level is TWO adjacent 8^3 boxes, dealt to 3 and 4 ranks, so some ranks own nothing and
    never enter a level's ghost exchange (the library calls a transport only on ranks that send or receive).  A
    transport that waits for all ranks hangs here or pairs the lists of two different exchanges (round-2 advisor finding
    on the in-process transport); run under a timeout, outputs byte-identical to the single-GPU run."""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
    per = np.asarray((1, 1, 0))
    plo, phi = np.zeros(3), np.asarray((1.0, 0.5, 0.5))
    l0 = Level(chop_box((0, 0, 0), (15, 7, 7), 8), np.zeros(3, dtype=np.int64), np.asarray((15, 7, 7)), per, plo, phi)
    l1 = Level(chop_box((8, 4, 4), (23, 11, 11), 8), np.zeros(3, dtype=np.int64), np.asarray((31, 15, 15)), per, plo, phi)
    H = Hierarchy([l0, l1], 2)
    assert l0.nboxes == 2 and l1.nboxes == 2
    mfs = make_states(H, 3, 0, field_flame, seed=5)
    p = str(tmp_path / "plt00005")
    write_plotfile(p, H, mfs, ["temp", "x_velocity", "density"], time=0.125, level_steps=[5, 5])
    ref = None
    for n in (1, 3, 4):
        d = tmp_path / f"n{n}"
        d.mkdir()
        _run(tool, ["infile=" + p] + args + [f"ngpus={n}", "gpu_share=1"], d, timeout=120)
        if suffix.startswith("surf"):
            got = {f: open(d / f, "rb").read() for f in os.listdir(d) if f.startswith("surf")}
            assert suffix in got
        else:
            got = _tree_bytes(str(d / ("plt00005" + suffix)))
            assert len(got) >= 4
        if ref is None:
            ref = got
        else:
            assert got.keys() == ref.keys()
            for k in ref:
                assert got[k] == ref[k], f"{tool} ngpus={n}: {k} differs from the single-GPU output"


@pytest.mark.gpu
def test_isosurface_tool_full_size_config4_closed_manifold(tmp_path):
    """BASELINE config 4 at full size through the drop-in binary: T-isotherm of the wrinkled-ellipsoid flame field on the
    3-level base-256^3 hierarchy (5.0e7 cells, 64^3 boxes), the surface crossing both coarse-fine interfaces (degenerate
    hexes).  Size-independent properties of the MEF it writes: every directed edge is used once and its reverse once
    (closed, consistently oriented manifold: the checkIso.cpp:127-149 invariant, also through checkIso3d.ex strict=1),
    Euler characteristic 2, every node on the iso value, mapped component interpolated, node ids contiguous; and the
    staging file of surface_is_large holds the same nodes in chunks of at most chunk_size."""
    from peleanalysis_amd.hierarchy import fill_analytic
    H = nested_hierarchy(256, 3, 64, is_per=(0, 0, 0))
    mfs = []
    for lv in H.levels:
        m = MultiFab(lv, 2, 0)
        fill_analytic(m, 0, lambda x, y, z: field_flame(x, y, z, 0))
        fill_analytic(m, 1, lambda x, y, z: x + 2.0 * y + 3.0 * z + 0 * x * y * z)
        mfs.append(m)
    p = str(tmp_path / "plt_c4")
    write_plotfile(p, H, mfs, ["temp", "lin"], time=0.5, level_steps=[1, 2, 4])
    del mfs
    out = _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 1", "outfile_base=" + str(tmp_path / "surf"), "surface_is_large=1",
                                   "chunk_size=50000", "tmpFile=" + str(tmp_path / "stage.fab"), "verbose=1"], tmp_path)
    _label, names, nodes, faces = read_mef(str(tmp_path / "surf.mef"))
    assert names == ["X", "Y", "Z", "temp", "lin"]
    elts = faces - 1
    assert len(elts) > 300000 and elts.min() == 0 and elts.max() == len(nodes) - 1
    assert np.unique(elts).size == len(nodes), "orphan nodes"
    a, b = np.concatenate([elts[:, 0], elts[:, 1], elts[:, 2]]).astype(np.int64), np.concatenate([elts[:, 1], elts[:, 2], elts[:, 0]]).astype(np.int64)
    fwd, rev = a * len(nodes) + b, b * len(nodes) + a
    assert np.unique(fwd).size == fwd.size, "a directed edge is used twice (orientation flip or duplicate element)"
    assert np.array_equal(np.sort(fwd), np.sort(rev)), "open edge: the surface is not closed"
    V, E, F = len(nodes), fwd.size // 2, len(elts)
    assert V - E + F == 2, (V, E, F)
    assert np.abs(nodes[:, 3] - 1150.0).max() < 1e-9
    assert np.abs(nodes[:, 4] - (nodes[:, 0] + 2.0 * nodes[:, 1] + 3.0 * nodes[:, 2])).max() < 1e-12
    host = subprocess.run([os.path.join(BIN, "isosurface3d.ex"), "infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 1", "outfile_base=" + str(tmp_path / "surf_host")],
                          cwd=tmp_path, capture_output=True, text=True, env=dict(os.environ, PA_ISO_HOST_MERGE="1"))
    assert host.returncode == 0, host.stderr
    assert open(tmp_path / "surf_host.mef", "rb").read() == open(tmp_path / "surf.mef", "rb").read(), "device merge and sequential host merge differ"
    chk = _run("checkIso3d.ex", ["isoFile=" + str(tmp_path / "surf.mef"), "strict=1"], tmp_path)  # exit 2 on an edge traversed twice the same way
    assert "All shared edges are consistently numbered." in chk.stdout
    # the staging file: FABs of <= chunk_size nodes, node-major, in order
    raw = open(tmp_path / "stage.fab", "rb").read()
    pos, got = 0, []
    while pos < len(raw):
        eol = raw.index(b"\n", pos)
        hdr = raw[pos:eol].decode()
        import re
        m = re.search(r"\(\((\d+),0,0\) \((\d+),0,0\) \(0,0,0\)\) (\d+)", hdr)
        lo, hi, nc = int(m.group(1)), int(m.group(2)), int(m.group(3))
        n = hi - lo + 1
        assert n <= 50000 and nc == 5 and lo == sum(len(g) for g in got)
        got.append(np.frombuffer(raw, dtype=np.float64, count=n * nc, offset=eol + 1).reshape(n, nc))
        pos = eol + 1 + 8 * n * nc
    assert np.array_equal(np.vstack(got).view(np.int64), np.ascontiguousarray(nodes).view(np.int64))
    assert "staging vertex data to disk in %d chunks" % len(got) in out.stdout


def _synth_ratio4(tmp_path, per=(0, 0, 0)):
    """2-level plotfile with refinement ratio 4: base 16^3 in 8^3 boxes, level 1 = coarse cells [4, 11]^3 refined to [16, 47]^3 in 16^3 boxes"""
    from peleanalysis_amd.hierarchy import Hierarchy, Level, chop_box
    l0 = Level(chop_box((0, 0, 0), (15, 15, 15), 8), (0, 0, 0), (15, 15, 15), per, np.zeros(3), np.ones(3))
    l1 = Level(chop_box((16, 16, 16), (47, 47, 47), 16), (0, 0, 0), (63, 63, 63), per, np.zeros(3), np.ones(3))
    H = Hierarchy([l0, l1], 4)
    from util import make_states, field_flame
    mfs = make_states(H, 3, 0, field_flame, seed=3)
    p = str(tmp_path / "plt00004")
    write_plotfile(p, H, mfs, ["temp", "x_velocity", "density"], time=0.25)
    return p, H, mfs


@pytest.mark.gpu
def test_filter_and_isosurface_tools_refinement_ratio_4(tmp_path, oracle):
    """the two tools whose reference takes the plotfile's refinement ratio (filterPlt.cpp:133,200; isosurface.cpp:1472,1518,1543)
    on a ratio-4 plotfile: filterPlt's filter-to-grid ratio grows by 4 (fgr 2 -> 8: 729 taps on level 1) and its ghost fill
    interpolates with ratio-4 offsets; isosurface masks by the finer level coarsened by 4 and fills coarse-fine ghosts from the
    parent of 4^3 children.  Both against the oracle run with ratio = 4; grad keeps refusing such a file like the reference's
    hard-coded 2 would mis-handle it"""
    p, H, mfs = _synth_ratio4(tmp_path)
    assert read_plotfile(p).hier.ref_ratio == 4
    _run("filterPlt3d.ex", ["infile=" + p, "max_grid_size=16", "variables=temp", "exact_filter=1"], tmp_path)
    r = read_plotfile(str(tmp_path / "plt00004_filtered"))
    ins = []
    for l, lv in enumerate(H.levels):
        s = MultiFab(lv, 1, 4)
        for b in range(lv.nboxes):
            s.valid(b)[0] = mfs[l].valid(b)[0]
        ins.append(s)
    outs = [MultiFab(lv, 1, 0) for lv in H.levels]
    info = oracle.filter_pipeline(H.levels, ins, outs, 1, base_fgr=2, ratio=4, interp_type=1)
    assert info == [(2, 1), (8, 4)]
    for l, lv in enumerate(H.levels):
        assert np.array_equal(r.hier.levels[l].boxes, lv.boxes)
        for b in range(lv.nboxes):
            assert np.array_equal(np.ascontiguousarray(r.mfs[l].valid(b)).view(np.int64), np.ascontiguousarray(outs[l].valid(b)).view(np.int64)), f"ratio 4 filter level {l} box {b}"
    _run("isosurface3d.ex", ["infile=" + p, "isoCompName=temp", "isoVal=1150", "comps=0 2"], tmp_path)
    label, names, nodes, faces = read_mef(p + "_temp_1150.mef")
    fields = [MultiFab(lv, 3, 0, mfs[l].data.copy()) for l, lv in enumerate(H.levels)]
    onodes, oelts = oracle.isosurface_pipeline(H.levels, fields, [0, 2], 0, 1150.0, MultiFab, ratio=4)
    assert len(oelts) > 100
    assert np.array_equal(faces, oelts + 1), "connectivity (1-based) differs"
    assert np.array_equal(nodes.view(np.int64), onodes.view(np.int64)), "node data not bit-identical"
    bad = subprocess.run([os.path.join(BIN, "grad3d.ex"), "infile=" + p], cwd=tmp_path, capture_output=True, text=True)
    assert bad.returncode != 0 and "refinement ratio 2" in bad.stderr
