"""The library's environment switches are ONE documented table (peleanalysis_amd/csrc/pa_internal.h: pa_options), read once and
re-read on request (pa_options_reload); the tools add five of their own.  Round 5 ended with 76 ad-hoc getenv sites; this file
keeps the table honest: nothing reads the environment outside pa_options_reload, every switch is in the table, every switch has
a test that flips it (named below), and the two that only print or only change how a tool exits are flipped here."""
import glob
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "peleanalysis_amd", "csrc")

# switch -> the test that flips it
FLIPPED_BY = {
    "PA_FILTER_EXACT": "tests/conftest.py::filter_mode (every filter test runs both)",
    "PA_ALLOW_UNVERIFIED_GAUSSIAN": "tests/test_abi.py::test_box_filter_weights_host_entry",
    "PA_RETILE_MAX": "tests/test_plotfile_tools.py::test_tools_retiled_outputs..., tests/test_sanitizers.py",
    "PA_FUSED2": "tests/test_gpu_gradcurv.py::test_gradcurv_fused_wide_boxes[first]",
    "PA_NCG": "tests/test_gpu_production_geometry.py (x faces mirrored on / off, same bits)",
    "PA_DIST_EARLY": "tests/test_gpu_dist.py::_worker (early tiles, same bits)",
    "PA_SMOOTH_REPLICATED": "tests/test_gpu_dist.py::test_sharded_smoothing_solve...[+rep]",
    "PA_SMOOTH_MG": "tests/test_gpu_smooth.py::test_smooth_multigrid_preconditioner",
    "PA_SMOOTH_MARCH": "tests/test_gpu_smooth.py::test_smooth_wide_boxes_marching_kernels",
    "PA_SMOOTH_TIMING": "tests/test_options.py::test_smooth_timing_and_tool_exit",
    "PA_SCRATCH_POISON": "tests/test_gpu_gradcurv.py::test_work_multifabs_are_never_read_before_they_are_written",
    "PA_FORCE_FALLBACKS": "tests/test_gpu_gradcurv.py::test_switched_off_paths_still_match, test_gpu_filter_mc.py (tiles, fillpatch)",
    # the tools' own (tools/common, tools/src)
    "PA_HOST_THP": "tests/test_plotfile_tools.py::test_cpp_template_tool_large_level_with_and_without_huge_pages",
    "PA_IO_THREADS": "tests/test_sanitizers.py",
    "PA_ISO_HOST_MERGE": "tests/test_plotfile_tools.py::test_isosurface_tool_end_to_end, tests/test_golden.py",
    "PA_ISO_XYZ": "tests/test_plotfile_tools.py::test_isosurface_tool_end_to_end",
    "PA_TOOL_EXIT": "tests/test_options.py::test_smooth_timing_and_tool_exit",
}


def _reads(paths):
    names = {}
    for p in paths:
        for m in re.finditer(r'(?:getenv|geti)\("(PA_[A-Z0-9_]+)"', open(p).read()):
            names.setdefault(m.group(1), set()).add(os.path.relpath(p, ROOT))
    return names


def test_every_switch_is_in_the_table_and_nowhere_else():
    lib = _reads(glob.glob(CSRC + "/*.hip") + glob.glob(CSRC + "/*.h"))
    # the library reads the environment in ONE function
    assert all(v == {"peleanalysis_amd/csrc/pa_core.hip"} for v in lib.values()), lib
    core = open(os.path.join(CSRC, "pa_core.hip")).read()
    body = core[core.index('extern "C" void pa_options_reload(void) {'):core.index("const pa_options& pa_opt()")]
    assert len(re.findall(r'(?:getenv|geti)\("PA_', core)) == len(re.findall(r'(?:getenv|geti)\("PA_', body)), "a getenv outside pa_options_reload"
    # ... and every one of them is documented in the struct
    table = open(os.path.join(CSRC, "pa_internal.h")).read()
    table = table[table.index("struct pa_options {"):table.index("const pa_options& pa_opt();")]
    documented = set(re.findall(r"//\s*(PA_[A-Z0-9_]+)", table))
    assert set(lib) == documented, (sorted(set(lib) ^ documented))
    tools = _reads(glob.glob(ROOT + "/tools/common/*.h") + glob.glob(ROOT + "/tools/src/*.cpp"))
    everything = set(lib) | set(tools)
    assert len(everything) <= 25, sorted(everything)
    assert everything == set(FLIPPED_BY), sorted(everything ^ set(FLIPPED_BY))
    # the tests named above exist and mention their switch
    for name, where in FLIPPED_BY.items():
        files = re.findall(r"tests/[a-z_0-9]+\.py", where)
        assert files and any(name in open(os.path.join(ROOT, f)).read() for f in files), (name, where)


def test_reload_is_exported_and_harmless_without_a_gpu():
    from peleanalysis_amd import capi
    capi.reload_options()
    capi.reload_options()


@pytest.mark.gpu
def test_smooth_timing_and_tool_exit(tmp_path):
    """PA_SMOOTH_TIMING=1: the solve reports its setup / iteration times on stderr (nothing else changes); PA_TOOL_EXIT=normal: the
    tool leaves through exit() instead of _Exit() -- the same files either way"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_plotfile_tools import BIN, _synth, _tree_bytes
    p, H, mfs = _synth(tmp_path, nlev=2, base=16, box=8, ncomp=1, names=("temp",))
    outs = {}
    for tag, env in (("plain", {}), ("switches", {"PA_SMOOTH_TIMING": "1", "PA_TOOL_EXIT": "normal"})):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([os.path.join(BIN, "curvature3d.ex"), "infile=" + p, "is_per=1 1 0", "do_smooth=1", "smoothing_time=1e-3"], cwd=d, capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        assert ("pa_smooth_solve: setup" in r.stderr) == (tag == "switches"), r.stderr
        outs[tag] = _tree_bytes(str(d / "plt00005_K"))
    assert outs["plain"] == outs["switches"] and len(outs["plain"]) >= 4
