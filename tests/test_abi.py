"""The C-ABI shared library loads on a GPU-less host and exports every symbol that
include/peleanalysis_amd.h declares; pure-host entry points agree with the python mirrors.
No compute calls are made here (no GPU in the CPU test tier)."""
import ctypes as C
import os

import numpy as np

from peleanalysis_amd import capi
from peleanalysis_amd.hierarchy import chop_box, comp_stride, mf_layout


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    declared = capi.declared_symbols()
    assert len(declared) >= 40
    assert [s for s in declared if not hasattr(lib, s)] == []
    assert lib._pa_missing == []
    # every declared function has a ctypes signature in the binding (same names, same arity source)
    assert sorted(set(declared) - set(lib._pa_signatures)) == []
    assert lib.pa_version() >= 100


def test_context_fails_loudly_without_gpu():
    """no CPU fallback: creating a context on a host without a HIP device raises"""
    import pytest
    lib = capi.load_library()
    if lib.pa_ctx_create(0, None):  # a GPU is present (GPU box): nothing to check here
        return
    with pytest.raises(capi.PaError):
        capi.Context(0)


def test_mf_layout_matches_python_mirror():
    lib = capi.load_library()
    boxes = np.vstack([chop_box((0, 0, 0), (127, 127, 127), 128), chop_box((128, 0, 0), (159, 40, 20), 11)]).astype(np.int32)
    for ncomp, ng in ((1, 0), (1, 2), (8, 0), (3, 1), (10, 4)):
        off = np.zeros(len(boxes), dtype=np.int64)
        cs = np.zeros(len(boxes), dtype=np.int64)
        tot = lib.pa_mf_layout(len(boxes), boxes.ctypes.data_as(C.POINTER(C.c_int32)), ncomp, ng, off.ctypes.data_as(C.POINTER(C.c_int64)),
                               cs.ctypes.data_as(C.POINTER(C.c_int64)))
        poff, pcs, ptot = mf_layout(boxes, ncomp, ng)
        assert tot == ptot and np.array_equal(off, poff) and np.array_equal(cs, pcs)
        assert np.all(off % 64 == 0) and np.all(cs % 64 == 0)  # 512-byte alignment of every fab and component
        n = np.prod(boxes[:, 3:] - boxes[:, :3] + 1 + 2 * ng, axis=1)
        assert np.all(cs >= n)
        if ncomp > 1:
            assert np.all(cs % 2048 != 0)  # never a multiple of 16 KiB (HBM channel aliasing)
            assert np.all((cs % 2048 == 256) | (n < 32768))  # boxes of >= 32^3 cells: 2 KiB past a multiple of 16 KiB
            assert np.all(cs - n < 2048 + 64)  # at most 16.5 KiB of padding per component
    assert int(comp_stride(128 ** 3, 8)) == 128 ** 3 + 256 and int(comp_stride(128 ** 3, 1)) == 128 ** 3
    assert int(comp_stride(16 ** 3, 8)) == 16 ** 3 + 64 and int(comp_stride(20 ** 3, 8)) == 8000


def test_box_filter_weights_host_entry(oracle):
    lib = capi.load_library()
    for fgr in (1, 2, 4, 6, 8, 16):
        w = (C.c_double * (fgr + 2))()
        ng = lib.pa_box_filter_weights(fgr, w)
        ong, ow = oracle.box_filter_weights(fgr)
        assert ng == ong and np.array_equal(np.array(w[:2 * ng + 1]), ow)
    assert lib.pa_box_filter_weights(0, (C.c_double * 4)()) < 0
    import os
    os.environ.pop("PA_ALLOW_UNVERIFIED_GAUSSIAN", None)
    lib.pa_options_reload()  # the library reads its switches once: re-read after every change
    assert lib.pa_filter_weights(2, 2, (C.c_double * 40)()) < 0  # the Gaussian is refused unless the caller opts in (weights unverified against PelePhysics)
    for ftype in range(-1, 12):  # the other filter types: the library's weights are the oracle's, bit for bit; the same types are refused
        for fgr in (1, 2, 4, 6, 16):
            w = (C.c_double * 40)()  # up to 2 * 16 + 1 weights (the Gaussian, type 2, is the widest)
            if ftype == 2:
                os.environ["PA_ALLOW_UNVERIFIED_GAUSSIAN"] = "1"
                lib.pa_options_reload()
            ng = lib.pa_filter_weights(ftype, fgr, w)
            os.environ.pop("PA_ALLOW_UNVERIFIED_GAUSSIAN", None)
            lib.pa_options_reload()
            want = oracle.filter_weights(ftype, fgr)
            assert (ng < 0) == (want is None), (ftype, fgr)
            if want is not None:
                assert ng == want[0] and np.array_equal(np.array(w[:2 * ng + 1]).view(np.int64), want[1].view(np.int64)), (ftype, fgr)


def test_mc_tables_exported_match_oracle(oracle):
    lib = capi.load_library()
    e = np.ctypeslib.as_array(lib.pa_mc_edge_table(), shape=(256,)).astype(np.int32)
    t = np.ctypeslib.as_array(lib.pa_mc_tri_table(), shape=(256, 16)).astype(np.int32)
    oe, ot = oracle.mc_tables()
    assert np.array_equal(e, oe) and np.array_equal(t, ot)


def test_header_cites_reference_call_sites():
    txt = open(os.path.join(os.path.dirname(capi.__file__), "..", "include", "peleanalysis_amd.h")).read()
    for cite in ("grad.cpp:211-236", "curvature.cpp:451-502", "filterPlt.cpp:217", "isosurface.cpp:1531-1592"):
        assert cite in txt
