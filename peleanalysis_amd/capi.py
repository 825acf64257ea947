"""ctypes binding of libpeleanalysis_amd.so (the C ABI in include/peleanalysis_amd.h).

This is the host-side mirror used by tests/ and bench.py; the C++ tool drivers in
tools/ call the same C ABI directly.  There is NO CPU fallback: if the HIP library is
missing or no GPU is present, constructing a Context raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from .hierarchy import Hierarchy, Level, MultiFab, mf_layout

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpeleanalysis_amd.so")

BC_PERIODIC, BC_NEUMANN, BC_REFLECT_ODD = 0, 1, 2


class PaFab(C.Structure):
    _fields_ = [("p", C.c_void_p), ("lo", C.c_int32 * 3), ("hi", C.c_int32 * 3), ("ncomp", C.c_int32), ("nstride", C.c_int64)]


class PaBox(C.Structure):
    _fields_ = [("lo", C.c_int32 * 3), ("hi", C.c_int32 * 3)]


class PaCurvParams(C.Structure):
    _fields_ = [("prog_min", C.c_double), ("prog_max", C.c_double), ("do_threshold", C.c_int32), ("threshold", C.c_double),
                ("fused", C.c_int32), ("do_gauss_curv", C.c_int32), ("do_strain", C.c_int32), ("get_strain_tensor", C.c_int32),
                ("do_velnormal", C.c_int32), ("vel_comp", C.c_int32), ("do_smooth", C.c_int32), ("smoothing_time", C.c_double),
                ("spacedim", C.c_int32)]


class PaXfer(C.Structure):
    _fields_ = [("peer", C.c_int32), ("sendbuf", C.c_void_p), ("nsend", C.c_int64), ("recvbuf", C.c_void_p), ("nrecv", C.c_int64)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(PaXfer))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32)


DONE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
DONE2_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)


class PaComm(C.Structure):
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int32), ("nranks", C.c_int32), ("exchange", EXCHANGE_FN), ("allreduce", ALLREDUCE_FN)]


class PaIsoFrag(C.Structure):
    _fields_ = [("verts", C.c_void_p), ("nvert", C.c_int64), ("tris", C.c_void_p), ("ntri", C.c_int64)]


class PaSdfGrid(C.Structure):
    _fields_ = [("ntri", C.c_int64), ("tri", C.c_void_p), ("nvert", C.c_int64), ("x", C.c_void_p), ("origin", C.c_float * 3), ("dx", C.c_float),
                ("n", C.c_int32 * 3), ("phi", C.c_void_p)]


_lib = None


def load_library() -> C.CDLL:
    """dlopen the HIP library; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pi32, pdbl = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    sig = {
        "pa_version": (C.c_int, []),
        "pa_options_reload": (None, []),
        "pa_ctx_create": (vp, [C.c_int, vp]),
        "pa_ctx_destroy": (None, [vp]),
        "pa_last_error": (C.c_char_p, [vp]),
        "pa_sync": (C.c_int, [vp]),
        "pa_ctx_stream": (vp, [vp]),
        "pa_device_malloc": (vp, [vp, i64]),
        "pa_device_free": (None, [vp, vp]),
        "pa_memcpy_h2d": (C.c_int, [vp, vp, vp, i64]),
        "pa_memcpy_d2h": (C.c_int, [vp, vp, vp, i64]),
        "pa_memcpy_d2d": (C.c_int, [vp, vp, vp, i64]),
        "pa_device_count": (C.c_int, []),
        "pa_profile_enable": (C.c_int, [vp, C.c_int]),
        "pa_sweep_kernel_name": (C.c_char_p, [vp]),
        "pa_sweep_occupancy": (C.c_int, [vp, C.c_int]),
        "pa_profile_read": (C.c_int, [vp, C.c_int, C.POINTER(i64), pdbl, C.c_int]),
        "pa_level_create": (vp, [vp, C.c_int, pi32, pi32, pi32, pi32, pdbl, pdbl]),
        "pa_level_create_sharded": (vp, [vp, C.c_int, pi32, pi32, C.c_int, C.c_int, pi32, pi32, pi32, pdbl, pdbl]),
        "pa_level_global_ids": (C.c_int, [vp, pi32]),
        "pa_ctx_set_comm": (C.c_int, [vp, C.POINTER(PaComm)]),
        "pa_ctx_set_delay_comm": (C.c_int, [vp, C.c_int, C.c_int, dbl, dbl]),
        "pa_delay_comm_stats": (C.c_int, [vp, C.POINTER(i64), pdbl]),
        "pa_rccl_unique_id": (C.c_int, [vp, vp]),
        "pa_ctx_init_rccl": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "pa_ctx_nranks": (C.c_int, [vp]),
        "pa_comm_selftest": (C.c_int, [vp, i64]),
        "pa_allreduce": (C.c_int, [vp, pdbl, C.c_int, C.c_int]),
        "pa_distribution_map": (C.c_int, [C.c_int, pi32, C.c_int, pi32]),
        "pa_plan_fill_boundary": (i64, [C.c_int, pi32, pi32, C.c_int, pi32, pi32, pi32, C.c_int, pi32, i64]),
        "pa_plan_coarse_source": (i64, [C.c_int, pi32, pi32, pi32, pi32, C.c_int, pi32, pi32, pi32, pi32, pi32, C.c_int, C.c_int, C.c_int, C.c_int, pi32,
                                        i64]),
        "pa_plan_restriction": (i64, [C.c_int, pi32, pi32, pi32, pi32, C.c_int, pi32, pi32, pi32, pi32, pi32, C.c_int, C.c_int, C.c_int, pi32, i64]),
        "pa_level_retile": (C.c_int, [C.c_int, pi32, pi32, C.c_int, pi32, C.c_int]),
        "pa_hierarchy_retile_limits": (C.c_int, [C.c_int, pi32, C.POINTER(pi32), C.c_int, pi32]),
        "pa_hierarchy_retile_limits_ranks": (C.c_int, [C.c_int, pi32, C.POINTER(pi32), C.c_int, C.c_int, pi32]),
        "pa_level_destroy": (None, [vp]),
        "pa_level_nboxes": (C.c_int, [vp]),
        "pa_mf_layout": (i64, [C.c_int, pi32, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]),
        "pa_mf_create": (vp, [vp, vp, C.c_int, C.c_int, vp]),
        "pa_mf_destroy": (None, [vp]),
        "pa_mf_data": (vp, [vp]),
        "pa_mf_size": (i64, [vp]),
        "pa_mf_upload": (C.c_int, [vp, vp, vp]),
        "pa_mf_download": (C.c_int, [vp, vp, vp]),
        "pa_mf_upload_comps": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
        "pa_mf_download_comps": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
        "pa_mf_setval": (C.c_int, [vp, vp, C.c_int, C.c_int, dbl]),
        "pa_mf_copy": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]),
        "pa_fill_boundary": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
        "pa_apply_bc": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, pi32, C.c_int, C.c_int]),
        "pa_bc_errors": (C.c_int, [vp]),
        "pa_grad_level": (C.c_int, [vp, vp, C.c_int, vp, C.c_int]),
        "pa_minmax_level": (C.c_int, [vp, vp, C.c_int, pdbl, pdbl]),
        "pa_progress_level": (C.c_int, [vp, vp, C.c_int, dbl, dbl, vp, C.c_int, C.c_int]),
        "pa_normal_level": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int]),
        "pa_div_level": (C.c_int, [vp, vp, C.c_int, dbl, vp, C.c_int, dbl, vp, C.c_int]),
        "pa_progress_shell_level": (C.c_int, [vp, vp, C.c_int, dbl, dbl, vp, C.c_int, C.c_int, C.c_int]),
        "pa_gradcurv_level": (C.c_int, [vp, vp, C.c_int, dbl, dbl, dbl, vp, C.c_int]),
        "pa_gradcurv_faces_level": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, pi32, C.c_int, dbl, vp, C.c_int, C.c_int]),
        "pa_grad_fab": (C.c_int, [vp, PaBox, C.POINTER(PaFab), C.c_int, pdbl, C.POINTER(PaFab), C.c_int]),
        "pa_progress_fab": (C.c_int, [vp, PaBox, C.POINTER(PaFab), C.c_int, dbl, dbl, C.POINTER(PaFab), C.c_int]),
        "pa_normal_fab": (C.c_int, [vp, PaBox, C.POINTER(PaFab), C.c_int, pdbl, C.POINTER(PaFab), C.c_int, C.POINTER(PaFab), C.c_int,
                                    C.POINTER(PaFab), C.c_int]),
        "pa_div_fab": (C.c_int, [vp, PaBox, C.POINTER(PaFab), C.c_int, pdbl, dbl, C.POINTER(PaFab), C.c_int]),
        "pa_gradcurv_fab": (C.c_int, [vp, PaBox, C.POINTER(PaFab), C.c_int, dbl, dbl, pdbl, dbl, C.POINTER(PaFab), C.c_int]),
        "pa_boxfilter_fab": (C.c_int, [vp, PaBox, C.POINTER(PaFab), C.POINTER(PaFab), C.c_int, C.c_int, C.c_int, pdbl]),
        "pa_box_filter_weights": (C.c_int, [C.c_int, pdbl]),
        "pa_filter_weights": (C.c_int, [C.c_int, C.c_int, pdbl]),
        "pa_boxfilter_level": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, pdbl]),
        "pa_boxfilter_hierarchy": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, pi32, C.POINTER(pdbl)]),
        "pa_boxfilter_level2d": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, pdbl]),
        "pa_foextrap": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
        "pa_fillpatch_two_levels": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "pa_fill_ghosts_hierarchy": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_int, C.c_int, pi32, C.c_int, C.c_int, C.c_int]),
        "pa_mc_count_fab": (C.c_int, [vp, PaBox, C.POINTER(PaFab), C.POINTER(PaFab), C.c_int, dbl, C.POINTER(i64), C.POINTER(i64)]),
        "pa_mc_emit_fab": (C.c_int, [vp, PaBox, C.POINTER(PaFab), C.POINTER(PaFab), C.c_int, dbl, vp, vp, vp, i64, i64]),
        "pa_iso_mask_level": (C.c_int, [vp, vp, C.c_int, vp, C.c_int]),
        "pa_iso_coords_level": (C.c_int, [vp, vp, C.c_int]),
        "pa_mc_level_fine": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(PaBox), C.c_int, dbl, C.POINTER(i64), C.POINTER(i64), C.POINTER(vp), C.POINTER(vp),
                                       C.POINTER(vp)]),
        "pa_msq_level_fine": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(PaBox), C.c_int, dbl, C.POINTER(i64), C.POINTER(i64), C.POINTER(vp), C.POINTER(vp),
                                        C.POINTER(vp)]),
        "pa_mc_hierarchy_fine": (C.c_int, [vp, C.c_int, C.POINTER(vp), pi32, C.c_int, C.POINTER(C.POINTER(PaBox)), C.c_int, dbl, C.POINTER(C.POINTER(i64)),
                                           C.POINTER(C.POINTER(i64)), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "pa_mc_hierarchy_xyz": (C.c_int, [vp, C.c_int, C.POINTER(vp), pi32, C.c_int, C.POINTER(C.POINTER(PaBox)), C.c_int, dbl, C.POINTER(C.POINTER(i64)),
                                          C.POINTER(C.POINTER(i64)), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "pa_msq_level": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(PaBox), C.c_int, dbl, C.POINTER(i64), C.POINTER(i64), C.POINTER(vp), C.POINTER(vp),
                                   C.POINTER(vp)]),
        "pa_mc_level": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(PaBox), C.c_int, dbl, C.POINTER(i64), C.POINTER(i64), C.POINTER(vp), C.POINTER(vp),
                                  C.POINTER(vp)]),
        "pa_iso_merge": (C.c_int, [vp, C.c_int, C.POINTER(PaIsoFrag), C.c_int, C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), C.POINTER(vp)]),
        "pa_mc_edge_table": (C.POINTER(C.c_uint16), []),
        "pa_mc_tri_table": (C.POINTER(C.c_int8), []),
        "pa_sdf_level_set3": (C.c_int, [vp, C.c_int, C.POINTER(PaSdfGrid), C.c_int]),
        "pa_sdf_signed_fab": (C.c_int, [vp, PaBox, vp, C.POINTER(PaFab), C.c_int, dbl, dbl, C.POINTER(PaFab), C.c_int]),
        "pa_smooth_solve": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp), C.c_int, dbl, pi32, dbl, C.c_int, C.POINTER(C.c_int), pdbl]),
        "pa_stream_trace": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_int, i64, pdbl, C.c_int, dbl, vp, pi32]),
        "pa_last_slow_cells": (C.c_int, [vp]),
        "pa_level_irregular_cells": (i64, [vp, vp]),
        "pa_stream_trace_ranks": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_int, i64, pdbl, C.c_int, dbl, vp, pi32, C.c_int]),
        "pa_smooth_last": (C.c_int, [vp, C.POINTER(C.c_int), pdbl]),
        "pa_curvature_last_path": (C.c_int, [vp]),
        "pa_level_free_scratch": (i64, [vp]),
        "pa_grad_run": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_int, pi32, C.POINTER(vp), C.c_int]),
        "pa_curvature_run": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_int, pi32, C.POINTER(PaCurvParams), C.POINTER(vp), C.c_int]),
        "pa_gradcurv_run": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_int, pi32, C.POINTER(PaCurvParams), C.POINTER(vp), C.POINTER(vp),
                                      C.c_int]),
        "pa_gradcurv_run_comps": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_int, C.c_int, pi32, C.POINTER(PaCurvParams), C.POINTER(vp), C.POINTER(vp),
                                            C.c_int, DONE_FN, vp]),
        "pa_gradcurv_run_comps2": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.c_int, C.c_int, pi32, C.POINTER(PaCurvParams), C.POINTER(vp), C.POINTER(vp),
                                             C.c_int, C.c_int, DONE2_FN, vp]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:  # symbol missing from the .so: calling it later raises AttributeError
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    L._pa_signatures = sig
    L._pa_missing = missing  # tests/test_abi.py asserts this is empty
    _lib = L
    return L


def declared_symbols(header: Optional[str] = None) -> List[str]:
    """Names of every function declared in include/peleanalysis_amd.h."""
    import re
    header = header or os.path.join(os.path.dirname(_HERE), "include", "peleanalysis_amd.h")
    txt = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", txt)))


def reload_options() -> None:
    """pa_options_reload: the library reads its PA_* environment switches once (first context); a caller that changes one
    afterwards -- a test, bench.py --ab -- asks for a re-read."""
    load_library().pa_options_reload()


class PaError(RuntimeError):
    pass


def _i3(v):
    return (C.c_int32 * 3)(*[int(x) for x in v])


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class Context:
    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = load_library()
        self.h = self.lib.pa_ctx_create(int(device), C.c_void_p(stream) if stream else None)
        if not self.h:
            raise PaError("pa_ctx_create failed: no MI355X/HIP device visible (no CPU fallback)")

    def check(self, rc: int):
        if rc != 0:
            raise PaError(self.lib.pa_last_error(self.h).decode())

    def sync(self):
        self.check(self.lib.pa_sync(self.h))

    def profile_enable(self, on=True):
        """False / True: no / every tag; an int > 1: bit mask (1 << tag) of the tags to time"""
        self.check(self.lib.pa_profile_enable(self.h, int(on)))

    def profile_read(self, tag: int, reset: bool = False):
        n, ms = C.c_int64(0), C.c_double(0.0)
        self.check(self.lib.pa_profile_read(self.h, tag, C.byref(n), C.byref(ms), int(reset)))
        return n.value, ms.value

    def bc_errors(self) -> int:
        return int(self.lib.pa_bc_errors(self.h))

    # ---- multi-GPU transports (include/peleanalysis_amd.h: pa_comm)
    def set_comm(self, comm: "PaComm"):
        """caller-supplied transport; the structure (and its callbacks) must outlive the context"""
        self._comm = comm
        self.check(self.lib.pa_ctx_set_comm(self.h, C.byref(comm) if comm is not None else None))

    def rccl_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self.check(self.lib.pa_rccl_unique_id(self.h, buf))
        return buf.raw

    def init_rccl(self, nranks: int, rank: int, unique_id: bytes):
        """built-in transport: grouped ncclSend / ncclRecv on the context's stream (RCCL over xGMI)"""
        assert len(unique_id) == 128
        self.check(self.lib.pa_ctx_init_rccl(self.h, int(nranks), int(rank), C.create_string_buffer(unique_id, 128)))

    def comm_selftest(self, n: int = 4096):
        self.check(self.lib.pa_comm_selftest(self.h, int(n)))

    def allreduce(self, vals, op: int):
        a = np.ascontiguousarray(vals, dtype=np.float64).copy()
        self.check(self.lib.pa_allreduce(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size, int(op)))
        return a

    def close(self):
        if self.h:
            self.lib.pa_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class DevBuf:
    """raw HBM buffer from pa_device_malloc (freed with the object)"""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.ptr = ctx.lib.pa_device_malloc(ctx.h, self.nbytes)
        if not self.ptr:
            raise PaError(ctx.lib.pa_last_error(ctx.h).decode())

    @classmethod
    def from_numpy(cls, ctx: Context, a: np.ndarray) -> "DevBuf":
        a = np.ascontiguousarray(a)
        b = cls(ctx, a.nbytes)
        ctx.check(ctx.lib.pa_memcpy_h2d(ctx.h, b.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return b

    def to_numpy(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx.check(self.ctx.lib.pa_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out

    def __del__(self):
        try:
            if self.ptr and self.ctx.h:
                self.ctx.lib.pa_device_free(self.ctx.h, self.ptr)
        except Exception:
            pass


class DevLevel:
    def __init__(self, ctx: Context, level: Level, owner: Optional[Sequence[int]] = None, rank: int = 0, nranks: int = 1):
        """level: the whole BoxArray of the AMR level.  owner (one rank per box) + rank + nranks: this context holds only the
        boxes with owner == rank (pa_level_create_sharded); self.level then describes those boxes, self.gids their indices
        in the BoxArray, self.glob the whole level."""
        self.ctx, self.glob = ctx, level
        b = np.ascontiguousarray(level.boxes, dtype=np.int32)
        if owner is None:
            self.level, self.gids = level, np.arange(level.nboxes)
            self.h = ctx.lib.pa_level_create(ctx.h, level.nboxes, b.ctypes.data_as(C.POINTER(C.c_int32)), _i3(level.domlo), _i3(level.domhi),
                                             _i3(level.is_per), _d3(level.prob_lo), _d3(level.prob_hi))
        else:
            o = np.ascontiguousarray(owner, dtype=np.int32)
            assert len(o) == level.nboxes
            self.gids = np.nonzero(o == rank)[0]
            self.level = Level(level.boxes[self.gids], level.domlo, level.domhi, level.is_per, level.prob_lo, level.prob_hi)
            self.h = ctx.lib.pa_level_create_sharded(ctx.h, level.nboxes, b.ctypes.data_as(C.POINTER(C.c_int32)), o.ctypes.data_as(C.POINTER(C.c_int32)),
                                                     int(rank), int(nranks), _i3(level.domlo), _i3(level.domhi), _i3(level.is_per),
                                                     _d3(level.prob_lo), _d3(level.prob_hi))
        if not self.h:
            raise PaError(ctx.lib.pa_last_error(ctx.h).decode())

    def close(self):
        """pa_level_destroy: the level's device arrays, cached plans and work multifabs.  There is NO finaliser: a level (and a multifab)
        lives until close() -- or the end of a `with` block -- is reached; destroy the multifabs on a level before the level, the levels
        before their context"""
        if self.h:
            self.ctx.lib.pa_level_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class DevMF:
    """MultiFab in HBM.  devptr: optional externally owned device pointer (e.g. torch tensor)."""

    def __init__(self, ctx: Context, dlev: DevLevel, ncomp: int, ng: int, devptr: Optional[int] = None):
        self.ctx, self.dlev, self.ncomp, self.ng = ctx, dlev, int(ncomp), int(ng)
        self.h = ctx.lib.pa_mf_create(ctx.h, dlev.h, ncomp, ng, C.c_void_p(devptr) if devptr else None)
        if not self.h:
            raise PaError(ctx.lib.pa_last_error(ctx.h).decode())
        self.size = int(ctx.lib.pa_mf_size(self.h))
        self.ptr = int(ctx.lib.pa_mf_data(self.h))

    @classmethod
    def from_host(cls, ctx, dlev, mf: MultiFab) -> "DevMF":
        d = cls(ctx, dlev, mf.ncomp, mf.ng)
        d.upload(mf)
        return d

    def upload(self, mf: MultiFab):
        assert mf.total == self.size, (mf.total, self.size)
        self.ctx.check(self.ctx.lib.pa_mf_upload(self.ctx.h, self.h, mf.data.ctypes.data_as(C.c_void_p)))

    def download(self) -> MultiFab:
        mf = MultiFab(self.dlev.level, self.ncomp, self.ng)
        assert mf.total == self.size
        self.ctx.check(self.ctx.lib.pa_mf_download(self.ctx.h, self.h, mf.data.ctypes.data_as(C.c_void_p)))
        return mf

    def setval(self, v: float, comp: int = 0, ncomp: Optional[int] = None):
        """pa_mf_setval: components comp .. comp + ncomp - 1 (default: all) of every FAB, ghost cells included"""
        self.ctx.check(self.ctx.lib.pa_mf_setval(self.ctx.h, self.h, int(comp), self.ncomp - int(comp) if ncomp is None else int(ncomp), float(v)))

    def fab(self, b: int) -> PaFab:
        """pa_fab for box b (device pointer into this multifab)."""
        lv = self.dlev.level
        off, cs, _ = mf_layout(lv.boxes, self.ncomp, self.ng)
        f = PaFab()
        f.nstride = int(cs[b])
        f.p = self.ptr + 8 * int(off[b])
        for d in range(3):
            f.lo[d] = int(lv.boxes[b, d]) - self.ng
            f.hi[d] = int(lv.boxes[b, 3 + d]) + self.ng
        f.ncomp = self.ncomp
        return f

    def close(self):
        if self.h:
            self.ctx.lib.pa_mf_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def box_of(level: Level, b: int, grow: int = 0) -> PaBox:
    bx = PaBox()
    for d in range(3):
        bx.lo[d] = int(level.boxes[b, d]) - grow
        bx.hi[d] = int(level.boxes[b, 3 + d]) + grow
    return bx


def _handles(mfs: Sequence[Optional[DevMF]]):
    return (C.c_void_p * len(mfs))(*[m.h if m is not None else None for m in mfs])


def bc_from_flags(is_per, sym_dir=(0, 0, 0)):
    """grad.cpp:180-193 / curvature.cpp:428-441"""
    return [BC_PERIODIC if p else (BC_REFLECT_ODD if s else BC_NEUMANN) for p, s in zip(is_per, sym_dir)]


def grad_run(ctx: Context, states: Sequence[DevMF], comp: int, bc, outs: Sequence[DevMF], ocomp: int):
    ctx.check(ctx.lib.pa_grad_run(ctx.h, len(states), _handles(states), comp, _i3(bc), _handles(outs), ocomp))


def curv_params(prog_min=None, prog_max=None, threshold=None, fused=True, do_gauss=False, do_strain=False, strain_tensor=False,
                do_velnormal=False, vel_comp=0, do_smooth=False, smoothing_time=1e-7, spacedim=3) -> PaCurvParams:
    p = PaCurvParams()
    p.spacedim = int(spacedim)
    p.do_smooth, p.smoothing_time = int(do_smooth), float(smoothing_time)
    p.do_gauss_curv, p.do_strain, p.get_strain_tensor, p.do_velnormal, p.vel_comp = int(do_gauss), int(do_strain), int(strain_tensor), int(do_velnormal), int(vel_comp)
    p.prog_min = 1e20 if prog_min is None else prog_min
    p.prog_max = -1e20 if prog_max is None else prog_max
    p.do_threshold = 0 if threshold is None else 1
    p.threshold = 0.0 if threshold is None else float(threshold)
    p.fused = 1 if fused else 0
    return p


def curvature_run(ctx, states, comp, bc, params: PaCurvParams, outs, ocomp):
    ctx.check(ctx.lib.pa_curvature_run(ctx.h, len(states), _handles(states), comp, _i3(bc), C.byref(params), _handles(outs), ocomp))


def gradcurv_run(ctx, states, comp, bc, params: PaCurvParams, works, outs, ocomp):
    ctx.check(ctx.lib.pa_gradcurv_run(ctx.h, len(states), _handles(states), comp, _i3(bc), C.byref(params), _handles(works),
                                      _handles(outs), ocomp))


def gradcurv_run_comps(ctx, states, comp0, ncomps, bc, params: PaCurvParams, works, outs, ocomp, done=None):
    """pa_gradcurv_run_comps; done(comp) is called when a component's results are complete in outs (sync before reading)"""
    err = []

    def _cb(user, comp):
        try:
            done(comp)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            err.append(e)
            return 1
    cb = DONE_FN(_cb) if done is not None else C.cast(None, DONE_FN)
    rc = ctx.lib.pa_gradcurv_run_comps(ctx.h, len(states), _handles(states), int(comp0), int(ncomps), _i3(bc), C.byref(params), _handles(works),
                                       _handles(outs), int(ocomp), cb, None)
    if err:
        raise err[0]
    ctx.check(rc)


def gradcurv_run_comps2(ctx, states, comp0, ncomps, bc, params: PaCurvParams, works, outs, ocomp, nbatch, done=None):
    """pa_gradcurv_run_comps2: batches of nbatch components, outs hold nbatch slots of 8 components from ocomp;
    done(comp, ocomp_of_its_slot) is called for every component of a batch once the batch is complete (sync before reading)"""
    err = []

    def _cb(user, comp, oc):
        try:
            done(comp, oc)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            err.append(e)
            return 1
    cb = DONE2_FN(_cb) if done is not None else C.cast(None, DONE2_FN)
    rc = ctx.lib.pa_gradcurv_run_comps2(ctx.h, len(states), _handles(states), int(comp0), int(ncomps), _i3(bc), C.byref(params), _handles(works),
                                        _handles(outs), int(ocomp), int(nbatch), cb, None)
    if err:
        raise err[0]
    ctx.check(rc)


def mc_hierarchy(ctx: Context, states, fine_mask, loops_per_level, isocomp: int, isoval: float, download: bool = True, ratio: int = 2, xyz: bool = False):
    """pa_mc_hierarchy_fine: marching cubes on every level in one call.  states: DevMF per level; fine_mask: flag per level
    (mask by the next finer level); loops_per_level: (nboxes, 6) arrays.  Returns per level the per-box list
    [(verts, vkeys, tris)] (download=False: only the per-box counts [(nv, nt)]).
    xyz=True: pa_mc_hierarchy_xyz -- the states hold the fields only (isocomp among them), vertex coordinates come from cell indices"""
    nlev = len(states)
    arrs, nvs, nts = [], [], []
    for l in range(nlev):
        lp = np.asarray(loops_per_level[l], dtype=np.int64).reshape(-1, 6)
        nb = len(lp)
        arr = (PaBox * max(nb, 1))()
        for b in range(nb):
            for d in range(3):
                arr[b].lo[d], arr[b].hi[d] = int(lp[b, d]), int(lp[b, 3 + d])
        arrs.append(arr)
        nvs.append((C.c_int64 * max(nb, 1))())
        nts.append((C.c_int64 * max(nb, 1))())
    parr = (C.POINTER(PaBox) * nlev)(*[C.cast(a, C.POINTER(PaBox)) for a in arrs])
    pnv = (C.POINTER(C.c_int64) * nlev)(*[C.cast(a, C.POINTER(C.c_int64)) for a in nvs])
    pnt = (C.POINTER(C.c_int64) * nlev)(*[C.cast(a, C.POINTER(C.c_int64)) for a in nts])
    fm = (C.c_int32 * nlev)(*[int(bool(f)) for f in fine_mask])
    pv, pk, pt = (C.c_void_p * nlev)(), (C.c_void_p * nlev)(), (C.c_void_p * nlev)()
    block = C.c_void_p()
    fn = ctx.lib.pa_mc_hierarchy_xyz if xyz else ctx.lib.pa_mc_hierarchy_fine
    ctx.check(fn(ctx.h, nlev, _handles(states), fm, int(ratio), parr, int(isocomp), float(isoval), pnv, pnt, pv, pk, pt, C.byref(block)))
    out = []
    try:
        for l in range(nlev):
            nb = len(np.asarray(loops_per_level[l]).reshape(-1, 6))
            nv, nt = nvs[l], nts[l]
            if not download:
                out.append([(int(nv[b]), int(nt[b])) for b in range(nb)])
                continue
            nc = states[l].ncomp + (3 if xyz else 0)
            tv, tt = int(sum(nv[:nb])), int(sum(nt[:nb]))
            V = np.empty((tv, nc)); K = np.empty((tv, 6), np.int32); T = np.empty((tt, 3), np.int32)
            if tv:
                ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, V.ctypes.data_as(C.c_void_p), pv[l], V.nbytes))
                ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, K.ctypes.data_as(C.c_void_p), pk[l], K.nbytes))
            if tt:
                ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, T.ctypes.data_as(C.c_void_p), pt[l], T.nbytes))
            lev, ov, ot = [], 0, 0
            for b in range(nb):
                lev.append((V[ov:ov + nv[b]], K[ov:ov + nv[b]], T[ot:ot + nt[b]]))
                ov += nv[b]; ot += nt[b]
            out.append(lev)
    finally:
        if block.value:
            ctx.lib.pa_device_free(ctx.h, block)
    return out


def sdf_level_set(ctx: Context, meshes, exact_band: int = 1):
    """pa_sdf_level_set3 on a batch: meshes = list of (tris (nt,3) uint32, verts (nv,3) float32, origin, dx, (ni,nj,nk));
    returns the list of phi arrays, float32 (nk, nj, ni)."""
    grids = (PaSdfGrid * len(meshes))()
    keep, outs = [], []
    for g, (tris, verts, origin, dx, n) in zip(grids, meshes):
        tris = np.ascontiguousarray(tris, dtype=np.uint32).reshape(-1, 3)
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        dt = DevBuf.from_numpy(ctx, tris) if len(tris) else None
        dv = DevBuf.from_numpy(ctx, verts) if len(verts) else None
        ni, nj, nk = (int(v) for v in n)
        dp = DevBuf(ctx, 4 * ni * nj * nk)
        keep += [dt, dv]
        outs.append((dp, (nk, nj, ni)))
        g.ntri, g.tri, g.nvert, g.x = len(tris), (dt.ptr if dt else None), len(verts), (dv.ptr if dv else None)
        for d in range(3):
            g.origin[d] = np.float32(origin[d])
            g.n[d] = (ni, nj, nk)[d]
        g.dx = np.float32(dx)
        g.phi = dp.ptr
    ctx.check(ctx.lib.pa_sdf_level_set3(ctx.h, len(meshes), grids, int(exact_band)))
    ctx.sync()
    return [dp.to_numpy(np.float32, shape) for dp, shape in outs]


def smooth_solve(ctx, rhs, rcomp, sol, scomp, dt, bc, tol=1e-12, maxiter=100):
    """pa_smooth_solve; returns (iterations, relative residual)"""
    it, res = C.c_int(0), C.c_double(0.0)
    ctx.check(ctx.lib.pa_smooth_solve(ctx.h, len(rhs), _handles(rhs), rcomp, _handles(sol), scomp, float(dt), _i3(bc), float(tol), int(maxiter),
                                      C.byref(it), C.byref(res)))
    return it.value, res.value


def stream_trace(ctx, vfield, vcomp, seeds, nsteps, dt):
    """pa_stream_trace -> (pos float64 [2*nseed][nsteps][3], number of redistributions)"""
    seeds = np.ascontiguousarray(seeds, dtype=np.float64).reshape(-1, 3)
    n = len(seeds)
    buf = DevBuf(ctx, max(8 * 2 * n * nsteps * 3, 8))
    nred = C.c_int32(0)
    ctx.check(ctx.lib.pa_stream_trace(ctx.h, len(vfield), _handles(vfield), int(vcomp), n, seeds.ctypes.data_as(C.POINTER(C.c_double)), int(nsteps),
                                      float(dt), C.c_void_p(buf.ptr), C.byref(nred)))
    return buf.to_numpy(np.float64, (2 * n, nsteps, 3)), nred.value


def iso_merge(ctx: Context, fragments, ncomp: int):
    """pa_iso_merge on host fragments [(verts [nv][ncomp], tris [nt][3])] (uploaded here).  Returns (nodes [n][ncomp],
    elts [m][3] int32), or None when the library reports clusters that are not transitive under the tolerance (code 2)."""
    bufs, arr = [], (PaIsoFrag * max(len(fragments), 1))()
    for f, (v, t) in enumerate(fragments):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1, ncomp)
        t = np.ascontiguousarray(t, dtype=np.int32).reshape(-1, 3)
        bv = DevBuf.from_numpy(ctx, v) if len(v) else None
        bt = DevBuf.from_numpy(ctx, t) if len(t) else None
        bufs += [bv, bt]
        arr[f].verts, arr[f].nvert, arr[f].tris, arr[f].ntri = (bv.ptr if bv else None), len(v), (bt.ptr if bt else None), len(t)
    nn, ne, pn, pe = C.c_int64(0), C.c_int64(0), C.c_void_p(), C.c_void_p()
    rc = ctx.lib.pa_iso_merge(ctx.h, len(fragments), arr, int(ncomp), C.byref(nn), C.byref(pn), C.byref(ne), C.byref(pe))
    if rc == 2:
        return None
    ctx.check(rc)
    try:
        nodes, elts = np.empty((nn.value, ncomp)), np.empty((ne.value, 3), np.int32)
        if nn.value:
            ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, nodes.ctypes.data_as(C.c_void_p), pn, nodes.nbytes))
        if ne.value:
            ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, elts.ctypes.data_as(C.c_void_p), pe, elts.nbytes))
    finally:
        for q in (pn, pe):
            if q.value:
                ctx.lib.pa_device_free(ctx.h, q)
    return nodes, elts


def mc_level(ctx: Context, state: "DevMF", mask: "DevMF", loops, isocomp: int, isoval: float, mcomp: int = 0, squares: bool = False):
    """Level-batched marching cubes (pa_mc_level) or, with squares=True, marching squares on the plane k = 0
    (pa_msq_level; segments come back as rows (id0, id1, -1)).  loops: (nboxes, 6) base-point boxes (lo > hi: skipped).
    Returns per-box lists [(verts [nv][ncomp], vkeys [nv][6], tris [nt][3] FAB-local ids)]."""
    loops = np.asarray(loops, dtype=np.int64).reshape(-1, 6)
    nb = len(loops)
    arr = (PaBox * max(nb, 1))()
    for b in range(nb):
        for d in range(3):
            arr[b].lo[d], arr[b].hi[d] = int(loops[b, d]), int(loops[b, 3 + d])
    nv, nt = (C.c_int64 * max(nb, 1))(), (C.c_int64 * max(nb, 1))()
    pv, pk, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    if mask is None or isinstance(mask, DevLevel):  # mask evaluated in the cell pass from the finer level (or nothing masked)
        fn = ctx.lib.pa_msq_level_fine if squares else ctx.lib.pa_mc_level_fine
        ctx.check(fn(ctx.h, state.h, mask.h if mask is not None else None, 2, arr, isocomp, isoval, nv, nt, C.byref(pv), C.byref(pk), C.byref(pt)))
    else:
        fn = ctx.lib.pa_msq_level if squares else ctx.lib.pa_mc_level
        ctx.check(fn(ctx.h, state.h, mask.h, mcomp, arr, isocomp, isoval, nv, nt, C.byref(pv), C.byref(pk), C.byref(pt)))
    nc = state.ncomp
    tv, tt = int(sum(nv[:nb])), int(sum(nt[:nb]))
    try:
        V = np.empty((tv, nc)); K = np.empty((tv, 6), np.int32); T = np.empty((tt, 3), np.int32)
        if tv:
            ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, V.ctypes.data_as(C.c_void_p), pv, V.nbytes))
            ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, K.ctypes.data_as(C.c_void_p), pk, K.nbytes))
        if tt:
            ctx.check(ctx.lib.pa_memcpy_d2h(ctx.h, T.ctypes.data_as(C.c_void_p), pt, T.nbytes))
    finally:
        if pv.value:  # one allocation, base = the vertex array
            ctx.lib.pa_device_free(ctx.h, pv)
    out, ov, ot = [], 0, 0
    for b in range(nb):
        out.append((V[ov:ov + nv[b]], K[ov:ov + nv[b]], T[ot:ot + nt[b]]))
        ov += nv[b]; ot += nt[b]
    return out
