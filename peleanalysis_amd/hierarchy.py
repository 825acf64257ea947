"""Host-side (numpy) description of a block-structured AMR hierarchy.

Pure bookkeeping -- no numerics of the hot path live here.  Mirrors the pieces of
AMReX the reference tools use to describe data: Box / BoxArray / Geometry
(grad.cpp:158-170) and the FArrayBox memory layout ([comp][k][j][i], i fastest).
The flat-buffer layout is the one of ``pa_mf_layout`` in include/peleanalysis_amd.h.
"""
from __future__ import annotations

import dataclasses
from typing import List, Sequence

import numpy as np


@dataclasses.dataclass
class Level:
    boxes: np.ndarray  # (n, 6) int32: lo0 lo1 lo2 hi0 hi1 hi2 (inclusive)
    domlo: np.ndarray  # (3,) int32
    domhi: np.ndarray  # (3,) int32
    is_per: np.ndarray  # (3,) int32
    prob_lo: np.ndarray  # (3,) float64
    prob_hi: np.ndarray  # (3,) float64

    def __post_init__(self):
        self.boxes = np.ascontiguousarray(self.boxes, dtype=np.int32).reshape(-1, 6)
        self.domlo = np.ascontiguousarray(self.domlo, dtype=np.int32)
        self.domhi = np.ascontiguousarray(self.domhi, dtype=np.int32)
        self.is_per = np.ascontiguousarray(self.is_per, dtype=np.int32)
        self.prob_lo = np.ascontiguousarray(self.prob_lo, dtype=np.float64)
        self.prob_hi = np.ascontiguousarray(self.prob_hi, dtype=np.float64)

    @property
    def nboxes(self) -> int:
        return self.boxes.shape[0]

    @property
    def dx(self) -> np.ndarray:
        return (self.prob_hi - self.prob_lo) / (self.domhi - self.domlo + 1).astype(np.float64)

    def box_shape(self, b: int, ng: int = 0):
        lo, hi = self.boxes[b, :3], self.boxes[b, 3:]
        n = hi - lo + 1 + 2 * ng
        return int(n[2]), int(n[1]), int(n[0])  # (nz, ny, nx)

    @property
    def ncells(self) -> int:
        n = self.boxes[:, 3:].astype(np.int64) - self.boxes[:, :3] + 1
        return int(np.prod(n, axis=1).sum())


def comp_stride(ncells, ncomp: int):
    """Same rule as pa_cstride: cells rounded up to 64 doubles (512 B); with more than one component, boxes of >= 32^3
    cells sit 2 KiB past a multiple of 16 KiB and smaller ones are kept off multiples of 16 KiB (HBM address interleave,
    see csrc/pa_internal.h)."""
    n = np.asarray(ncells, dtype=np.int64)
    cs = (n + 63) // 64 * 64
    if ncomp > 1:
        big = cs + (256 + 2048 - cs % 2048) % 2048
        small = np.where(cs % 2048 == 0, cs + 64, cs)
        cs = np.where(n >= 32768, big, small)
    return cs


def mf_layout(boxes: np.ndarray, ncomp: int, ng: int):
    """Same rule as pa_mf_layout: returns (box offsets, component strides, total) in doubles."""
    n = boxes[:, 3:].astype(np.int64) - boxes[:, :3] + 1 + 2 * ng
    cs = comp_stride(np.prod(n, axis=1), ncomp)
    size = ncomp * cs
    off = np.zeros(len(boxes), dtype=np.int64)
    off[1:] = np.cumsum(size)[:-1]
    return off, cs.astype(np.int64), int(size.sum())


class MultiFab:
    """Host multifab: one flat float64 buffer + per-box numpy views."""

    def __init__(self, level: Level, ncomp: int, ng: int, data: np.ndarray | None = None, fill: float = 0.0):
        self.level = level
        self.ncomp = int(ncomp)
        self.ng = int(ng)
        self.off, self.cstride, self.total = mf_layout(level.boxes, ncomp, ng)
        if data is None:
            data = np.full(self.total, fill, dtype=np.float64)
        assert data.dtype == np.float64 and data.size == self.total
        self.data = data

    def fab(self, b: int) -> np.ndarray:
        nz, ny, nx = self.level.box_shape(b, self.ng)
        cs = int(self.cstride[b])
        base = self.data[self.off[b]:self.off[b] + self.ncomp * cs]
        return np.lib.stride_tricks.as_strided(base, shape=(self.ncomp, nz, ny, nx), strides=(8 * cs, 8 * ny * nx, 8 * nx, 8))

    def valid(self, b: int) -> np.ndarray:
        g = self.ng
        f = self.fab(b)
        return f[:, g:f.shape[1] - g, g:f.shape[2] - g, g:f.shape[3] - g] if g else f

    def copy(self) -> "MultiFab":
        return MultiFab(self.level, self.ncomp, self.ng, self.data.copy())

    def valid_concat(self, comp: int | None = None) -> np.ndarray:
        """All valid cells, box by box, flattened (for whole-level comparisons)."""
        out = []
        for b in range(self.level.nboxes):
            v = self.valid(b)
            out.append((v if comp is None else v[comp]).ravel())
        return np.concatenate(out)


@dataclasses.dataclass
class Hierarchy:
    levels: List[Level]
    ref_ratio: int = 2

    @property
    def nlev(self) -> int:
        return len(self.levels)


# ----------------------------------------------------------------------------- builders
def chop_box(lo: Sequence[int], hi: Sequence[int], max_size: int) -> np.ndarray:
    """BoxArray::maxSize restated: chop into pieces <= max_size per direction (even split)."""
    lo = np.asarray(lo)
    hi = np.asarray(hi)
    cuts = []
    for d in range(3):
        n = hi[d] - lo[d] + 1
        nparts = (n + max_size - 1) // max_size
        base, rem = divmod(n, nparts)
        sizes = [base + (1 if p < rem else 0) for p in range(nparts)]
        starts = lo[d] + np.concatenate([[0], np.cumsum(sizes)[:-1]])
        cuts.append([(int(s), int(s + z - 1)) for s, z in zip(starts, sizes)])
    boxes = []
    for (z0, z1) in cuts[2]:
        for (y0, y1) in cuts[1]:
            for (x0, x1) in cuts[0]:
                boxes.append([x0, y0, z0, x1, y1, z1])
    return np.asarray(boxes, dtype=np.int32)


def nested_hierarchy(base_n: int, nlev: int, box_size: int, is_per=(1, 1, 0), prob_lo=(0.0, 0.0, 0.0),
                     prob_hi=(1.0, 1.0, 1.0)) -> Hierarchy:
    """level l+1 refines the central half-width cube of level l (ratio 2),
    so every level has base_n^3 cells; nested, centred, convex refinement regions."""
    levels = []
    lo = np.zeros(3, dtype=np.int64)
    hi = np.full(3, base_n - 1, dtype=np.int64)
    domlo = np.zeros(3, dtype=np.int64)
    domhi = np.full(3, base_n - 1, dtype=np.int64)
    for l in range(nlev):
        levels.append(Level(chop_box(lo, hi, box_size), domlo.copy(), domhi.copy(), np.asarray(is_per), np.asarray(prob_lo, float),
                            np.asarray(prob_hi, float)))
        n = hi - lo + 1
        clo = lo + n // 4
        chi = clo + n // 2 - 1
        lo, hi = 2 * clo, 2 * chi + 1
        domlo, domhi = 2 * domlo, 2 * domhi + 1
    return Hierarchy(levels, 2)


def cell_centers(level: Level, b: int, ng: int):
    """x, y, z coordinate arrays (broadcastable to (nz,ny,nx)) of box b grown by ng."""
    lo = level.boxes[b, :3]
    nz, ny, nx = level.box_shape(b, ng)
    dx = level.dx
    x = level.prob_lo[0] + (np.arange(lo[0] - ng, lo[0] - ng + nx) + 0.5) * dx[0]
    y = level.prob_lo[1] + (np.arange(lo[1] - ng, lo[1] - ng + ny) + 0.5) * dx[1]
    z = level.prob_lo[2] + (np.arange(lo[2] - ng, lo[2] - ng + nz) + 0.5) * dx[2]
    return x[None, None, :], y[None, :, None], z[:, None, None]


def fill_analytic(mf: MultiFab, comp: int, fn, valid_only: bool = True) -> None:
    """mf[comp] = fn(x,y,z) at cell centres (ghosts left untouched if valid_only)."""
    for b in range(mf.level.nboxes):
        if valid_only:
            x, y, z = cell_centers(mf.level, b, 0)
            mf.valid(b)[comp] = fn(x, y, z)
        else:
            x, y, z = cell_centers(mf.level, b, mf.ng)
            mf.fab(b)[comp] = fn(x, y, z)


# ----------------------------------------------------------------------------- synthetic fields
def field_trig(x, y, z, m: int = 0):
    """C1/C2 field: smooth periodic, exact gradient known."""
    ph = 0.37 * m
    return (1.0 + 0.1 * m) * (1000.0 + 600.0 * np.sin(2 * np.pi * x + ph) * np.cos(4 * np.pi * y) * np.sin(2 * np.pi * z + 0.3))


def field_flame(x, y, z, m: int = 0):
    """C3/headline field: wrinkled ellipsoidal tanh front crossing both c/f interfaces."""
    xc, yc, zc = x - 0.5, y - 0.5, z - 0.5
    r = np.sqrt((xc / 0.30) ** 2 + (yc / 0.15) ** 2 + (zc / 0.18) ** 2)
    theta = np.arctan2(yc, xc)
    rho = np.sqrt(xc * xc + yc * yc + zc * zc) + 1e-30
    phi = np.arccos(np.clip(zc / rho, -1.0, 1.0))
    s = r - 0.03 * np.sin(6 * theta) * np.sin(5 * phi)
    return (1.0 + 0.1 * m) * (300.0 + 850.0 * (1.0 + np.tanh((s - 1.0) / 0.08))) + 3.0 * m * np.sin(2 * np.pi * (x + 0.37 * m))
