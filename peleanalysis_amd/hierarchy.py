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


# ----------------------------------------------------------------------------- general (non-convex) BoxArrays
# What a Pele plotfile's fine levels look like: unions of rectangles -- L / T shapes, disjoint patches, box faces that are
# partly covered by a neighbour and partly coarse-fine, concave coarse-fine corners (the reference runs the same MLPoisson /
# FillBoundary code on whatever BoxArray the file holds, grad.cpp:173-213, curvature.cpp:426-457).
def _occupancy(level: Level) -> np.ndarray:
    """bool[nz, ny, nx] over the level's domain: valid cells"""
    n = level.domhi - level.domlo + 1
    occ = np.zeros((int(n[2]), int(n[1]), int(n[0])), dtype=bool)
    for lo0, lo1, lo2, hi0, hi1, hi2 in level.boxes - np.concatenate([level.domlo, level.domlo]):
        occ[lo2:hi2 + 1, lo1:hi1 + 1, lo0:hi0 + 1] = True
    return occ


def _nested_blocks(U: np.ndarray, s: int, occ: np.ndarray, is_per, buf: int = 2) -> np.ndarray:
    """drop the blocks (s coarse cells per side, bool[bz, by, bx]) whose `buf`-cell halo is not valid coarse data: proper
    nesting -- the fine ghost cells (2 layers = 1 coarse cell) and the +-1 (one-sided: 2) coarse cells of the coarse-fine
    interpolation stencils must find coarse values.  Cells beyond a wall do not count, periodic directions wrap."""
    nz, ny, nx = occ.shape
    out = U.copy()
    for bk, bj, bi in np.argwhere(U):
        idx = []
        ok = True
        for d, (b, n) in enumerate(((bi, nx), (bj, ny), (bk, nz))):
            r = np.arange(b * s - buf, (b + 1) * s + buf)
            if is_per[d]:
                r = r % n
            else:
                r = r[(r >= 0) & (r < n)]
            idx.append(r)
        if not occ[np.ix_(idx[2], idx[1], idx[0])].all():
            ok = False
        out[bk, bj, bi] = ok
    return out


def blocks_to_boxes(U: np.ndarray, s: int, max_blocks, rng=None) -> np.ndarray:
    """greedy merge of a block mask (bool[bz, by, bx], s cells per block side) into boxes of at most max_blocks blocks per
    direction: grow in x, then y, then z while every block is set and unassigned (what a grid generator's chop + merge gives)"""
    U = U.copy()
    bz, by, bx = U.shape
    mb = np.broadcast_to(np.asarray(max_blocks), (3,))
    boxes = []
    for k in range(bz):
        for j in range(by):
            for i in range(bx):
                if not U[k, j, i]:
                    continue
                lim = [int(rng.integers(1, m + 1)) if rng is not None else int(m) for m in mb]
                i1 = i
                while i1 + 1 < bx and i1 + 1 - i < lim[0] and U[k, j, i1 + 1]:
                    i1 += 1
                j1 = j
                while j1 + 1 < by and j1 + 1 - j < lim[1] and U[k, j1 + 1, i:i1 + 1].all():
                    j1 += 1
                k1 = k
                while k1 + 1 < bz and k1 + 1 - k < lim[2] and U[k1 + 1, j:j1 + 1, i:i1 + 1].all():
                    k1 += 1
                U[k:k1 + 1, j:j1 + 1, i:i1 + 1] = False
                boxes.append([i * s, j * s, k * s, (i1 + 1) * s - 1, (j1 + 1) * s - 1, (k1 + 1) * s - 1])
    return np.asarray(boxes, dtype=np.int32).reshape(-1, 6)


def union_hierarchy(seed: int, nlev: int = 3, n0=None, is_per=None, nrect=(2, 5), block=(2, 5), max_blocks=3, base_box=None) -> Hierarchy:
    """random hierarchy whose fine levels are UNIONS of 2-4 rectangles of blocks (ratio 2): L / T shapes, disjoint patches,
    partly covered faces, concave coarse-fine corners; properly nested (2 coarse cells of buffer, or flush with the domain)"""
    rng = np.random.default_rng(seed)
    n = np.asarray(n0 if n0 is not None else rng.integers(12, 33, size=3), dtype=np.int64)
    per = np.asarray(is_per if is_per is not None else rng.integers(0, 2, size=3))
    dom_hi = n - 1
    size0 = int(base_box if base_box is not None else rng.integers(6, 20))
    levels = [Level(chop_box((0, 0, 0), dom_hi, size0), (0, 0, 0), dom_hi, per, np.zeros(3), np.ones(3))]
    for l in range(1, nlev):
        crse = levels[-1]
        occ = _occupancy(crse)
        cn = crse.domhi - crse.domlo + 1
        s = int(rng.integers(block[0], block[1]))  # coarse cells per block side
        nb = cn // s
        if np.any(nb < 3):
            break
        U = np.zeros((int(nb[2]), int(nb[1]), int(nb[0])), dtype=bool)
        # rectangles around where the coarse level has data
        where = np.argwhere(occ)
        clo, chi = where.min(axis=0)[::-1] // s, where.max(axis=0)[::-1] // s
        for _ in range(int(rng.integers(nrect[0], nrect[1]))):
            a = np.array([rng.integers(clo[d], chi[d] + 1) for d in range(3)])
            e = np.array([rng.integers(1, max(2, (chi[d] - clo[d] + 1) // 2 + 1)) for d in range(3)])
            b = np.minimum(a + e, nb) - 1
            U[a[2]:b[2] + 1, a[1]:b[1] + 1, a[0]:b[0] + 1] = True
        U = _nested_blocks(U, s, occ, per)
        if not U.any():
            break
        boxes = blocks_to_boxes(U, 2 * s, max_blocks, rng)
        domhi_f = 2 * (crse.domhi + 1) - 1
        levels.append(Level(boxes, (0, 0, 0), domhi_f, per, np.zeros(3), np.ones(3)))
    return Hierarchy(levels, 2)


def _erode_blocks(occ: np.ndarray, is_per) -> np.ndarray:
    """blocks whose 26 neighbour blocks are all occupied (beyond a wall counts as occupied, periodic directions wrap)"""
    out = occ.copy()
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx == dy == dz == 0:
                    continue
                sh = occ
                for ax, (d, pd) in enumerate(((dz, is_per[2]), (dy, is_per[1]), (dx, is_per[0]))):
                    if d == 0:
                        continue
                    sh = np.roll(sh, d, axis=ax)
                    if not pd:  # the slice that wrapped around came from beyond a wall
                        idx = [slice(None)] * 3
                        idx[ax] = slice(0, 1) if d == 1 else slice(-1, None)
                        sh = sh.copy()
                        sh[tuple(idx)] = True
                out &= sh
    return out


def tagged_hierarchy(base_n: int, nlev: int, fn, bf: int = 16, max_box: int = 128, frac=(0.30, 0.30), base_box: int = 128, is_per=(1, 1, 0),
                     seed: int = 0, random_sizes: bool = False) -> Hierarchy:
    """an irregular hierarchy the way a Pele run makes one: level l + 1 covers the blocks of bf coarse cells (blocking factor
    2 bf on the fine level) where |grad fn| at the block centre is among the largest `frac[l-1]` of the level's blocks,
    properly nested (one block of buffer to the edge of the coarse level), merged into boxes of 2 bf .. max_box cells per
    side (greedy, as large as the tagged region allows; random_sizes: random limits per box instead).  Works at block
    granularity, so a 512^3 base with three levels takes a fraction of a second."""
    rng = np.random.default_rng(seed)
    per = np.asarray(is_per)
    n = np.full(3, base_n, dtype=np.int64)
    levels = [Level(chop_box((0, 0, 0), n - 1, base_box), (0, 0, 0), n - 1, per, np.zeros(3), np.ones(3))]
    nb = base_n // bf
    occ = np.ones((nb, nb, nb), dtype=bool)  # valid region of the current level in blocks of bf of ITS cells
    cn = base_n
    for l in range(1, nlev):
        h = 1.0 / float(cn)
        c = (np.arange(nb) + 0.5) * bf * h
        X, Y, Z = c[None, None, :], c[None, :, None], c[:, None, None]
        e = 0.5 * h
        g = np.sqrt((fn(X + e, Y, Z) - fn(X - e, Y, Z)) ** 2 + (fn(X, Y + e, Z) - fn(X, Y - e, Z)) ** 2 + (fn(X, Y, Z + e) - fn(X, Y, Z - e)) ** 2)
        g = np.broadcast_to(g, (nb, nb, nb)).copy()
        g[~occ] = -1.0
        k = max(1, int(frac[min(l - 1, len(frac) - 1)] * occ.sum()))
        thr = np.sort(g.ravel())[-k]
        U = (g >= thr) & _erode_blocks(occ, per)
        if not U.any():
            break
        boxes = blocks_to_boxes(U, 2 * bf, max(1, max_box // (2 * bf)), rng if random_sizes else None)
        cn *= 2
        levels.append(Level(boxes, (0, 0, 0), (cn - 1,) * 3, per, np.zeros(3), np.ones(3)))
        occ = np.repeat(np.repeat(np.repeat(U, 2, axis=0), 2, axis=1), 2, axis=2)
        nb *= 2
    return Hierarchy(levels, 2)


# ----------------------------------------------------------------------------- internal re-tiling (pa_level_retile)
def retile_level(level: Level, max_size=(128, 128, 128), min_thick: int = 3) -> Level:
    """The level on the BoxArray pa_level_retile returns for it: the same cells in fewer, larger boxes (host arithmetic
    in the library, no GPU).  max_size: cells per direction (x, y, z)."""
    import ctypes as C
    from . import capi
    lib = capi.load_library()
    b = np.ascontiguousarray(level.boxes, dtype=np.int32)
    cap = 4 * level.nboxes + 16
    out = np.zeros((cap, 6), dtype=np.int32)
    mx = (C.c_int32 * 3)(*[int(v) for v in np.broadcast_to(np.asarray(max_size), (3,))])
    n = lib.pa_level_retile(level.nboxes, b.ctypes.data_as(C.POINTER(C.c_int32)), mx, int(min_thick), out.ctypes.data_as(C.POINTER(C.c_int32)), cap)
    if n < 0:
        raise ValueError("pa_level_retile: bad arguments")
    return Level(out[:n].copy(), level.domlo, level.domhi, level.is_per, level.prob_lo, level.prob_hi)


def regrid_copy(src: MultiFab, dst: MultiFab, scomp: int = 0, dcomp: int = 0, ncomp: int | None = None) -> None:
    """valid cells of src -> valid cells of dst wherever their boxes intersect (two tilings of one cell set: what the tools'
    read_comp / write_plotfile do between the file's BoxArray and the re-tiled one)."""
    nc = src.ncomp - scomp if ncomp is None else ncomp
    sb, db = src.level.boxes.astype(np.int64), dst.level.boxes.astype(np.int64)
    for a in range(len(db)):
        lo = np.maximum(db[a, :3], sb[:, :3])
        hi = np.minimum(db[a, 3:], sb[:, 3:])
        dv = None
        for b in np.nonzero((lo <= hi).all(axis=1))[0]:
            if dv is None:
                dv = dst.valid(a)
            sv = src.valid(int(b))
            l, h = lo[b], hi[b]
            ds = tuple(slice(int(l[d] - db[a, d]), int(h[d] - db[a, d]) + 1) for d in (2, 1, 0))
            ss = tuple(slice(int(l[d] - sb[b, d]), int(h[d] - sb[b, d]) + 1) for d in (2, 1, 0))
            dv[(slice(dcomp, dcomp + nc),) + ds] = sv[(slice(scomp, scomp + nc),) + ss]


def retile_hierarchy(H: Hierarchy, max_size=None, min_thick: int = 3, nranks: int = 1) -> Hierarchy:
    """every level re-tiled with the limits the tools use (pa_hierarchy_retile_limits[_ranks]) or with max_size"""
    import ctypes as C
    from . import capi
    if max_size is None:
        lib = capi.load_library()
        pi32 = C.POINTER(C.c_int32)
        bs = [np.ascontiguousarray(lv.boxes, dtype=np.int32) for lv in H.levels]
        nb = (C.c_int32 * H.nlev)(*[lv.nboxes for lv in H.levels])
        ptrs = (pi32 * H.nlev)(*[b.ctypes.data_as(pi32) for b in bs])
        mx = (C.c_int32 * 3)()
        if lib.pa_hierarchy_retile_limits_ranks(H.nlev, nb, ptrs, int(min_thick), int(nranks), mx) != 0:
            raise ValueError("pa_hierarchy_retile_limits: bad arguments")
        max_size = tuple(mx)
    return Hierarchy([retile_level(lv, max_size, min_thick) for lv in H.levels], H.ref_ratio)
