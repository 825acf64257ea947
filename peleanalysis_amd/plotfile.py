"""AMReX plotfile (HyperCLaw-V1.1) reader / writer in numpy -- host-side format code used by the
tests and the synthetic-data generator (Docs/source/data.rst:19-32 of the
reference).  The C++ tool drivers in tools/ carry their own implementation of the same format.

Layout: <plt>/Header, <plt>/Level_<n>/Cell_H, <plt>/Level_<n>/Cell_D_00000 (one data file per level).
FAB payload: ASCII header line + ncomp*nx*ny*nz little-endian doubles, component-major, x fastest.
"""
from __future__ import annotations

import os
import re
from typing import List, Sequence

import numpy as np

from .hierarchy import Hierarchy, Level, MultiFab

FAB_DESC = "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))"


def _box_str(lo, hi, dim=3):
    if dim == 2:
        return "((%d,%d) (%d,%d) (0,0))" % (lo[0], lo[1], hi[0], hi[1])
    return "((%d,%d,%d) (%d,%d,%d) (0,0,0))" % (lo[0], lo[1], lo[2], hi[0], hi[1], hi[2])


def write_plotfile(path: str, H: Hierarchy, mfs: Sequence[MultiFab], names: Sequence[str], time: float = 0.0,
                   level_steps: Sequence[int] | None = None, dim: int = 3, precision: int = 64) -> None:
    """WriteMultiLevelPlotfile restated (valid cells only, one Cell_D file per level).  dim = 2: the hierarchy is one
    plane of cells (k = 0) and the file is what a 2-D AMReX code writes (2 entries per index / coordinate tuple)."""
    if dim == 2:
        assert all((lv.boxes[:, 2] == 0).all() and (lv.boxes[:, 5] == 0).all() for lv in H.levels)
    # precision = 32: FABio::FAB_NATIVE_32 (what AmrLevel-based codes write by default): IEEE floats, little endian
    desc, dt = (FAB_DESC, "<f8") if precision == 64 else ("FAB ((8, (32 8 23 0 1 9 0 127)),(4, (4 3 2 1)))", "<f4")

    nlev = H.nlev
    ncomp = len(names)
    level_steps = list(level_steps) if level_steps is not None else [0] * nlev
    os.makedirs(path, exist_ok=True)
    L0 = H.levels[0]
    with open(os.path.join(path, "Header"), "w") as f:
        f.write("HyperCLaw-V1.1\n%d\n" % ncomp)
        for n in names:
            f.write(n + "\n")
        f.write("%d\n%.17g\n%d\n" % (dim, time, nlev - 1))
        f.write(" ".join("%.17g" % v for v in L0.prob_lo[:dim]) + " \n")
        f.write(" ".join("%.17g" % v for v in L0.prob_hi[:dim]) + " \n")
        f.write(" ".join(str(H.ref_ratio) for _ in range(nlev - 1)) + " \n")
        f.write(" ".join(_box_str(lv.domlo, lv.domhi, dim) for lv in H.levels) + " \n")
        f.write(" ".join(str(s) for s in level_steps) + " \n")
        for lv in H.levels:
            f.write(" ".join("%.17g" % v for v in lv.dx[:dim]) + " \n")
        f.write("0\n0\n")
        for l, lv in enumerate(H.levels):
            f.write("%d %d %.17g\n%d\n" % (l, lv.nboxes, time, level_steps[l]))
            dx = lv.dx
            for b in range(lv.nboxes):
                for d in range(dim):
                    f.write("%.17g %.17g\n" % (lv.prob_lo[d] + lv.boxes[b, d] * dx[d], lv.prob_lo[d] + (lv.boxes[b, 3 + d] + 1) * dx[d]))
            f.write("Level_%d/Cell\n" % l)
    for l, lv in enumerate(H.levels):
        d = os.path.join(path, "Level_%d" % l)
        os.makedirs(d, exist_ok=True)
        offs, mins, maxs = [], [], []
        with open(os.path.join(d, "Cell_D_00000"), "wb") as f:
            for b in range(lv.nboxes):
                offs.append(f.tell())
                v = np.ascontiguousarray(mfs[l].valid(b)[:ncomp], dtype=dt)
                f.write((desc + _box_str(lv.boxes[b, :3], lv.boxes[b, 3:], dim) + " %d\n" % ncomp).encode())
                f.write(v.tobytes())
                mins.append(v.reshape(ncomp, -1).min(axis=1))
                maxs.append(v.reshape(ncomp, -1).max(axis=1))
        with open(os.path.join(d, "Cell_H"), "w") as f:
            f.write("1\n1\n%d\n0\n(%d 0\n" % (ncomp, lv.nboxes))
            for b in range(lv.nboxes):
                f.write(_box_str(lv.boxes[b, :3], lv.boxes[b, 3:], dim) + "\n")
            f.write(")\n%d\n" % lv.nboxes)
            for b in range(lv.nboxes):
                f.write("FabOnDisk: Cell_D_00000 %d\n" % offs[b])
            f.write("\n%d,%d\n" % (lv.nboxes, ncomp))
            for m in mins:
                f.write("".join("%.17g," % x for x in m) + "\n")
            f.write("\n%d,%d\n" % (lv.nboxes, ncomp))
            for m in maxs:
                f.write("".join("%.17g," % x for x in m) + "\n")


_BOX_RE = re.compile(r"\(\((-?\d+),(-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+),(-?\d+)\)\)")
_BOX2_RE = re.compile(r"\(\((-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+)\)\)")


class PlotfileData:
    def __init__(self, names, time, H, mfs, level_steps):
        self.names, self.time, self.hier, self.mfs, self.level_steps = names, time, H, mfs, level_steps


def read_plotfile(path: str, is_per=(0, 0, 0)) -> PlotfileData:
    """Read every component of every level (valid cells).  The Header stores no periodicity: it comes
    from the caller (the tools' is_per key), as in the reference."""
    with open(os.path.join(path, "Header")) as f:
        lines = f.read().split("\n")
    it = iter(lines)
    version = next(it).strip()
    if not version.startswith("HyperCLaw") and not version.startswith("NavierStokes"):
        raise ValueError("not a plotfile Header: " + version)
    ncomp = int(next(it))
    names = [next(it).strip() for _ in range(ncomp)]
    dim = int(next(it))
    if dim not in (2, 3):
        raise ValueError("only 2-D and 3-D plotfiles are supported")
    box_re = _BOX_RE if dim == 3 else _BOX2_RE

    def box6(m):  # a 2-D box is the plane k = 0
        v = [int(x) for x in m]
        return v[:6] if dim == 3 else [v[0], v[1], 0, v[2], v[3], 0]
    time = float(next(it))
    finest = int(next(it))
    nlev = finest + 1
    prob_lo = np.array([float(x) for x in next(it).split()] + ([0.0] if dim == 2 else []))
    prob_hi = np.array([float(x) for x in next(it).split()] + ([1.0] if dim == 2 else []))
    rr = [int(x) for x in re.findall(r"-?\d+", next(it))]
    if len(set(rr[:nlev - 1])) > 1:
        raise ValueError("levels with different refinement ratios are not supported by this reader")
    ratio = rr[0] if nlev > 1 and rr else 2
    doms = box_re.findall(next(it))
    steps = [int(x) for x in next(it).split()]
    for _ in range(nlev):
        next(it)  # dx lines (recomputed from prob size / cells like amrex::Geometry)
    next(it)  # coord sys
    next(it)  # boundary width
    levels, mfs = [], []
    for l in range(nlev):
        hdr = next(it).split()
        ngrids = int(hdr[1])
        next(it)
        for _ in range(dim * ngrids):
            next(it)
        rel = next(it).strip()
        d = box6(doms[l])
        cell_h = os.path.join(path, rel + "_H")
        with open(cell_h) as f:
            txt = f.read()
        head, rest = txt.split("(", 1)
        hl = head.split()
        nc_file = int(hl[2])
        blk = rest[: rest.index("FabOnDisk")]  # the BoxArray: "(n 0\n((lo) (hi) (0,0,0))\n...)\n n"
        boxes = np.array([box6(m) for m in box_re.findall(blk)], dtype=np.int32)
        assert len(boxes) == ngrids and nc_file == ncomp
        fods = re.findall(r"FabOnDisk:\s*(\S+)\s+(\d+)", rest)
        lv = Level(boxes, d[0:3], d[3:6], is_per, prob_lo, prob_hi)
        mf = MultiFab(lv, ncomp, 0)
        for b, (fn, off) in enumerate(fods):
            with open(os.path.join(os.path.dirname(cell_h), fn), "rb") as f:
                f.seek(int(off))
                line = f.readline().decode()
                m = box6(box_re.findall(line)[-1])
                flo, fhi = np.array(m[0:3]), np.array(m[3:6])
                nc = int(line.strip().split()[-1])
                n = fhi - flo + 1
                if "(8, (8 7 6 5 4 3 2 1))" in line:
                    data = np.frombuffer(f.read(8 * nc * int(n.prod())), dtype="<f8").reshape(nc, n[2], n[1], n[0])
                elif "(4, (4 3 2 1))" in line:  # FAB_NATIVE_32: converted to double like AmrData does
                    data = np.frombuffer(f.read(4 * nc * int(n.prod())), dtype="<f4").astype(np.float64).reshape(nc, n[2], n[1], n[0])
                else:
                    raise ValueError("unsupported FAB RealDescriptor: " + line[:80])
            g = boxes[b, :3] - flo  # file fabs may carry ghost cells
            nz, ny, nx = lv.box_shape(b)
            mf.valid(b)[:] = data[:ncomp, g[2]:g[2] + nz, g[1]:g[1] + ny, g[0]:g[0] + nx]
        levels.append(lv)
        mfs.append(mf)
    return PlotfileData(names, time, Hierarchy(levels, ratio), mfs, steps)


def read_mef(path: str):
    """MEF surface file (isosurface.cpp:2097-2134; mef2vtk.py:28-50): label, variable names,
    'nElts nodesPerElt', FAB of Box (0..N-1,0,0) x ncomp stored node-major, then 1-based int32 faces."""
    with open(path, "rb") as f:
        label = f.readline().decode().strip()
        names = f.readline().decode().split()
        nelts, npe = [int(x) for x in f.readline().decode().split()]
        line = f.readline().decode()
        m3 = _BOX_RE.findall(line)
        if m3:
            nnodes = int(m3[-1][3]) - int(m3[-1][0]) + 1
        else:  # 2-D build: Box (0..N-1, 0)
            m2 = re.findall(r"\(\((-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+)\)\s*\((-?\d+),(-?\d+)\)\)", line)[-1]
            nnodes = int(m2[2]) - int(m2[0]) + 1
        nc = int(line.strip().split()[-1])
        nodes = np.frombuffer(f.read(8 * nnodes * nc), dtype="<f8").reshape(nnodes, nc)
        faces = np.frombuffer(f.read(4 * nelts * npe), dtype="<i4").reshape(nelts, npe)
        rest = f.read()
    assert rest == b"" and nc == len(names)
    return label, names, nodes, faces
