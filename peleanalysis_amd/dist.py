"""Multi-GPU sharding of a level's BoxArray (one process per GPU) and the cross-rank half of
FillBoundary.

The reference distributes boxes with AMReX `DistributionMapping(ba)` over MPI ranks and exchanges
ghost cells with point-to-point messages inside `FabArray::FillBoundary` (grad.cpp:162,169;
SURVEY 2.1).  Here each rank owns a subset of the boxes; for every (destination box, source box,
periodic shift) pair that lives on two different ranks both sides derive the same region list in
the same order, the sender packs its regions into ONE buffer per peer
(`pa_pack_regions`), the buffers travel with `torch.distributed` point-to-point ops (backend
"nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests) and the receiver scatters them
into its ghost cells (`pa_unpack_regions`).  No other collective is on the data path.
"""
from __future__ import annotations

import dataclasses
import itertools
from typing import Dict, List, Optional, Sequence

import numpy as np

from .hierarchy import Hierarchy, Level, MultiFab, chop_box


@dataclasses.dataclass
class ExchangePlan:
    """regions7 rows = [local box index, lo0, lo1, lo2, hi0, hi1, hi2] in that box's index space"""
    send: Dict[int, np.ndarray]  # peer -> (n,7) int32: what to pack for that peer (source regions)
    recv: Dict[int, np.ndarray]  # peer -> (n,7) int32: where that peer's buffer goes (ghost regions)

    def size(self, regs: np.ndarray, ncomp: int) -> int:
        if len(regs) == 0:
            return 0
        n = regs[:, 4:7].astype(np.int64) - regs[:, 1:4] + 1
        return int(ncomp * np.prod(n, axis=1).sum())


def build_plan(boxes: np.ndarray, owner: Sequence[int], domlo, domhi, is_per, rank: int, ng: int) -> ExchangePlan:
    """Region lists for `rank`.  Pairs are enumerated in (dst box, src box, shift) order on every rank,
    so the sender's pack order equals the receiver's unpack order."""
    boxes = np.asarray(boxes, dtype=np.int64).reshape(-1, 6)
    owner = np.asarray(owner)
    domlo, domhi = np.asarray(domlo, dtype=np.int64), np.asarray(domhi, dtype=np.int64)
    length = domhi - domlo + 1
    local_index = {int(g): i for i, g in enumerate(np.nonzero(owner == rank)[0])}
    shifts = [np.array(s) * length for s in itertools.product(*[((-1, 0, 1) if is_per[d] else (0,)) for d in range(3)])]
    send: Dict[int, List[List[int]]] = {}
    recv: Dict[int, List[List[int]]] = {}
    for d in range(len(boxes)):
        glo, ghi = boxes[d, :3] - ng, boxes[d, 3:] + ng
        for s in range(len(boxes)):
            if owner[d] == owner[s] or (owner[d] != rank and owner[s] != rank):
                continue
            for sh in shifts:
                lo = np.maximum(glo, boxes[s, :3] + sh)
                hi = np.minimum(ghi, boxes[s, 3:] + sh)
                if np.any(lo > hi):
                    continue
                if owner[d] == rank:  # I receive ghost cells of my box d from owner[s]
                    recv.setdefault(int(owner[s]), []).append([local_index[d], *lo, *hi])
                else:                 # I send valid cells of my box s (un-shifted) to owner[d]
                    send.setdefault(int(owner[d]), []).append([local_index[s], *(lo - sh), *(hi - sh)])
    to_arr = lambda d: {p: np.asarray(v, dtype=np.int32).reshape(-1, 7) for p, v in d.items()}
    return ExchangePlan(to_arr(send), to_arr(recv))


# ------------------------------------------------------------------------------- host backend
def host_pack(mf: MultiFab, comp: int, ncomp: int, regs: np.ndarray) -> np.ndarray:
    out = []
    for r in regs:
        b = int(r[0])
        f = mf.fab(b)
        o = mf.level.boxes[b, :3] - mf.ng
        out.append(f[comp:comp + ncomp, r[3] - o[2]:r[6] - o[2] + 1, r[2] - o[1]:r[5] - o[1] + 1, r[1] - o[0]:r[4] - o[0] + 1].ravel())
    return np.concatenate(out) if out else np.zeros(0)


def host_unpack(mf: MultiFab, comp: int, ncomp: int, regs: np.ndarray, buf: np.ndarray) -> None:
    p = 0
    for r in regs:
        b = int(r[0])
        f = mf.fab(b)
        o = mf.level.boxes[b, :3] - mf.ng
        shp = (ncomp, r[6] - r[3] + 1, r[5] - r[2] + 1, r[4] - r[1] + 1)
        n = int(np.prod(shp))
        f[comp:comp + ncomp, r[3] - o[2]:r[6] - o[2] + 1, r[2] - o[1]:r[5] - o[1] + 1, r[1] - o[0]:r[4] - o[0] + 1] = buf[p:p + n].reshape(shp)
        p += n


def exchange(plan: ExchangePlan, ncomp: int, pack, unpack, make_buffer, device=None) -> None:
    """One ghost exchange: pack(regs) -> 1-D float64 torch tensor, unpack(regs, tensor).  Point-to-point
    sends/recvs batched per peer (`batch_isend_irecv`: grouped ncclSend/ncclRecv on RCCL)."""
    import torch.distributed as dist
    peers = sorted(set(plan.send) | set(plan.recv))
    if not peers:
        return
    ops, rbufs = [], {}
    for p in peers:
        if p in plan.send:
            ops.append(dist.P2POp(dist.isend, pack(plan.send[p]), p))
        if p in plan.recv:
            rbufs[p] = make_buffer(plan.size(plan.recv[p], ncomp))
            ops.append(dist.P2POp(dist.irecv, rbufs[p], p))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if device is not None:
        import torch
        torch.cuda.synchronize(device)
    for p, t in rbufs.items():
        unpack(plan.recv[p], t)


def exchange_host(plan: ExchangePlan, mf: MultiFab, comp: int, ncomp: int) -> None:
    import torch
    exchange(plan, ncomp,
             pack=lambda regs: torch.from_numpy(host_pack(mf, comp, ncomp, regs).copy()),
             unpack=lambda regs, t: host_unpack(mf, comp, ncomp, regs, t.numpy()),
             make_buffer=lambda n: torch.empty(n, dtype=torch.float64))


def exchange_device(plan: ExchangePlan, ctx, dmf, comp: int, ncomp: int, device) -> None:
    """device path: HIP pack/unpack kernels around RCCL point-to-point"""
    import ctypes as C

    import torch

    def pack(regs):
        t = torch.empty(plan.size(regs, ncomp), dtype=torch.float64, device=device)
        r = np.ascontiguousarray(regs, dtype=np.int32)
        ctx.check(ctx.lib.pa_pack_regions(ctx.h, dmf.h, comp, ncomp, len(r), r.ctypes.data_as(C.POINTER(C.c_int32)), t.data_ptr()))
        return t

    def unpack(regs, t):
        r = np.ascontiguousarray(regs, dtype=np.int32)
        ctx.check(ctx.lib.pa_unpack_regions(ctx.h, dmf.h, comp, ncomp, len(r), r.ctypes.data_as(C.POINTER(C.c_int32)), t.data_ptr()))

    exchange(plan, ncomp, pack, unpack, lambda n: torch.empty(n, dtype=torch.float64, device=device), device=device)


def exchange_device_staged(plan: ExchangePlan, ctx, dmf, comp: int, ncomp: int, device, group=None) -> None:
    """same region lists and HIP pack/unpack kernels, but the packed buffers travel through host
    memory and a gloo group (fallback when RCCL point-to-point is unavailable; single-GPU tests)"""
    import ctypes as C

    import torch
    import torch.distributed as dist

    peers = sorted(set(plan.send) | set(plan.recv))
    if not peers:
        return
    ops, rb = [], {}
    for p in peers:
        if p in plan.send:
            regs = np.ascontiguousarray(plan.send[p], dtype=np.int32)
            t = torch.empty(plan.size(regs, ncomp), dtype=torch.float64, device=device)
            ctx.check(ctx.lib.pa_pack_regions(ctx.h, dmf.h, comp, ncomp, len(regs), regs.ctypes.data_as(C.POINTER(C.c_int32)), t.data_ptr()))
            ops.append(dist.P2POp(dist.isend, t.cpu(), p, group=group))
        if p in plan.recv:
            rb[p] = torch.empty(plan.size(plan.recv[p], ncomp), dtype=torch.float64)
            ops.append(dist.P2POp(dist.irecv, rb[p], p, group=group))
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    for p, t in rb.items():
        d = t.to(device)
        torch.cuda.synchronize(device)
        regs = np.ascontiguousarray(plan.recv[p], dtype=np.int32)
        ctx.check(ctx.lib.pa_unpack_regions(ctx.h, dmf.h, comp, ncomp, len(regs), regs.ctypes.data_as(C.POINTER(C.c_int32)), d.data_ptr()))


# ------------------------------------------------------------------------------- slab decomposition
@dataclasses.dataclass
class RankLevels:
    glob: Hierarchy                 # the global hierarchy (all boxes of all ranks)
    owner: List[np.ndarray]         # per level: owner rank of every global box
    local: Hierarchy                # this rank's boxes
    remote: List[np.ndarray]        # per level: boxes owned by other ranks
    plans: List[ExchangePlan]


def slab_hierarchy(base_n: int, nlev: int, box: int, nranks: int, rank: int, ng: int, is_per=(1, 1, 0)) -> RankLevels:
    """Weak-scaling workload: `nranks` copies of the nested 3-level hierarchy side by side in x
    (global base level (nranks*base_n) x base_n x base_n, periodic in x); rank r owns slab r on every level.
    The only cross-rank ghost cells are on the level-0 slab faces."""
    glob_levels, owners, loc_levels, remotes, plans = [], [], [], [], []
    lo = np.zeros(3, dtype=np.int64)
    hi = np.full(3, base_n - 1, dtype=np.int64)
    n0 = base_n
    for l in range(nlev):
        boxes, own = [], []
        for r in range(nranks):
            off = np.array([r * n0, 0, 0])
            boxes.append(chop_box(lo + off, hi + off, box))
            own.append(np.full(len(boxes[-1]), r))
        boxes, own = np.vstack(boxes), np.concatenate(own)
        domlo, domhi = np.zeros(3, dtype=np.int64), np.array([nranks * n0 - 1, n0 - 1, n0 - 1])
        prob_hi = (float(nranks), 1.0, 1.0)
        glob_levels.append(Level(boxes, domlo, domhi, is_per, (0.0, 0.0, 0.0), prob_hi))
        owners.append(own)
        loc_levels.append(Level(boxes[own == rank], domlo, domhi, is_per, (0.0, 0.0, 0.0), prob_hi))
        remotes.append(boxes[own != rank])
        plans.append(build_plan(boxes, own, domlo, domhi, is_per, rank, ng))
        n = hi - lo + 1
        clo = lo + n // 4
        chi = clo + n // 2 - 1
        lo, hi = 2 * clo, 2 * chi + 1
        n0 *= 2
    return RankLevels(Hierarchy(glob_levels, 2), owners, Hierarchy(loc_levels, 2), remotes, plans)
