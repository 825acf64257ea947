"""One AMR hierarchy sharded over ranks (one process per GPU): the host-side mirror of the multi-GPU part of the C ABI.

The reference distributes the boxes of every level with AMReX `DistributionMapping(ba)` over MPI ranks and moves
ghost data with point-to-point messages inside `FabArray::FillBoundary`, `FillPatchTwoLevels` and the MLMG boundary
registers (grad.cpp:162,169,212; curvature.cpp:289,443-445,514-518).  The library does all of that
itself once a level is created with its whole BoxArray + owner map (`capi.DevLevel(ctx, level, owner, rank, nranks)`)
and the context has a transport:

* `init_rccl(ctx)`: the built-in one -- RCCL over xGMI, grouped ncclSend/ncclRecv issued by the library on its own
  stream; torch.distributed is only used here to broadcast the 128-byte communicator id;
* `GlooComm`: a `pa_comm` whose callbacks stage the packed buffers through host memory and a gloo group -- what the
  tests use when several ranks share one GPU (RCCL wants one GPU per rank), and bench.py's fallback.

Everything else in this file is host arithmetic for tests: the owner map (`pa_distribution_map`), the region lists
behind the exchanges (`pa_plan_*`), and numpy pack / unpack.
"""
from __future__ import annotations

import ctypes as C
import traceback
from typing import Dict, List, Sequence

import numpy as np

from . import capi
from .hierarchy import Hierarchy, Level, MultiFab


def _pi32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _i3(v):
    return (C.c_int32 * 3)(*[int(x) for x in v])


def distribution_map(boxes: np.ndarray, nranks: int) -> np.ndarray:
    """DistributionMapping(ba) restated (pa_distribution_map): Morton order of the boxes' low corners cut into nranks
    contiguous pieces of equal cell count.  Host arithmetic (no GPU needed)."""
    lib = capi.load_library()
    b = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 6)
    owner = np.zeros(len(b), dtype=np.int32)
    if lib.pa_distribution_map(len(b), _pi32(b), int(nranks), _pi32(owner)) != 0:
        raise ValueError("pa_distribution_map: bad arguments")
    return owner


def shard(H: Hierarchy, nranks: int) -> List[np.ndarray]:
    """owner map of every level of a hierarchy"""
    return [distribution_map(lv.boxes, nranks) for lv in H.levels]


def _rows(fn, *args) -> np.ndarray:
    n = fn(*args, None, 0)
    if n < 0:
        raise ValueError("plan: bad arguments")
    rows = np.zeros((max(n, 1), 9), dtype=np.int32)
    fn(*args, _pi32(rows), n)
    return rows[:n]


def plan_fill_boundary(level: Level, owner: Sequence[int], rank: int, ng: int) -> np.ndarray:
    """rows {kind (0 send, 1 recv), peer, global box, lo[3], hi[3]} of the cross-rank half of FillBoundary(ng) for `rank`"""
    lib = capi.load_library()
    b = np.ascontiguousarray(level.boxes, dtype=np.int32)
    o = np.ascontiguousarray(owner, dtype=np.int32)
    return _rows(lib.pa_plan_fill_boundary, level.nboxes, _pi32(b), _pi32(o), int(rank), _i3(level.domlo), _i3(level.domhi), _i3(level.is_per), int(ng))


def plan_coarse_source(fine: Level, fowner, crse: Level, cowner, rank: int, mode: int = 0, ng: int = 0, halo: int = 0) -> np.ndarray:
    """rows {kind, peer, global coarse box, lo[3], hi[3]}: kind 2 = a piece of the coarse level `rank` keeps a copy of (peer =
    owner of the coarse box), in the order of the rank's coarse-source BoxArray; kind 0 = what `rank` sends to `peer`"""
    lib = capi.load_library()
    fb, cb = np.ascontiguousarray(fine.boxes, dtype=np.int32), np.ascontiguousarray(crse.boxes, dtype=np.int32)
    fo, co = np.ascontiguousarray(fowner, dtype=np.int32), np.ascontiguousarray(cowner, dtype=np.int32)
    return _rows(lib.pa_plan_coarse_source, fine.nboxes, _pi32(fb), _pi32(fo), _i3(fine.domlo), _i3(fine.domhi), crse.nboxes, _pi32(cb), _pi32(co),
                 _i3(crse.domlo), _i3(crse.domhi), _i3(fine.is_per), int(rank), int(mode), int(ng), int(halo))


def plan_restriction(fine: Level, fowner, crse: Level, cowner, rank: int, which: int, ratio: int = 2) -> np.ndarray:
    """rows {kind, peer, global box, lo[3], hi[3]} of one restriction plan of the distributed smoothing solve (which = 0: child
    averages, 1 + 2 * dir + side: the flux register of that face orientation); kind 0 send (fine box), 1 receive (coarse box),
    3 / 4 source / destination of a same-rank copy"""
    lib = capi.load_library()
    fb, cb = np.ascontiguousarray(fine.boxes, dtype=np.int32), np.ascontiguousarray(crse.boxes, dtype=np.int32)
    fo, co = np.ascontiguousarray(fowner, dtype=np.int32), np.ascontiguousarray(cowner, dtype=np.int32)
    return _rows(lib.pa_plan_restriction, fine.nboxes, _pi32(fb), _pi32(fo), _i3(fine.domlo), _i3(fine.domhi), crse.nboxes, _pi32(cb), _pi32(co),
                 _i3(crse.domlo), _i3(crse.domhi), _i3(fine.is_per), int(rank), int(ratio), int(which))


# ------------------------------------------------------------------------------- numpy pack / unpack (CPU-tier tests)
def host_region(mf: MultiFab, b: int, lo, hi, comp: int, ncomp: int) -> np.ndarray:
    """view of components comp..comp+ncomp of box b (local index) over the index-space region lo..hi (ghost cells allowed)"""
    o = mf.level.boxes[b, :3] - mf.ng
    return mf.fab(b)[comp:comp + ncomp, lo[2] - o[2]:hi[2] - o[2] + 1, lo[1] - o[1]:hi[1] - o[1] + 1, lo[0] - o[0]:hi[0] - o[0] + 1]


def host_exchange(rows: np.ndarray, glocal: Dict[int, int], src: MultiFab, dst: MultiFab, comp: int, ncomp: int, recv_kind: int = 1, recv_box=None) -> None:
    """the exchange a plan describes, through torch.distributed point-to-point ops on host tensors: one message per peer,
    regions in row order on both sides.  glocal: global box index -> local index in src (send rows); recv rows address dst
    through glocal too (FillBoundary) or through recv_box(row index among the rank's kind-2 rows) (coarse source)."""
    import torch
    import torch.distributed as dist
    me = dist.get_rank()
    send: Dict[int, list] = {}
    recv: Dict[int, list] = {}
    npiece = 0
    for r in rows:
        kind, peer, box = int(r[0]), int(r[1]), int(r[2])
        if kind == 0:
            send.setdefault(peer, []).append(host_region(src, glocal[box], r[3:6], r[6:9], comp, ncomp).ravel().copy())
        elif kind == recv_kind:
            b = glocal[box] if recv_box is None else recv_box(npiece)
            npiece += 1
            if peer == me:  # a piece cut from a coarse box this rank owns itself: local copy
                host_region(dst, b, r[3:6], r[6:9], 0 if recv_box else comp, ncomp)[...] = host_region(src, glocal[box], r[3:6], r[6:9], comp, ncomp)
            else:
                recv.setdefault(peer, []).append((b, r[3:6].copy(), r[6:9].copy()))
    ops, rb = [], {}
    for p in sorted(set(send) | set(recv)):
        if p in send:
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.concatenate(send[p])), p))
        if p in recv:
            n = sum(int(np.prod(hi - lo + 1)) * ncomp for _, lo, hi in recv[p])
            rb[p] = torch.empty(n, dtype=torch.float64)
            ops.append(dist.P2POp(dist.irecv, rb[p], p))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    for p, t in rb.items():
        a, at = t.numpy(), 0
        for b, lo, hi in recv[p]:
            v = host_region(dst, b, lo, hi, 0 if recv_box else comp, ncomp)
            v[...] = a[at:at + v.size].reshape(v.shape)
            at += v.size


# ------------------------------------------------------------------------------- transports
def broadcast_rccl_id(ctx: "capi.Context", group=None) -> bytes:
    """rank 0 creates the RCCL communicator id, torch.distributed (any backend) broadcasts it.  Every rank always takes part in
    the broadcast -- rank 0 sends an error marker when it could not make the id -- so that a failure on rank 0 raises on ALL
    ranks and their next collective still lines up.  A torch.distributed collective: call it from the thread that issues the
    process group's other collectives."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    msg = [None, None]
    if rank == 0:
        try:
            msg[0] = ctx.rccl_unique_id()
        except Exception as e:  # librccl missing, ncclGetUniqueId failed
            msg[1] = repr(e)[:300]
    dist.broadcast_object_list(msg, src=0, group=group)
    if msg[1] is not None or msg[0] is None:
        raise RuntimeError("RCCL unique id not available on rank 0: " + str(msg[1]))
    return msg[0]


def init_rccl(ctx: "capi.Context", group=None) -> None:
    """built-in RCCL transport: broadcast_rccl_id + pa_ctx_init_rccl (ncclCommInitRank blocks until every rank has joined)"""
    import torch.distributed as dist
    uid = broadcast_rccl_id(ctx, group)
    ctx.init_rccl(dist.get_world_size(group), dist.get_rank(group), uid)


class GlooComm:
    """pa_comm over a torch.distributed group with host tensors (gloo): packed device buffers are staged through host
    memory.  For ranks that share a GPU (tests, rehearsals) and as the fallback when RCCL point-to-point is unavailable."""

    def __init__(self, ctx: "capi.Context", group=None):
        import torch
        import torch.distributed as dist
        self.ctx, self.group, self.torch, self.dist = ctx, group, torch, dist
        self.rank, self.nranks = dist.get_rank(group), dist.get_world_size(group)
        self.nexchange = 0
        self.bytes_sent = 0
        self._ex = capi.EXCHANGE_FN(self._exchange)
        self._ar = capi.ALLREDUCE_FN(self._allreduce)
        self.comm = capi.PaComm(None, self.rank, self.nranks, self._ex, self._ar)
        ctx.set_comm(self.comm)

    def _exchange(self, user, stream, n, x):
        try:
            torch, dist, lib, h = self.torch, self.dist, self.ctx.lib, self.ctx.h
            ops, recvs, keep = [], [], []
            for i in range(n):
                xi = x[i]
                if xi.nsend > 0:
                    a = np.empty(xi.nsend, dtype=np.float64)
                    if lib.pa_memcpy_d2h(h, a.ctypes.data_as(C.c_void_p), C.c_void_p(xi.sendbuf), 8 * xi.nsend) != 0:  # synchronises the stream
                        return 1
                    keep.append(a)
                    ops.append(dist.P2POp(dist.isend, torch.from_numpy(a), xi.peer, group=self.group))
                    self.bytes_sent += 8 * xi.nsend
                if xi.nrecv > 0:
                    t = torch.empty(xi.nrecv, dtype=torch.float64)
                    recvs.append((xi.recvbuf, t))
                    ops.append(dist.P2POp(dist.irecv, t, xi.peer, group=self.group))
            if ops:
                for q in dist.batch_isend_irecv(ops):
                    q.wait()
            for ptr, t in recvs:
                a = t.numpy()
                if lib.pa_memcpy_h2d(h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes) != 0:
                    return 1
            self.nexchange += 1
            return 0
        except Exception:  # never let an exception cross the C boundary
            traceback.print_exc()
            return 1

    def _allreduce(self, user, vals, n, op):
        try:
            t = self.torch.tensor([vals[i] for i in range(n)], dtype=self.torch.float64)
            self.dist.all_reduce(t, op=(self.dist.ReduceOp.MIN, self.dist.ReduceOp.MAX, self.dist.ReduceOp.SUM)[op], group=self.group)
            for i in range(n):
                vals[i] = float(t[i])
            return 0
        except Exception:
            traceback.print_exc()
            return 1
