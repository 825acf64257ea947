"""N > 1 path on CPU: two ranks (gloo) each own one slab of the level BoxArray; after the local
FillBoundary + the cross-rank exchange (same region lists the HIP pack/unpack kernels consume) every
ghost cell equals what FillBoundary on the undistributed level gives."""
import os
import socket

import numpy as np
import pytest

from peleanalysis_amd import dist as padist
from peleanalysis_amd.hierarchy import MultiFab, cell_centers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _field(x, y, z, c):
    return (1 + c) * (np.sin(2 * np.pi * x / 2.0) + 0.3 * np.cos(2 * np.pi * y) + z * z) + 0 * x * y * z


def _fill(mf):
    for b in range(mf.level.nboxes):
        x, y, z = cell_centers(mf.level, b, 0)
        for c in range(mf.ncomp):
            mf.valid(b)[c] = _field(x, y, z, c)


def _worker(rank, world, port, ng):
    import torch.distributed as dist
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        R = padist.slab_hierarchy(16, 2, 8, world, rank, ng)
        for l in range(2):
            glob = MultiFab(R.glob.levels[l], 2, ng, fill=np.nan)
            _fill(glob)
            O.fill_boundary(glob, 0, 2, ng)  # the undistributed answer
            loc = MultiFab(R.local.levels[l], 2, ng, fill=np.nan)
            _fill(loc)
            O.fill_boundary(loc, 0, 2, ng)   # local half (what pa_fill_boundary does on a rank)
            padist.exchange_host(R.plans[l], loc, 0, 2)
            mine = np.nonzero(R.owner[l] == rank)[0]
            for i, g in enumerate(mine):
                a, b = loc.fab(i), glob.fab(int(g))
                same = (a.view(np.int64) == b.view(np.int64)) | (np.isnan(a) & np.isnan(b))
                assert same.all(), f"rank {rank} level {l} box {g}: {np.count_nonzero(~same)} ghost cells differ"
            if l == 0:
                assert len(R.plans[l].recv) >= 1 and all(len(v) > 0 for v in R.plans[l].recv.values())
                # sender and receiver agree on the buffer sizes
                sz = {p: R.plans[l].size(v, 2) for p, v in R.plans[l].send.items()}
                got = [None] * world
                dist.all_gather_object(got, (rank, sz, {p: R.plans[l].size(v, 2) for p, v in R.plans[l].recv.items()}))
                for (r, s_, _), in [(g,) for g in got]:
                    for p, n in s_.items():
                        assert got[p][2][r] == n
            else:
                assert not R.plans[l].send and not R.plans[l].recv  # fine levels sit inside their slab
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ng", [1, 2])
def test_two_rank_ghost_exchange_matches_undistributed_fillboundary(ng):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), ng), nprocs=2, join=True)


def test_plan_is_symmetric_and_ordered():
    """both sides enumerate (dst box, src box, shift) identically: rank a's send list for b has the same
    shapes, in the same order, as b's recv list from a"""
    world = 3
    Rs = [padist.slab_hierarchy(16, 1, 8, world, r, 2) for r in range(world)]
    for a in range(world):
        for b, regs in Rs[a].plans[0].send.items():
            peer = Rs[b].plans[0].recv[a]
            assert len(peer) == len(regs)
            assert np.array_equal(regs[:, 4:7] - regs[:, 1:4], peer[:, 4:7] - peer[:, 1:4])
