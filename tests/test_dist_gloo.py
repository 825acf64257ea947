"""N > 1 path on the CPU tier: the owner map, the region lists behind the cross-rank exchanges (host arithmetic of the
C ABI, no GPU) and -- with 2 and 4 gloo ranks -- the data movement itself on a hierarchy whose owner map is SCATTERED
(neighbouring boxes and the coarse parents of most fine boxes live on other ranks): after the local FillBoundary + the
exchange every ghost cell equals what FillBoundary on the undistributed level gives, every cell of a rank's coarse-source
copy equals the undistributed coarse level, and the min / max reduction equals the global one."""
import os
import socket
import sys

import numpy as np
import pytest

from peleanalysis_amd import dist as padist
from peleanalysis_amd.hierarchy import Level, MultiFab, cell_centers, nested_hierarchy


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _field(x, y, z, c):
    return (1 + c) * (np.sin(2 * np.pi * x) + 0.3 * np.cos(2 * np.pi * y) + z * z) + 0 * x * y * z


def _fill(mf):
    for b in range(mf.level.nboxes):
        x, y, z = cell_centers(mf.level, b, 0)
        for c in range(mf.ncomp):
            mf.valid(b)[c] = _field(x, y, z, c)


def scattered_owner(n, nranks, seed):
    """every rank owns boxes all over the level (the opposite of a space-filling-curve map)"""
    rng = np.random.default_rng(seed)
    o = np.arange(n) % nranks
    rng.shuffle(o)
    return o.astype(np.int32)


def _sub_level(lv, idx):
    return Level(lv.boxes[idx], lv.domlo, lv.domhi, lv.is_per, lv.prob_lo, lv.prob_hi)


def test_distribution_map_is_balanced_and_follows_the_morton_curve():
    H = nested_hierarchy(64, 2, 16)  # 64 boxes per level
    for lv in H.levels:
        for n in (1, 2, 3, 4, 8):
            o = padist.distribution_map(lv.boxes, n)
            cnt = np.bincount(o, minlength=n)
            assert cnt.sum() == lv.nboxes and cnt.max() - cnt.min() <= 1
        # 8 ranks on a 4 x 4 x 4 arrangement of equal boxes: every rank gets one 2 x 2 x 2 block (a Morton octant)
        o = padist.distribution_map(lv.boxes, 8)
        for r in range(8):
            bx = lv.boxes[o == r]
            assert np.all(bx[:, 3:].max(0) - bx[:, :3].min(0) + 1 == 32)
    # unequal boxes: cell counts, not box counts, are balanced
    boxes = np.array([[0, 0, 0, 31, 31, 31], [32, 0, 0, 47, 15, 15], [32, 16, 0, 47, 31, 15], [32, 0, 16, 47, 15, 31], [32, 16, 16, 47, 31, 31],
                      [48, 0, 0, 63, 31, 31]], dtype=np.int32)
    o = padist.distribution_map(boxes, 2)
    vol = np.prod(boxes[:, 3:] - boxes[:, :3] + 1, axis=1)
    assert abs(vol[o == 0].sum() - vol[o == 1].sum()) <= vol.max()


@pytest.mark.parametrize("nranks", [2, 3, 4])
@pytest.mark.parametrize("ng", [1, 2])
def test_fill_boundary_plans_are_symmetric_and_ordered(nranks, ng):
    """rank a's send list for b holds the same shapes, in the same order, as b's receive list from a (both sides enumerate
    (destination box, source box, periodic shift) identically), and a sent region is the received one minus a period"""
    H = nested_hierarchy(16, 2, 8, is_per=(1, 1, 0))
    for li, lv in enumerate(H.levels):
        own = scattered_owner(lv.nboxes, nranks, 5 + li)
        rows = [padist.plan_fill_boundary(lv, own, r, ng) for r in range(nranks)]
        for a in range(nranks):
            for b in range(nranks):
                if a == b:
                    continue
                s = rows[a][(rows[a][:, 0] == 0) & (rows[a][:, 1] == b)]
                r = rows[b][(rows[b][:, 0] == 1) & (rows[b][:, 1] == a)]
                assert len(s) == len(r) and len(s) > 0
                assert np.array_equal(s[:, 6:9] - s[:, 3:6], r[:, 6:9] - r[:, 3:6])
                n = lv.domhi - lv.domlo + 1
                assert np.all((r[:, 3:6] - s[:, 3:6]) % n == 0)
                assert np.all(own[s[:, 2]] == a) and np.all(own[r[:, 2]] == b)


def _cf_ghost_cells(lv, gbox):
    """ring-1 face ghost cells of a box that are inside the domain (periodic wrap applied) and covered by no box of the level"""
    covered = np.zeros(tuple(lv.domhi - lv.domlo + 1)[::-1], bool)
    for b in lv.boxes:
        covered[b[2]:b[5] + 1, b[1]:b[4] + 1, b[0]:b[3] + 1] = True
    lo, hi = lv.boxes[gbox, :3], lv.boxes[gbox, 3:]
    n = lv.domhi - lv.domlo + 1
    out = []
    for d in range(3):
        for side in (0, 1):
            for v in range(lo[(d + 2) % 3], hi[(d + 2) % 3] + 1):
                for u in range(lo[(d + 1) % 3], hi[(d + 1) % 3] + 1):
                    q = [0, 0, 0]
                    q[d] = hi[d] + 1 if side else lo[d] - 1
                    q[(d + 1) % 3], q[(d + 2) % 3] = u, v
                    p = list(q)
                    ok = True
                    for t in range(3):
                        if p[t] < lv.domlo[t] or p[t] > lv.domhi[t]:
                            if not lv.is_per[t]:
                                ok = False
                            p[t] = (p[t] - lv.domlo[t]) % n[t] + lv.domlo[t]
                    if ok and not covered[p[2], p[1], p[0]]:
                        out.append((d, tuple(q)))
    return out


@pytest.mark.parametrize("nranks", [2, 4])
def test_coarse_source_pieces_are_disjoint_cover_the_stencils_and_match_the_senders(nranks):
    H = nested_hierarchy(16, 3, 8, is_per=(1, 1, 0))
    for l in (1, 2):
        fine, crse = H.levels[l], H.levels[l - 1]
        fown, cown = scattered_owner(fine.nboxes, nranks, 11 + l), scattered_owner(crse.nboxes, nranks, 23 + l)
        rows = [padist.plan_coarse_source(fine, fown, crse, cown, r) for r in range(nranks)]
        ncr = crse.domhi - crse.domlo + 1
        for r in range(nranks):
            pieces = rows[r][rows[r][:, 0] == 2]
            have = np.zeros(tuple(ncr)[::-1], np.int32)
            for p in pieces:
                cb = crse.boxes[p[2]]
                assert np.all(p[3:6] >= cb[:3]) and np.all(p[6:9] <= cb[3:]) and p[1] == cown[p[2]]  # inside the coarse box it is cut from
                have[p[5]:p[8] + 1, p[4]:p[7] + 1, p[3]:p[6] + 1] += 1
            assert have.max() == 1, "pieces overlap"
            # every coarse cell the order-3 boundary interpolation can touch for a coarse-fine ghost cell of this rank's boxes
            for g in np.nonzero(fown == r)[0]:
                for d, q in _cf_ghost_cells(fine, g):
                    qc = [v // 2 for v in q]
                    t0, t1 = [a for a in range(3) if a != d]
                    for a0 in range(-2, 3):
                        for a1 in range(-2, 3):
                            c = list(qc)
                            c[t0] += a0
                            c[t1] += a1
                            inside = True
                            for t in range(3):
                                if c[t] < crse.domlo[t] or c[t] > crse.domhi[t]:
                                    if not crse.is_per[t]:
                                        inside = False
                                    c[t] = (c[t] - crse.domlo[t]) % ncr[t] + crse.domlo[t]
                            if inside:
                                assert have[c[2], c[1], c[0]] == 1, (r, g, q, c)
            # what the other ranks send to r is exactly r's remote pieces, in order
            for a in range(nranks):
                if a == r:
                    continue
                s = rows[a][(rows[a][:, 0] == 0) & (rows[a][:, 1] == r)]
                want = pieces[pieces[:, 1] == a]
                assert np.array_equal(s[:, 2:], want[:, 2:])


def _worker(rank, world, port, ng):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H = nested_hierarchy(16, 3, 8, is_per=(1, 1, 0))
        owners = [scattered_owner(lv.nboxes, world, 31 + l) for l, lv in enumerate(H.levels)]
        globs, locs = [], []
        for l, lv in enumerate(H.levels):
            glob = MultiFab(lv, 2, ng, fill=np.nan)
            _fill(glob)
            O.fill_boundary(glob, 0, 2, ng)  # the undistributed answer
            mine = np.nonzero(owners[l] == rank)[0]
            glocal = {int(g): i for i, g in enumerate(mine)}
            loc = MultiFab(_sub_level(lv, mine), 2, ng, fill=np.nan)
            _fill(loc)
            O.fill_boundary(loc, 0, 2, ng)   # the local half (what pa_fill_boundary's kernel does on a rank)
            rows = padist.plan_fill_boundary(lv, owners[l], rank, ng)
            assert len(rows) > 0
            padist.host_exchange(rows, glocal, loc, loc, 0, 2)
            for i, g in enumerate(mine):
                a, b = loc.fab(i), glob.fab(int(g))
                same = (a.view(np.int64) == b.view(np.int64)) | (np.isnan(a) & np.isnan(b))
                assert same.all(), f"rank {rank} level {l} box {g}: {np.count_nonzero(~same)} ghost cells differ"
            globs.append(glob)
            locs.append((loc, glocal))
        # coarse-source copies: every piece equals the undistributed coarse level there
        for l in (1, 2):
            rows = padist.plan_coarse_source(H.levels[l], owners[l], H.levels[l - 1], owners[l - 1], rank)
            pieces = rows[rows[:, 0] == 2]
            assert len(pieces) > 0 and (pieces[:, 1] != rank).any(), "the scattered owner map must put coarse parents on other ranks"
            cs_lev = Level(pieces[:, 3:9], H.levels[l - 1].domlo, H.levels[l - 1].domhi, H.levels[l - 1].is_per, H.levels[l - 1].prob_lo, H.levels[l - 1].prob_hi)
            cs = MultiFab(cs_lev, 2, 0, fill=np.nan)
            loc, glocal = locs[l - 1]
            padist.host_exchange(rows, glocal, loc, cs, 0, 2, recv_kind=2, recv_box=lambda i: i)
            for i, p in enumerate(pieces):
                want = padist.host_region(globs[l - 1], int(p[2]), p[3:6], p[6:9], 0, 2)
                assert np.array_equal(cs.fab(i).view(np.int64), np.ascontiguousarray(want).view(np.int64)), (rank, l, i)
        # curvature.cpp:147-148: the progress range is reduced over the ranks
        import torch
        lo = min(float(np.nanmin(locs[l][0].valid_concat(0))) for l in range(3) if locs[l][0].level.nboxes)
        t = torch.tensor([lo], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        assert float(t) == min(float(g.valid_concat(0).min()) for g in globs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ng", [(2, 1), (2, 2), (4, 2)])
def test_ranks_exchange_ghost_cells_and_coarse_data_like_the_undistributed_level(world, ng):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), ng), nprocs=world, join=True)


@pytest.mark.parametrize("seed", [0, 4, 15, 21])
def test_coarse_source_plan_on_general_boxarrays_covers_the_neighbours_view(seed):
    """general BoxArrays (unions of rectangles): a box that may hold irregular cells rebuilds ghost normals as the NEIGHBOURING box
    sees them, i.e. boundary values of every direction at cells one layer outside the box -- the plan must hold every EXISTING
    coarse cell those interpolations can touch (the coarse plane of the ghost cell, +-2 coarse cells in the two other directions,
    for all three directions), stay disjoint, and match what the senders enumerate"""
    from peleanalysis_amd.hierarchy import union_hierarchy, _occupancy
    H = union_hierarchy(7000 + seed)
    nranks = 3
    for l in range(1, H.nlev):
        fine, crse = H.levels[l], H.levels[l - 1]
        fown, cown = scattered_owner(fine.nboxes, nranks, 11 + l), scattered_owner(crse.nboxes, nranks, 23 + l)
        rows = [padist.plan_coarse_source(fine, fown, crse, cown, r) for r in range(nranks)]
        ncr = crse.domhi - crse.domlo + 1
        nf = fine.domhi - fine.domlo + 1
        cocc, focc = _occupancy(crse), _occupancy(fine)

        def wrap(c, n, lv):
            c = list(c)
            for t in range(3):
                if c[t] < 0 or c[t] >= n[t]:
                    if not lv.is_per[t]:
                        return None
                    c[t] %= n[t]
            return c
        nchecked = 0
        for r in range(nranks):
            pieces = rows[r][rows[r][:, 0] == 2]
            have = np.zeros(tuple(ncr)[::-1], np.int32)
            for p in pieces:
                cb = crse.boxes[p[2]]
                assert np.all(p[3:6] >= cb[:3]) and np.all(p[6:9] <= cb[3:]) and p[1] == cown[p[2]]
                have[p[5]:p[8] + 1, p[4]:p[7] + 1, p[3]:p[6] + 1] += 1
            assert have.max() <= 1, "pieces overlap"
            for g in np.nonzero(fown == r)[0]:
                lo, hi = fine.boxes[g, :3], fine.boxes[g, 3:]
                # box-level "may hold irregular cells": some face ghost layer is partly covered, or an edge ghost line is a concave corner
                def valid(q):
                    w = wrap(q, nf, fine)
                    return w is not None and bool(focc[w[2], w[1], w[0]])
                suspect = False
                for d in range(3):
                    for side in (0, 1):
                        t0, t1 = [a for a in range(3) if a != d]
                        vs = []
                        for v in range(lo[t1], hi[t1] + 1):
                            for u in range(lo[t0], hi[t0] + 1):
                                q = [0, 0, 0]
                                q[d] = hi[d] + 1 if side else lo[d] - 1
                                q[t0], q[t1] = u, v
                                vs.append(valid(q))
                        suspect = suspect or (any(vs) and not all(vs))
                if not suspect:
                    continue
                # every cell within one layer of the box (faces and edges) that is not a valid cell: its boundary value in EVERY direction
                for k in range(lo[2] - 1, hi[2] + 2):
                    for j in range(lo[1] - 1, hi[1] + 2):
                        for i in range(lo[0] - 1, hi[0] + 2):
                            out = [(i < lo[0]) + (i > hi[0]), (j < lo[1]) + (j > hi[1]), (k < lo[2]) + (k > hi[2])]
                            if sum(out) == 0 or sum(out) == 3:
                                continue
                            w = wrap([i, j, k], nf, fine)
                            if w is None or focc[w[2], w[1], w[0]]:
                                continue
                            qc = [i // 2, j // 2, k // 2]
                            for t in range(3):
                                a0d, a1d = [a for a in range(3) if a != t]
                                for a0 in range(-2, 3):
                                    for a1 in range(-2, 3):
                                        c = list(qc)
                                        c[a0d] += a0
                                        c[a1d] += a1
                                        cw = wrap(c, ncr, crse)
                                        if cw is not None and cocc[cw[2], cw[1], cw[0]]:
                                            assert have[cw[2], cw[1], cw[0]] == 1, (seed, l, r, int(g), (i, j, k), c)
                                            nchecked += 1
            for a in range(nranks):
                if a == r:
                    continue
                s = rows[a][(rows[a][:, 0] == 0) & (rows[a][:, 1] == r)]
                want = pieces[pieces[:, 1] == a]
                assert np.array_equal(s[:, 2:], want[:, 2:])
        if l == 1 and seed == 0:
            assert nchecked > 0, (seed, l)  # this draw does contain boxes with mixed faces on level 1


@pytest.mark.parametrize("seed", range(6))
def test_restriction_plans_of_the_distributed_smoothing_solve(seed):
    """pa_plan_restriction (host arithmetic behind pa_smooth_solve on sharded levels): on general BoxArrays dealt to 3 ranks
    with scattered owners, (a) what a rank sends to a peer and what that peer expects from it pair up region by region, the
    destination being the source folded into the coarse domain; (b) the child averages reach every covered coarse cell exactly
    once; (c) per face orientation, every coarse cell across a coarse-fine face of the fine level receives exactly one flux
    from the ghost slab behind that face -- and no coarse cell receives two of one orientation."""
    from peleanalysis_amd.hierarchy import union_hierarchy, _occupancy
    H = union_hierarchy(7100 + seed)
    nranks = 3
    for l in range(1, H.nlev):
        fine, crse = H.levels[l], H.levels[l - 1]
        assert not fine.domlo.any() and not crse.domlo.any()
        fown, cown = scattered_owner(fine.nboxes, nranks, 41 + l), scattered_owner(crse.nboxes, nranks, 43 + l)
        ncr = (crse.domhi - crse.domlo + 1).astype(np.int64)
        cocc, focc = _occupancy(crse), _occupancy(fine)
        covered = focc[::2, ::2, ::2]
        cfb = fine.boxes.copy()
        cfb[:, :3] //= 2
        cfb[:, 3:] //= 2
        for which in range(7):
            rows = [padist.plan_restriction(fine, fown, crse, cown, r, which) for r in range(nranks)]
            got = np.zeros(cocc.shape, np.int32)

            def dst_mark(row, r):
                cb = crse.boxes[row[2]]
                assert cown[row[2]] == r and np.all(row[3:6] >= cb[:3]) and np.all(row[6:9] <= cb[3:]), "destination inside a coarse box of the receiver"
                got[row[5]:row[8] + 1, row[4]:row[7] + 1, row[3]:row[6] + 1] += 1

            def src_check(row, r):
                b = cfb[row[2]]
                assert fown[row[2]] == r
                if which == 0:
                    assert np.all(row[3:6] >= b[:3]) and np.all(row[6:9] <= b[3:]), "source inside the coarsened fine box"
                else:
                    d, side = (which - 1) >> 1, (which - 1) & 1
                    for t in range(3):
                        if t == d:
                            assert row[3 + t] == row[6 + t] == (b[3 + t] + 1 if side else b[t] - 1), "source in the ghost slab behind the face"
                        else:
                            assert row[3 + t] >= b[t] and row[6 + t] <= b[3 + t]

            def pair_check(src, dst):
                assert np.array_equal(src[6:9] - src[3:6], dst[6:9] - dst[3:6]), "same shape on both sides"
                sh = src[3:6] - dst[3:6]
                for t in range(3):
                    assert sh[t] == 0 or (which > 0 and crse.is_per[t] and abs(sh[t]) == ncr[t]), "destination = source folded through a periodic image"

            for r in range(nranks):
                R = rows[r]
                for row in R[R[:, 0] == 0]:
                    src_check(row, r)
                for row in R[R[:, 0] == 1]:
                    dst_mark(row, r)
                loc_s, loc_d = R[R[:, 0] == 3], R[R[:, 0] == 4]
                assert len(loc_s) == len(loc_d)
                for a, b in zip(loc_s, loc_d):
                    src_check(a, r)
                    dst_mark(b, r)
                    pair_check(a, b)
                for q in range(nranks):
                    if q == r:
                        continue
                    snd = R[(R[:, 0] == 0) & (R[:, 1] == q)]
                    rcv = rows[q][(rows[q][:, 0] == 1) & (rows[q][:, 1] == r)]
                    assert len(snd) == len(rcv), f"rank {r} -> {q}: {len(snd)} regions sent, {len(rcv)} expected"
                    for a, b in zip(snd, rcv):
                        pair_check(a, b)
            if which == 0:
                assert np.array_equal(got, (covered & cocc).astype(np.int32)), "every covered coarse cell gets its child average exactly once"
            else:
                d, side = (which - 1) >> 1, (which - 1) & 1
                want = np.zeros(cocc.shape, np.int32)
                for b in cfb:
                    lo, hi = b[:3].copy(), b[3:].copy()
                    lo[d] = hi[d] = b[3 + d] + 1 if side else b[d] - 1
                    if lo[d] < 0 or lo[d] >= ncr[d]:
                        if not crse.is_per[d]:
                            continue
                        lo[d] = hi[d] = lo[d] % ncr[d]
                    sl = (slice(lo[2], hi[2] + 1), slice(lo[1], hi[1] + 1), slice(lo[0], hi[0] + 1))
                    want[sl] += (cocc[sl] & ~covered[sl]).astype(np.int32)
                assert want.max() <= 1 and got.max() <= 1, "one flux per coarse cell and face orientation"
                assert np.all(got[want == 1] == 1), "every coarse cell across a coarse-fine face receives its flux"
