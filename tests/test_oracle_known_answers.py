"""The oracle is pinned by analytic known-answer tests (the reference ships no golden vectors and
cannot be built here: PARITY UNPINNED).  CPU only."""
import numpy as np
import pytest

from peleanalysis_amd.hierarchy import MultiFab, cell_centers, fill_analytic, field_trig, nested_hierarchy


def test_grad_multipass_equals_fused_bitwise(oracle):
    """reference-shaped pass-by-pass form (face fluxes, 1/bscalar, average, mult(-1)) == one sweep"""
    H = nested_hierarchy(32, 1, 16, is_per=(1, 1, 1))
    lv = H.levels[0]
    s = MultiFab(lv, 1, 1)
    fill_analytic(s, 0, field_trig)
    a, b = MultiFab(lv, 4, 0), MultiFab(lv, 4, 0)
    bc = oracle.bc_from_flags(lv.is_per)
    oracle.grad_pipeline([lv], [s.copy()], 0, bc, [a], 0, multipass=True)
    oracle.grad_pipeline([lv], [s.copy()], 0, bc, [b], 0, multipass=False)
    assert np.array_equal(a.data.view(np.int64), b.data.view(np.int64))


def test_grad_trig_central_difference_factor(oracle):
    """C1 config: d/dx of sin(kx) by central differences = k cos(kx) * sin(k h)/(k h), exactly"""
    H = nested_hierarchy(64, 1, 32, is_per=(1, 1, 1))
    lv = H.levels[0]
    s = MultiFab(lv, 1, 1)
    fill_analytic(s, 0, field_trig)
    o = MultiFab(lv, 4, 0)
    oracle.grad_pipeline([lv], [s], 0, oracle.bc_from_flags(lv.is_per), [o], 0)
    h, k = lv.dx[0], 2 * np.pi
    for b in range(lv.nboxes):
        x, y, z = cell_centers(lv, b, 0)
        gx = 600 * k * np.cos(k * x) * np.cos(2 * k * y) * np.sin(k * z + 0.3) * np.sin(k * h) / (k * h)
        gy = -600 * 2 * k * np.sin(k * x) * np.sin(2 * k * y) * np.sin(k * z + 0.3) * np.sin(2 * k * h) / (2 * k * h)
        v = o.valid(b)
        assert np.abs(v[0] - gx).max() < 1e-9 * 600 * k
        assert np.abs(v[1] - gy).max() < 1e-9 * 1200 * k
        assert np.allclose(v[3], np.sqrt(v[0] ** 2 + v[1] ** 2 + v[2] ** 2), rtol=0, atol=0)


@pytest.mark.parametrize("fn,exact", [
    (lambda x, y, z: 3.0 * x - 2.0 * y + 0.5 * z + 1.0 + 0 * x * y * z, lambda x, y, z: (3.0 + 0 * x * y * z, -2.0 + 0 * x * y * z, 0.5 + 0 * x * y * z)),
    (lambda x, y, z: x * x + 2 * y * y - z * z + x * y + y * z - 0.5 * x * z + x,
     lambda x, y, z: (2 * x + y - 0.5 * z + 1 + 0 * y * z, 4 * y + x + z + 0 * x * z, -2 * z + y - 0.5 * x + 0 * y)),
])
def test_grad_exact_for_quadratics_across_coarse_fine(oracle, fn, exact):
    """central differences + the coarse-fine boundary interpolation (tangential order 3 with cross
    term, normal cubic) are exact for quadratics: fine levels must reproduce the analytic gradient"""
    H = nested_hierarchy(32, 3, 16, is_per=(0, 0, 0))
    sts, outs = [], []
    for lv in H.levels:
        s = MultiFab(lv, 1, 1)
        fill_analytic(s, 0, fn)
        sts.append(s)
        outs.append(MultiFab(lv, 4, 0))
    oracle.grad_pipeline(H.levels, sts, 0, oracle.bc_from_flags((0, 0, 0)), outs, 0)
    for l in (1, 2):  # fine levels do not touch the physical walls
        lv = H.levels[l]
        for b in range(lv.nboxes):
            x, y, z = cell_centers(lv, b, 0)
            ex = exact(x, y, z)
            for d in range(3):
                assert np.abs(outs[l].valid(b)[d] - ex[d]).max() < 2e-13, (l, b, d)


def test_neumann_and_reflect_odd_walls(oracle):
    """wall cell gradient: Neumann ghost = interior -> half a one-sided difference; reflect_odd ghost = -interior"""
    H = nested_hierarchy(16, 1, 8, is_per=(1, 1, 0))
    lv = H.levels[0]
    f = lambda x, y, z: 2.0 + z + 0 * x * y
    for sym, expect in ((0, 0.5), (1, None)):
        s = MultiFab(lv, 1, 1)
        fill_analytic(s, 0, f)
        o = MultiFab(lv, 4, 0)
        oracle.grad_pipeline([lv], [s], 0, oracle.bc_from_flags((1, 1, 0), (0, 0, sym)), [o], 0)
        for b in range(lv.nboxes):
            if lv.boxes[b, 2] == 0:
                gz = o.valid(b)[2][0]
                z0 = 0.5 * lv.dx[2]
                want = 0.5 if sym == 0 else (2.0 + 1.5 * lv.dx[2] + (2.0 + z0)) / (2 * lv.dx[2])
                assert np.allclose(gz, want, rtol=1e-13)


def test_curvature_of_sphere(oracle):
    """c = distance from a point: n = -r_hat, mean curvature 0.5 div n = -1/r, Gaussian = 1/r^2, O(h^2)"""
    H = nested_hierarchy(32, 3, 16, is_per=(1, 1, 0))
    f = lambda x, y, z: np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) + 0 * x * y * z
    sts, outs = [], []
    for lv in H.levels:
        s = MultiFab(lv, 1, 2)
        fill_analytic(s, 0, f)
        sts.append(s)
        outs.append(MultiFab(lv, 8, 0))
    oracle.curvature_pipeline(H.levels, sts, 0, oracle.bc_from_flags((1, 1, 0)), outs, 0, MultiFab, do_gauss=True)
    errs = []
    for l, lv in enumerate(H.levels):
        e = eg = 0.0
        for b in range(lv.nboxes):
            x, y, z = cell_centers(lv, b, 0)
            r = f(x, y, z)
            m = (r > 0.1) & (r < 0.22)
            if m.any():
                e = max(e, np.abs(outs[l].valid(b)[1][m] * r[m] + 1).max())
                eg = max(eg, np.abs(outs[l].valid(b)[5][m] * r[m] ** 2 - 1).max())
                nn = np.sqrt(sum(outs[l].valid(b)[2 + d][m] ** 2 for d in range(3)))
                assert np.abs(nn - 1).max() < 1e-12  # unit normal
        errs.append((e, eg))
    assert errs[0][0] < 0.08 and errs[1][0] < 0.03 and errs[2][0] < 0.008  # second order
    assert errs[2][1] < 0.02


def test_threshold_clips_curvature_and_normal(oracle):
    H = nested_hierarchy(16, 2, 8, is_per=(1, 1, 1))
    f = lambda x, y, z: np.tanh((np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) - 0.25) / 0.1) + 0 * x * y * z
    sts = []
    outs = [MultiFab(lv, 5, 0) for lv in H.levels]
    for lv in H.levels:
        s = MultiFab(lv, 1, 2)
        fill_analytic(s, 0, f)
        sts.append(s)
    oracle.curvature_pipeline(H.levels, sts, 0, [0, 0, 0], outs, 0, MultiFab, threshold=0.2)
    for l, lv in enumerate(H.levels):
        for b in range(lv.nboxes):
            v = outs[l].valid(b)
            clipped = (v[0] < 0.2) | (v[0] > 0.8)
            assert clipped.any() and (~clipped).any()
            for c in (1, 2, 3, 4):
                assert np.all(v[c][clipped] == 0.0)
            assert np.all(np.abs(v[2][~clipped]) + np.abs(v[3][~clipped]) + np.abs(v[4][~clipped]) > 0)


def test_box_filter_weights_and_identity(oracle):
    for fgr, ng in ((2, 1), (4, 2), (8, 4), (6, 3)):
        n, w = oracle.box_filter_weights(fgr)
        assert n == ng and len(w) == 2 * ng + 1
        assert abs(w.sum() - 1.0) < 1e-15 and w[0] == w[-1] == 0.5 / fgr and np.all(w[1:-1] == 1.0 / fgr)
    # constants and linear fields are reproduced in the interior (symmetric weights, sum 1)
    H = nested_hierarchy(16, 1, 8, is_per=(1, 1, 1))
    lv = H.levels[0]
    s = MultiFab(lv, 2, 2)
    fill_analytic(s, 0, lambda x, y, z: 7.5 + 0 * x * y * z)
    fill_analytic(s, 1, lambda x, y, z: 1.0 + 2 * x - 3 * y + 0.25 * z + 0 * x * y * z)
    o = [MultiFab(lv, 2, 0)]
    oracle.filter_pipeline([lv], [s], o, 2, base_fgr=4)
    for b in range(lv.nboxes):
        x, y, z = cell_centers(lv, b, 0)
        assert np.abs(o[0].valid(b)[0] - 7.5).max() < 1e-13
        interior = (x > 0.2) & (x < 0.8) & (y > 0.2) & (y < 0.8) & (z > 0.2) & (z < 0.8)  # away from the periodic jump
        lin = 1.0 + 2 * x - 3 * y + 0.25 * z + 0 * x * y * z
        if interior.any():
            assert np.abs(o[0].valid(b)[1] - lin)[interior].max() < 1e-13


def _manifold_checks(nodes, elts):
    """checkIso.cpp:127-149: every edge is used by exactly two triangles, in opposite directions"""
    edges = {}
    for t in elts:
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            edges[(a, b)] = edges.get((a, b), 0) + 1
    for (a, b), n in edges.items():
        assert n == 1, "directed edge used twice (orientation flip)"
        assert (b, a) in edges, "open edge: surface not closed"
    V, E, F = len(nodes), len(edges) // 2, len(elts)
    return V - E + F


def test_isosurface_sphere_closed_manifold_amr(oracle):
    """sphere crossing both coarse-fine interfaces of a 3-level hierarchy: closed, orientable
    (checkIso invariant), Euler characteristic 2, area -> 4 pi R^2, mapped component interpolated"""
    H = nested_hierarchy(32, 3, 16, is_per=(0, 0, 0))
    R = 0.3
    fields = []
    for lv in H.levels:
        s = MultiFab(lv, 2, 0)
        fill_analytic(s, 0, lambda x, y, z: 1000.0 + 400.0 * (np.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2) - R) + 0 * x * y * z)
        fill_analytic(s, 1, lambda x, y, z: x + 2 * y + 3 * z + 0 * x * y * z)
        fields.append(s)
    nodes, elts = oracle.isosurface_pipeline(H.levels, fields, [0, 1], 0, 1000.0, MultiFab)
    assert len(elts) > 2000
    assert _manifold_checks(nodes, elts) == 2
    p = nodes[:, :3]
    r = np.sqrt(((p - 0.5) ** 2).sum(1))
    assert np.abs(r - R).max() < 2e-3                      # vertices on the sphere (linear interpolation of a distance)
    assert np.abs(nodes[:, 3] - 1000.0).max() < 1e-9       # iso component equals isoVal at every vertex
    assert np.abs(nodes[:, 4] - (p[:, 0] + 2 * p[:, 1] + 3 * p[:, 2])).max() < 1e-12  # linear field mapped exactly
    a, b, c = p[elts[:, 0]], p[elts[:, 1]], p[elts[:, 2]]
    area = 0.5 * np.sqrt((np.cross(b - a, c - a) ** 2).sum(1)).sum()
    assert abs(area / (4 * np.pi * R * R) - 1) < 0.01
    # consistent orientation: normals all point the same way relative to the centre
    nrm = np.cross(b - a, c - a)
    s = np.sign((nrm * ((a + b + c) / 3 - 0.5)).sum(1))
    assert np.all(s == s[0])


def test_marching_squares_circle_known_answer(oracle):
    """2-D restatement (Segmentise + node/element sets + MakeCLines): the contour of r = 0.3 on a 64^2 grid is ONE closed
    line whose length is 2 pi r to second order; every node lies on the circle to second order; the seed segment of the
    line search is consumed (reference quirk), so the line holds one segment fewer than the element set"""
    n = 64
    x = (np.arange(-1, n + 1) + 0.5) / n
    X, Y = np.meshgrid(x, x)  # [j][i]
    r = np.sqrt((X - 0.5) ** 2 + (Y - 0.5) ** 2)
    state = np.stack([X, Y, r])
    mask = np.ones_like(r)
    lo, hi = np.array([-1, -1]), np.array([n, n])
    verts, vkeys, segs = oracle.msq_fab(state, mask, lo, hi, 2, 0.3, np.array([0, 0]), np.array([n - 2, n - 2]))
    assert len(segs) == len(verts) > 100  # closed curve: as many segments as vertices
    assert np.abs(np.sqrt((verts[:, 0] - 0.5) ** 2 + (verts[:, 1] - 0.5) ** 2) - 0.3).max() < 1.0 / n ** 2
    assert np.abs(verts[:, 2] - 0.3).max() < 1e-12  # the iso component is interpolated to the iso value
    order = np.lexsort((vkeys[:, 2], vkeys[:, 3], vkeys[:, 0], vkeys[:, 1]))
    assert np.array_equal(order, np.arange(len(vkeys)))  # vertCache order: (j, i) of the lower endpoint, then of the upper
    nodes, elts = oracle.iso2d_merge([(verts, segs)], 3)
    assert len(nodes) == len(verts) and len(elts) == len(segs) and (elts[:, 0] < elts[:, 1]).all()
    length = np.sqrt(((nodes[elts[:, 0], :2] - nodes[elts[:, 1], :2]) ** 2).sum(axis=1)).sum()
    assert abs(length - 2 * np.pi * 0.3) < 2e-3
    lines = oracle.make_clines(elts)
    assert len(lines) == 1 and len(lines[0]) == len(elts) - 1
    for (a, b), (c, d) in zip(lines[0][:-1], lines[0][1:]):
        assert b == c  # consecutive segments share a node
    # a masked corner removes its four squares; the exact-hit branches of VI_doIt copy an endpoint
    i5, j5 = int(vkeys[5][0]), int(vkeys[5][1])  # a grid point next to the contour
    mask2 = mask.copy()
    mask2[j5 + 1, i5 + 1] = -1.0
    v2, _, s2 = oracle.msq_fab(state, mask2, lo, hi, 2, 0.3, np.array([0, 0]), np.array([n - 2, n - 2]))
    assert len(s2) < len(segs)
    st3 = state.copy()
    st3[2, j5 + 1, i5 + 1] = 0.3
    v3, _, _ = oracle.msq_fab(st3, mask, lo, hi, 2, 0.3, np.array([0, 0]), np.array([n - 2, n - 2]))
    assert ((v3[:, 0] == X[j5 + 1, i5 + 1]) & (v3[:, 1] == Y[j5 + 1, i5 + 1])).any()


# ------------------------------------------------------------------------------------------------------------------
# Known answers for the RECALLED AMReX pieces of the oracle.  Which lines each test constrains is
# listed in DESIGN.md section 1.
def _two_level(ncrse, clo, chi, is_per, box=None):
    """coarse level ncrse^3 on [0,1]^3 + one fine level over the coarse cells clo..chi (inclusive, per direction)"""
    from peleanalysis_amd.hierarchy import Level, chop_box
    clo, chi = np.asarray(clo), np.asarray(chi)
    L0 = Level(chop_box((0, 0, 0), (ncrse - 1,) * 3, box or ncrse), (0, 0, 0), (ncrse - 1,) * 3, is_per, (0., 0., 0.), (1., 1., 1.))
    flo, fhi = 2 * clo, 2 * chi + 1
    L1 = Level(chop_box(flo, fhi, box or 2 * ncrse), (0, 0, 0), (2 * ncrse - 1,) * 3, is_per, (0., 0., 0.), (1., 1., 1.))
    return L0, L1


def _cf_face_ghosts(lv, b, d, side):
    """index arrays (into fab(b) with ng ghost layers) of the ring-1 ghost cells behind face (d, side) of box b"""
    lo, hi = lv.boxes[b, :3], lv.boxes[b, 3:]
    return lo, hi


@pytest.mark.parametrize("d", [0, 1, 2])
def test_cf_ghost_exact_for_normal_cubic_plus_tangential_quadratic(oracle, d):
    """applyBC at a coarse-fine face (pa_oracle.c orc_apply_bc + cf_bndry_value): the ghost value is the cubic through
    {boundary value at the coarse cell centre, 3 interior cells} evaluated at the ghost centre, and the boundary value
    is the order-3 tangential interpolant WITH the cross term.  Both are exact for f = cubic(normal) + quadratic
    (tangential, incl. the mixed term), so every coarse-fine ghost cell must hold f(ghost centre)."""
    L0, L1 = _two_level(16, (4, 4, 4), (11, 11, 11), (0, 0, 0), box=8)
    t0, t1 = [a for a in range(3) if a != d]

    def f(x, y, z):
        X = (x, y, z)
        n, a, b = X[d], X[t0], X[t1]
        return 1.0 + 0.7 * n - 1.3 * n * n + 2.1 * n ** 3 + 0.4 * a - 0.9 * b + 1.7 * a * a - 0.6 * b * b + 2.3 * a * b + 0 * x * y * z
    c, s = MultiFab(L0, 1, 1), MultiFab(L1, 1, 1)
    fill_analytic(c, 0, f)
    fill_analytic(s, 0, f)
    oracle.fill_boundary(s, 0, 1, 1)
    oracle.apply_bc(s, 0, c, 0, oracle.bc_from_flags((0, 0, 0)))
    nchecked = 0
    for b in range(L1.nboxes):
        x, y, z = cell_centers(L1, b, 1)
        want = f(x, y, z)
        lo, hi = L1.boxes[b, :3], L1.boxes[b, 3:]
        for side in (0, 1):
            on_cf = (lo[d] == 8) if side == 0 else (hi[d] == 23)  # faces of the fine REGION: every ghost cell is coarse-fine
            if not on_cf:
                continue
            sl = [slice(1, -1)] * 3
            sl[2 - d] = slice(0, 1) if side == 0 else slice(-1, None)
            got = s.fab(b)[0][tuple(sl)]
            assert np.abs(got - want[tuple(sl)]).max() < 5e-14, (d, b, side)
            nchecked += got.size
    assert nchecked == 2 * 16 * 16


def test_cf_cross_term_only_when_all_four_diagonals_are_coarse_fine(oracle):
    """InterpBndryData drops the mixed term (and shifts the tangential stencil one-sided) where a tangential neighbour
    of the ghost cell is not itself a coarse-fine ghost position.  Fine region flush with the z wall, f = y z on an
    x-face: the boundary value of ghost cells in the first fine z-pair (coarse k = 0) misses exactly the term
    yy * zz * Dyz = (+-1/4)(+-1/4) H^2, which reaches the ghost value through the cubic's weight 16/35;
    all other ghost cells of the face are exact."""
    L0, L1 = _two_level(16, (4, 4, 0), (11, 11, 7), (0, 0, 0))
    f = lambda x, y, z: y * z + 0 * x
    c, s = MultiFab(L0, 1, 1), MultiFab(L1, 1, 1)
    fill_analytic(c, 0, f)
    fill_analytic(s, 0, f)
    oracle.apply_bc(s, 0, c, 0, oracle.bc_from_flags((0, 0, 0)), only_dir=0)
    x, y, z = cell_centers(L1, 0, 1)
    want = f(x, y, z)
    H = 1.0 / 16
    for xs in (0, -1):
        got = s.fab(0)[0][1:-1, 1:-1, xs]
        err = got - want[1:-1, 1:-1, xs]
        assert np.abs(err[2:, :]).max() < 1e-15  # coarse k >= 1: centred z stencil, cross term on -> exact
        # coarse k = 0 (fine k = 0, 1): the quadratic one-sided z stencil is exact for a field linear in z, the cross term is missing
        jj = np.arange(8, 24)
        yy = np.where(jj % 2 == 0, -0.25, 0.25)[None, :]
        zz = np.array([-0.25, 0.25])[:, None]
        # the boundary value enters the normal cubic with the Lagrange weight 16/35
        assert np.abs(err[:2, :] + (16.0 / 35.0) * yy * zz * H * H).max() < 1e-15


def _cons_interp_numpy(cr, parent, child_off):
    """independent restatement of mf_cell_cons_lin_interp_mcslope + mf_cell_cons_lin_interp for ratio 2 on a periodic
    coarse array cr[k][j][i] (numpy, written from the formulas in SURVEY A.6 / DESIGN 1, not from the C oracle)"""
    k, j, i = parent
    n = cr.shape[0]
    u = lambda di, dj, dk: cr[(k + dk) % n, (j + dj) % n, (i + di) % n]
    u0 = u(0, 0, 0)
    s = []
    for e in ((1, 0, 0), (0, 1, 0), (0, 0, 1)):
        up, um = u(*e), u(-e[0], -e[1], -e[2])
        dc, df, db = 0.5 * (up - um), 2.0 * (up - u0), 2.0 * (u0 - um)
        sl = min(abs(df), abs(db)) if df * db >= 0.0 else 0.0
        s.append(np.copysign(1.0, dc) * min(sl, abs(dc)))
    alpha = 1.0
    if any(v != 0.0 for v in s):
        dumax = abs(s[0]) * 1.0 / 4.0 + abs(s[1]) * 1.0 / 4.0 + abs(s[2]) * 1.0 / 4.0
        nb = [u(a, b, c) for c in (-1, 0, 1) for b in (-1, 0, 1) for a in (-1, 0, 1)]
        umax, umin = max(nb), min(nb)
        if dumax * alpha > umax - u0:
            alpha = (umax - u0) / dumax
        if dumax * alpha > u0 - umin:
            alpha = (u0 - umin) / dumax
    return u0 + child_off[0] * (s[0] * alpha) + child_off[1] * (s[1] * alpha) + child_off[2] * (s[2] * alpha)


def _ghost_children(L1, s, ng=2):
    """(parent (ic,jc,kc), child offset (+-1/4)^3, value) of every ghost cell of the one-box fine level that no fine box covers"""
    lo, hi = L1.boxes[0, :3], L1.boxes[0, 3:]
    f = s.fab(0)[0]
    out = []
    for k in range(lo[2] - ng, hi[2] + ng + 1):
        for j in range(lo[1] - ng, hi[1] + ng + 1):
            for i in range(lo[0] - ng, hi[0] + ng + 1):
                if lo[0] <= i <= hi[0] and lo[1] <= j <= hi[1] and lo[2] <= k <= hi[2]:
                    continue
                p = (i // 2, j // 2, k // 2)
                off = tuple(0.25 if q % 2 else -0.25 for q in (i, j, k))
                out.append((p, off, f[k - lo[2] + ng, j - lo[1] + ng, i - lo[0] + ng]))
    return out


def test_cell_conservative_interp_conserves_and_stays_in_bounds(oracle):
    """FillPatchTwoLevels / mf_cell_cons_interp (pa_oracle.c orc_fillpatch_two_levels) on random coarse data: the 8 children
    of a coarse cell average to the parent (offsets +-1/4 cancel), every child lies within the min / max of the parent's
    27 neighbours, and every child equals the independent numpy restatement bit for bit."""
    L0, L1 = _two_level(8, (2, 2, 2), (5, 5, 5), (1, 1, 1))
    rng = np.random.default_rng(7)
    cr = rng.uniform(-1, 1, size=(8, 8, 8))
    c, s = MultiFab(L0, 1, 0), MultiFab(L1, 1, 2)
    c.valid(0)[0] = cr
    assert oracle.fillpatch_two_levels(s, c, 0, 1, 2) == 0
    groups = {}
    for p, off, v in _ghost_children(L1, s):
        assert v == _cons_interp_numpy(cr, (p[2], p[1], p[0]), off), (p, off)
        groups.setdefault(p, []).append(v)
        nb = [cr[(p[2] + a) % 8, (p[1] + b) % 8, (p[0] + d) % 8] for a in (-1, 0, 1) for b in (-1, 0, 1) for d in (-1, 0, 1)]
        assert min(nb) - 1e-15 <= v <= max(nb) + 1e-15
    full = [p for p, vs in groups.items() if len(vs) == 8]
    assert len(full) == 6 ** 3 - 4 ** 3  # the ring of coarse cells around the fine region is covered by whole children sets
    for p in full:
        assert abs(np.mean(groups[p]) - cr[p[2], p[1], p[0]]) < 2e-16 * 8


def test_cell_conservative_interp_known_answers(oracle):
    """(i) a linear field is reproduced exactly away from extrema; (ii) at a local extremum all slopes vanish (children ==
    parent); (iii) the common factor uses dumax = sum |s_d| (r-1)/(2r), the excursion of the fine CELL CENTRES: with
    u(-1) = -2, u(0) = 0, u(+1) = 1/2 in every direction and all other neighbours <= 1/2 the limited slopes are 1, 1, 1,
    dumax = 3/4 > umax - u0 = 1/2, alpha = 2/3, and the (+,+,+) child lands exactly ON the bound 1/2 (the cell-corner
    form 1/2 sum |s_d| would give alpha = 1/3 and 1/4)."""
    L0, L1 = _two_level(8, (2, 2, 2), (5, 5, 5), (1, 1, 1))
    c, s = MultiFab(L0, 1, 0), MultiFab(L1, 1, 2)
    # (i) linear in the interior of the periodic box (the wrap-around jump is 3 cells away from every parent used)
    x, y, z = cell_centers(L0, 0, 0)
    lin = lambda x, y, z: 2.0 * x - 3.0 * y + 0.5 * z + 0 * x * y * z
    c.valid(0)[0] = lin(x, y, z)
    assert oracle.fillpatch_two_levels(s, c, 0, 1, 2) == 0
    xf, yf, zf = cell_centers(L1, 0, 2)
    m = np.ones(s.fab(0)[0].shape, bool)
    m[2:-2, 2:-2, 2:-2] = False
    assert np.abs(s.fab(0)[0] - lin(xf, yf, zf))[m].max() < 1e-15
    # (ii) + (iii): hand-made 3 x 3 x 3 neighbourhood around the parent (1, 3, 3) (a ghost parent on the low-x side)
    cr = np.full((8, 8, 8), -5.0)
    P = (3, 3, 1)  # [k][j][i]
    cr[P] = 0.0
    for ax in range(3):
        lo, hi = list(P), list(P)
        lo[ax] -= 1
        hi[ax] += 1
        cr[tuple(lo)], cr[tuple(hi)] = -2.0, 0.5
    c.valid(0)[0] = cr
    assert oracle.fillpatch_two_levels(s, c, 0, 1, 2) == 0
    kids = {off: v for p, off, v in _ghost_children(L1, s) if p == (1, 3, 3)}
    assert len(kids) == 8
    assert kids[(0.25, 0.25, 0.25)] == 0.5 and kids[(-0.25, -0.25, -0.25)] == -0.5
    assert abs(kids[(0.25, -0.25, 0.25)] - (0.25 * 2.0 / 3.0)) < 1e-16
    cr2 = np.full((8, 8, 8), -5.0)
    cr2[P] = 1.0  # strict local maximum: df * db < 0 in every direction
    c.valid(0)[0] = cr2
    assert oracle.fillpatch_two_levels(s, c, 0, 1, 2) == 0
    assert all(v == 1.0 for p, off, v in _ghost_children(L1, s) if p == (1, 3, 3))


def test_cell_conservative_interp_next_to_a_wall(oracle):
    """filterPlt fills the coarse ghost cells beyond a non-periodic wall with foextrap (filterPlt.cpp:164-173) and
    mf_compute_slopes keeps the central form for foextrap: at a wall-adjacent coarse parent u(-1) == u(0), so db = 0 and the
    wall-normal slope is 0 -- the two children across that direction are equal -- while the tangential slopes are untouched."""
    L0, L1 = _two_level(8, (2, 2, 0), (5, 5, 3), (0, 0, 0))
    c, s = MultiFab(L0, 1, 0), MultiFab(L1, 1, 2)
    x, y, z = cell_centers(L0, 0, 0)
    c.valid(0)[0] = 1.0 + 2.0 * x + 3.0 * z + 0 * y
    assert oracle.fillpatch_two_levels(s, c, 0, 1, 2) == 0
    f = s.fab(0)[0]  # box lo = (4, 4, 0), ng = 2: fab index = cell - lo + 2
    xf, yf, zf = cell_centers(L1, 0, 2)
    # ghost cells at fine i = 2, 3 (parent ic = 1), fine k = 0, 1 (parent kc = 0, on the wall), any j inside
    g = f[2:4, 4:-4, 0:2]
    assert np.all(g[0] == g[1])                                    # no z slope at the wall parent
    assert np.abs(g[0] - (1.0 + 2.0 * xf[0, 0, 0:2] + 3.0 * (0.5 / 8))).max() < 1e-15  # x slope exact, z frozen at the parent centre
    g2 = f[4:6, 4:-4, 0:2]                                         # parent kc = 1: centred, linear field exact
    assert np.abs(g2 - (1.0 + 2.0 * xf[:, :, 0:2] + 3.0 * zf[4:6] + 0 * yf[:, 4:-4])).max() < 1e-15


def test_filter_type_weights_match_their_moment_conditions(oracle):
    """the closed-form PelePhysics filter types restated without the PelePhysics source (types 3 / 7 and 4 / 8): what pins
    them is their DEFINITION -- unit sum, second moment fgr^2/12 of the box / Gaussian filter, and for the 5-point forms
    the fourth moment fgr^4/80 (box) or fgr^4/48 (Gaussian of the same variance) -- hence exactness on polynomials:
    a filtered x^2 is x^2 + fgr^2 h^2 / 12.  Type 0 is the identity, type 1 the trapezoid box weights."""
    for fgr in (1, 2, 3, 4, 8):
        k3 = np.arange(-1, 2)
        k5 = np.arange(-2, 3)
        for t in (3, 7):
            ng, w = oracle.filter_weights(t, fgr)
            assert ng == 1 and abs(w.sum() - 1) < 1e-15 and abs((w * k3 ** 2).sum() - fgr ** 2 / 12) < 1e-14
        for t, m4 in ((4, fgr ** 4 / 80), (8, fgr ** 4 / 48)):
            ng, w = oracle.filter_weights(t, fgr)
            assert ng == 2 and abs(w.sum() - 1) < 1e-14 and abs((w * k5 ** 2).sum() - fgr ** 2 / 12) < 1e-13 and abs((w * k5 ** 4).sum() - m4) < 1e-12
            assert np.array_equal(w, w[::-1])
        assert np.array_equal(oracle.filter_weights(3, fgr)[1], oracle.filter_weights(7, fgr)[1])
        assert oracle.filter_weights(0, fgr) [0] == 0 and oracle.filter_weights(0, fgr)[1].tolist() == [1.0]
        if fgr % 2 == 0:
            ng, w = oracle.filter_weights(1, fgr)
            assert np.array_equal(w, oracle.box_filter_weights(fgr)[1])
    for t in (5, 6, 9, 10, 11, -1):  # (2, the Gaussian: test_filter_weights_moment_conditions)
        assert oracle.filter_weights(t, 2) is None
    # through the pipeline: one periodic level, f = x^2-like polynomial in index space is reproduced + fgr^2/12 per direction
    from peleanalysis_amd.hierarchy import Level, MultiFab, chop_box
    n = 16
    lv = Level(chop_box((0, 0, 0), (n - 1,) * 3, 8), (0, 0, 0), (n - 1,) * 3, (1, 1, 1), (0, 0, 0), (1, 1, 1))
    for t, ng in ((3, 1), (4, 2), (8, 2)):
        m = MultiFab(lv, 1, ng)
        for b in range(lv.nboxes):
            B = lv.boxes[b]
            z, y, x = np.meshgrid(np.arange(B[2], B[5] + 1), np.arange(B[1], B[4] + 1), np.arange(B[0], B[3] + 1), indexing="ij")
            m.valid(b)[0] = np.cos(2 * np.pi * x / n) + 0.5 * np.sin(2 * np.pi * (y + 2 * z) / n)
        out = MultiFab(lv, 1, 0)
        oracle.filter_pipeline([lv], [m], [out], 1, base_fgr=2, filter_type=t)
        _, w = oracle.filter_weights(t, 2)
        tf = lambda kk: sum(w[q + ng] * np.cos(2 * np.pi * kk * q / n) for q in range(-ng, ng + 1))  # transfer function of the symmetric stencil
        for b in range(lv.nboxes):
            B = lv.boxes[b]
            z, y, x = np.meshgrid(np.arange(B[2], B[5] + 1), np.arange(B[1], B[4] + 1), np.arange(B[0], B[3] + 1), indexing="ij")
            want = tf(1) * tf(0) * tf(0) * np.cos(2 * np.pi * x / n) + 0.5 * tf(0) * tf(1) * tf(2) * np.sin(2 * np.pi * (y + 2 * z) / n)
            assert np.abs(out.valid(b)[0] - want).max() < 1e-14


def test_filter_weights_moment_conditions(oracle):
    """Known answers for the PelePhysics filter types whose weights are restated from their defining conditions (the source is
    not in the reference tree): every type sums to one and is symmetric (constants and linear fields pass unchanged); the 3-point
    types carry the box filter's second moment fgr^2 / 12; the 5-point types also the fourth moment fgr^4 / 80 (box, type 4) or
    3 (fgr^2 / 12)^2 (Gaussian, type 8); the sampled Gaussian (type 2, flagged unverified) is positive, decreasing from the
    centre, cut at 4 standard deviations and within 4 % of the continuous kernel's second moment."""
    import numpy as np
    for fgr in (2, 3, 4, 6, 8):
        for ftype in (1, 2, 3, 4, 7, 8):
            if ftype == 1 and fgr % 2:
                continue
            got = oracle.filter_weights(ftype, fgr)
            assert got is not None, (ftype, fgr)
            ng, w = got
            i = np.arange(-ng, ng + 1, dtype=np.float64)
            assert len(w) == 2 * ng + 1 and abs(w.sum() - 1.0) < 1e-14 and np.array_equal(w, w[::-1]), (ftype, fgr)
            m2, m4 = float((w * i ** 2).sum()), float((w * i ** 4).sum())
            if ftype in (3, 7, 4, 8):
                assert abs(m2 - fgr ** 2 / 12.0) < 1e-13 * max(1.0, fgr ** 2), (ftype, fgr, m2)
            if ftype == 4:
                assert abs(m4 - fgr ** 4 / 80.0) < 1e-12 * fgr ** 4, (fgr, m4)
            if ftype == 8:
                assert abs(m4 - fgr ** 4 / 48.0) < 1e-12 * fgr ** 4, (fgr, m4)
            if ftype == 1:  # trapezoid over [-fgr/2, fgr/2]: second moment fgr^2 / 12 + 1 / 6 (end-point rule)
                assert abs(m2 - (fgr ** 2 / 12.0 + 1.0 / 6.0)) < 1e-13 * fgr ** 2, (fgr, m2)
            if ftype == 2:
                assert ng == max(1, int(np.ceil(4.0 * fgr / np.sqrt(12.0)))) and np.all(w > 0) and np.all(np.diff(w[ng:]) < 0)
                assert abs(m2 - fgr ** 2 / 12.0) < 0.04 * fgr ** 2 / 12.0, (fgr, m2)
                ref = np.exp(-6.0 * i ** 2 / fgr ** 2)
                assert np.allclose(w, ref / ref.sum(), rtol=1e-15, atol=0)
    assert oracle.filter_weights(5, 2) is None and oracle.filter_weights(2, 16) is None  # tabulated types / wider than 16 ghost cells: refused


def test_fused_single_sweep_cpu_variant_equals_the_pass_by_pass_pipelines(oracle):
    """bench.py's cpu_baseline "fused" variant (oracle.gradcurv_fused_pipeline: one sweep per level + the first layer of every
    box) against grad_pipeline + curvature_pipeline on random hierarchies incl. unions of rectangles, bit for bit"""
    import numpy as np
    from peleanalysis_amd.hierarchy import MultiFab
    from test_retile import _draw_h
    from util import bits_equal, make_states
    for seed in range(6):
        H, per, sym, fn = _draw_h(seed)
        states = make_states(H, 1, 2, fn, seed=seed)
        bc = oracle.bc_from_flags(per, sym)
        og = [MultiFab(lv, 4, 0) for lv in H.levels]
        oracle.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0, multipass=True)
        oc = [MultiFab(lv, 5, 0) for lv in H.levels]
        pmin, pmax = oracle.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab)
        g2, n2, k2 = [MultiFab(lv, 4, 0) for lv in H.levels], [MultiFab(lv, 3, 1) for lv in H.levels], [MultiFab(lv, 1, 0) for lv in H.levels]
        oracle.gradcurv_fused_pipeline(H.levels, [s.copy() for s in states], 0, bc, g2, n2, k2, MultiFab, pmin, pmax)
        for l, lv in enumerate(H.levels):
            for b in range(lv.nboxes):
                assert bits_equal(g2[l].valid(b), og[l].valid(b)), (seed, l, b, "gradient")
                assert bits_equal(n2[l].valid(b), oc[l].valid(b)[2:5]), (seed, l, b, "normal")
                assert bits_equal(k2[l].valid(b)[0], oc[l].valid(b)[1]), (seed, l, b, "curvature")
