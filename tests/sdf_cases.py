"""Triangle-mesh + grid test cases for the distance function (make_level_set3), shared by the golden
generator, the oracle-vs-reference test and the GPU parity test.  Meshes come from the oracle's own
marching cubes over analytic fields (closed sphere, open wrinkled sheet cut by the grid boundary),
plus hand-made degenerate inputs."""
import numpy as np


def _mc_mesh(O, n, fun, iso):
    x = (np.arange(n) + 0.5) / n
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    f = fun(X, Y, Z)
    state = np.zeros((4, n, n, n))
    state[0], state[1], state[2], state[3] = X.transpose(2, 1, 0), Y.transpose(2, 1, 0), Z.transpose(2, 1, 0), f.transpose(2, 1, 0)
    v, _, t = O.mc_fab(state, np.ones((n, n, n)), (0, 0, 0), (n - 1, n - 1, n - 1), 3, iso, (0, 0, 0), (n - 2, n - 2, n - 2))
    return t.astype(np.uint32), v[:, :3].astype(np.float32)


def cases(O):
    """-> list of dict(name, tris, verts, origin, dx, n, band)"""
    out = []
    sph = lambda X, Y, Z: np.sqrt((X - 0.47) ** 2 + (Y - 0.52) ** 2 + (Z - 0.5) ** 2)
    t, v = _mc_mesh(O, 16, sph, 0.31)
    out.append(dict(name="sphere16", tris=t, verts=v, origin=(0.0, 0.0, 0.0), dx=1.0 / 16, n=(16, 16, 16), band=1))
    # grid shifted and anisotropic extents, band 2, mesh partly outside the grid
    out.append(dict(name="sphere16_shift", tris=t, verts=v, origin=(0.21, -0.1, 0.3), dx=1.0 / 20, n=(14, 19, 9), band=2))
    sheet = lambda X, Y, Z: Z - 0.5 - 0.08 * np.sin(7.0 * X) * np.cos(5.0 * Y)
    t2, v2 = _mc_mesh(O, 20, sheet, 0.01)
    out.append(dict(name="sheet20", tris=t2, verts=v2, origin=(0.0, 0.0, 0.0), dx=1.0 / 20, n=(20, 20, 20), band=1))
    # thin grids: an extent of 1 or 2 (sweeps with no or one plane)
    out.append(dict(name="sheet_thin", tris=t2, verts=v2, origin=(0.1, 0.1, 0.45), dx=1.0 / 20, n=(12, 1, 2), band=1))
    # degenerate triangles (zero area, repeated vertices) + one far-away triangle + one vertex on a grid point
    vd = np.array([[0.25, 0.25, 0.25], [0.25, 0.25, 0.25], [0.75, 0.25, 0.25], [0.5, 0.5, 0.5], [0.5, 0.75, 0.5], [0.5, 0.5, 0.75],
                   [5.0, 5.0, 5.0], [5.5, 5.0, 5.0], [5.0, 5.5, 5.0]], dtype=np.float32)
    td = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8], [3, 3, 3]], dtype=np.uint32)
    out.append(dict(name="degenerate", tris=td, verts=vd, origin=(0.0, 0.0, 0.0), dx=0.125, n=(9, 9, 9), band=1))
    # no triangles at all: phi stays at the upper bound
    out.append(dict(name="empty", tris=np.zeros((0, 3), np.uint32), verts=np.zeros((0, 3), np.float32), origin=(0.0, 0.0, 0.0), dx=0.1, n=(4, 5, 6),
                    band=1))
    return out
