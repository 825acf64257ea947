"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from peleanalysis_amd.hierarchy import MultiFab, fill_analytic, nested_hierarchy, field_flame, field_trig


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    return np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(b).view(np.int64))


def assert_valid_bits_equal(got: MultiFab, want: MultiFab, comps, what=""):
    """bit-exact comparison of the valid cells of the given comps (pairs (gcomp, wcomp))."""
    for gc, wc in comps:
        for b in range(got.level.nboxes):
            g, w = got.valid(b)[gc], want.valid(b)[wc]
            if not bits_equal(g, w):
                bad = np.argwhere(g.view(np.int64) != w.view(np.int64))
                k, j, i = bad[0]
                raise AssertionError(f"{what}: comp {gc} box {b} {got.level.boxes[b]}: {len(bad)} cells differ, first at "
                                     f"(i,j,k)=({i},{j},{k}) local: got {g[k, j, i]!r} want {w[k, j, i]!r}")


def rel_err(got: MultiFab, want: MultiFab, gc, wc):
    """SURVEY 8(d) parity metric: |gpu-cpu| / max(|cpu|, L_inf of the component over the level)."""
    w_all = want.valid_concat(wc)
    scale = np.abs(w_all).max()
    d = np.abs(got.valid_concat(gc) - w_all)
    return float((d / np.maximum(np.abs(w_all), scale if scale > 0 else 1.0)).max())


def assert_filter_parity(got: MultiFab, want: MultiFab, comps, what, mode):
    """exact mode: bit for bit; separable mode: SURVEY 8(d)'s metric <= 1e-12 (the separable sum differs by a few ulp)"""
    if mode == "exact":
        assert_valid_bits_equal(got, want, comps, what)
    else:
        for gc, wc in comps:
            e = rel_err(got, want, gc, wc)
            assert e <= 1e-12, f"{what}: separable filter differs from the oracle by {e:.3e} (comp {gc})"


def make_states(H, ncomp, ng, fn, seed=None):
    out = []
    for lev in H.levels:
        s = MultiFab(lev, ncomp, ng, fill=0.0)
        for c in range(ncomp):
            fill_analytic(s, c, (lambda x, y, z, c=c: fn(x, y, z, c)))
        if seed is not None:
            rng = np.random.default_rng(seed + 17 * len(out))
            for b in range(lev.nboxes):
                v = s.valid(b)
                v += 1e-3 * rng.uniform(-1, 1, size=v.shape)
        # poison ghosts so an unfilled ghost cell cannot go unnoticed
        for b in range(lev.nboxes if ng else 0):
            f = s.fab(b)
            m = np.ones(f.shape[1:], bool)
            m[ng:-ng, ng:-ng, ng:-ng] = False
            f[:, m] = np.nan
        out.append(s)
    return out


CONFIGS = {
    # name: (base_n, nlev, box, is_per, sym_dir, field)
    "c1_periodic_1lev": (32, 1, 16, (1, 1, 1), (0, 0, 0), field_trig),
    "wall_1lev": (32, 1, 16, (1, 0, 1), (0, 0, 0), field_trig),
    "amr3_wall_z": (32, 3, 16, (1, 1, 0), (0, 0, 0), field_flame),
    "amr3_sym_x": (32, 3, 8, (0, 1, 1), (1, 0, 0), field_flame),
    "amr2_allwalls_ragged": (24, 2, 8, (0, 0, 0), (0, 1, 0), field_flame),
    "amr5_wall_z": (16, 5, 8, (1, 1, 0), (0, 0, 0), field_flame),  # more levels than one batched launch takes (PA_MAXB = 4)
}


def build_config(name):
    n, nlev, box, per, sym, fn = CONFIGS[name]
    H = nested_hierarchy(n, nlev, box, is_per=per)
    return H, per, sym, fn
