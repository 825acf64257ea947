#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz with the CPU oracle.

The reference ships no golden vectors and cannot be built here (PARITY UNPINNED), so these fixtures
pin the ORACLE against itself over time (regression) and give the GPU tests fixed input/expected
pairs that travel to the GPU box as plain data.  Re-run only when the restated algorithm changes:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as O  # noqa: E402
from peleanalysis_amd.hierarchy import MultiFab, field_flame, nested_hierarchy  # noqa: E402
from util import make_states  # noqa: E402


def level_arrays(H):
    d = {"nlev": np.int32(H.nlev)}
    for l, lv in enumerate(H.levels):
        d[f"boxes{l}"] = lv.boxes
        d[f"dom{l}"] = np.concatenate([lv.domlo, lv.domhi, lv.is_per]).astype(np.int32)
    return d


def main():
    O.build()
    # --- grad + curvature on a 3-level hierarchy (base 16^3, 8^3 boxes), wall in z, threshold on
    H = nested_hierarchy(16, 3, 8, is_per=(1, 1, 0))
    states = make_states(H, 1, 2, field_flame, seed=101)
    bc = O.bc_from_flags((1, 1, 0))
    og = [MultiFab(lv, 4, 0) for lv in H.levels]
    oc = [MultiFab(lv, 5, 0) for lv in H.levels]
    O.grad_pipeline(H.levels, [s.copy() for s in states], 0, bc, og, 0)
    pm = O.curvature_pipeline(H.levels, [s.copy() for s in states], 0, bc, oc, 0, MultiFab, threshold=0.02)
    d = level_arrays(H)
    d["prog_minmax"] = np.array(pm)
    d["threshold"] = np.float64(0.02)
    for l in range(H.nlev):
        d[f"in{l}"] = np.stack([states[l].valid(b)[0] for b in range(H.levels[l].nboxes)])
        d[f"grad{l}"] = np.stack([og[l].valid(b) for b in range(H.levels[l].nboxes)])
        d[f"curv{l}"] = np.stack([oc[l].valid(b) for b in range(H.levels[l].nboxes)])
    np.savez_compressed(os.path.join(HERE, "gradcurv_amr3.npz"), **d)

    # --- box filter, 2 levels, non-periodic, cell-conservative interpolation
    H = nested_hierarchy(16, 2, 8, is_per=(0, 0, 0))
    ins = make_states(H, 1, 2, field_flame, seed=202)
    outs = [MultiFab(lv, 1, 0) for lv in H.levels]
    O.filter_pipeline(H.levels, [s.copy() for s in ins], outs, 1, base_fgr=2, interp_type=1)
    d = level_arrays(H)
    for l in range(H.nlev):
        d[f"in{l}"] = np.stack([ins[l].valid(b)[0] for b in range(H.levels[l].nboxes)])
        d[f"out{l}"] = np.stack([outs[l].valid(b)[0] for b in range(H.levels[l].nboxes)])
    np.savez_compressed(os.path.join(HERE, "filter_amr2.npz"), **d)

    # --- isosurface, 2 levels
    H = nested_hierarchy(16, 2, 8, is_per=(0, 0, 0))
    fields = make_states(H, 2, 0, field_flame, seed=303)
    nodes, elts = O.isosurface_pipeline(H.levels, fields, [0, 1], 0, 1150.0, MultiFab)
    d = level_arrays(H)
    for l in range(H.nlev):
        d[f"in{l}"] = np.stack([fields[l].valid(b) for b in range(H.levels[l].nboxes)])
    d["isoval"] = np.float64(1150.0)
    d["nodes"] = nodes
    d["elts"] = elts
    np.savez_compressed(os.path.join(HERE, "iso_amr2.npz"), **d)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
