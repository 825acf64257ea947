#!/usr/bin/env python3
"""Golden vectors for the distance function, produced by the REFERENCE's own make_level_set3
(Tools/SDFGen/makelevelset3.cpp compiled in place into oracle/_ref/libsdfgen_ref.so by
`make -C oracle ref`).  Only runs where /root/reference exists; the .npz holds inputs and the
reference's outputs (data, no source):
    python tests/golden/make_golden_sdf.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as O  # noqa: E402
from sdf_cases import cases  # noqa: E402


def main():
    O.build()
    if O.sdf_ref_lib() is None:
        raise SystemExit("oracle/_ref/libsdfgen_ref.so is missing and /root/reference is not available")
    d = {}
    names = []
    for c in cases(O):
        phi = O.sdf_level_set_ref(c["tris"], c["verts"], c["origin"], c["dx"], c["n"], c["band"])
        k = c["name"]
        names.append(k)
        d[k + "_tris"], d[k + "_verts"] = c["tris"], c["verts"]
        d[k + "_origin"] = np.array(c["origin"], dtype=np.float32)
        d[k + "_dx"] = np.float32(c["dx"])
        d[k + "_n"] = np.array(c["n"], dtype=np.int32)
        d[k + "_band"] = np.int32(c["band"])
        d[k + "_phi_ref"] = phi
    d["names"] = np.array(names)
    out = os.path.join(HERE, "sdf_ref.npz")
    np.savez_compressed(out, **d)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
