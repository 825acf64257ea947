"""Marching-cubes tables: digests pinned in SURVEY.md 8(c) (computed from isosurface.cpp:451-741 in
the development container) + structural properties of the Lorensen/Bourke tables."""
import hashlib
import struct

import numpy as np

EDGE_SHA = "ffc58719f11be7a8b34988740a15dcd043dc314a3e2fe01e917fd796001815b9"
TRI_SHA = "85e6eb7486ad0101a95aaf3b332a15ba6e5d187a24d31c5eb77874d3aa45d996"

EDGE_ENDS = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def test_table_digests(oracle):
    e, t = oracle.mc_tables()
    assert hashlib.sha256(struct.pack("<256i", *e.tolist())).hexdigest() == EDGE_SHA
    assert hashlib.sha256(struct.pack("<4096i", *t.ravel().tolist())).hexdigest() == TRI_SHA


def test_table_structure(oracle):
    e, t = oracle.mc_tables()
    ntri = 0
    for c in range(256):
        row = t[c]
        n = int(np.argmax(row == -1)) if (row == -1).any() else 16
        assert n % 3 == 0 and n <= 15 and np.all(row[n:] == -1)
        ntri += n // 3
        used = 0
        for q in row[:n]:
            used |= 1 << int(q)
        assert used == e[c], "edgeTable mask = OR of the edges its triTable row uses"
        # an edge is flagged iff its two corners lie on different sides
        for k, (a, b) in enumerate(EDGE_ENDS):
            assert bool(e[c] & (1 << k)) == (((c >> a) & 1) != ((c >> b) & 1))
        assert e[c] == e[255 - c]
    assert ntri == 820
