"""The marching-cubes triangle table (emfusion_amd/csrc/mc_tables.h): structural checks that any
correct table passes, and the digest of the reference's table (tests/golden/mc_tables_v1.json)."""
import hashlib
import json
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
CORNER = [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)]


def load_tables():
    text = (ROOT / "emfusion_amd/csrc/mc_tables.h").read_text()
    ints = lambda t: [int(v) for v in re.findall(r"-?\d+", t)]
    edges = np.array(ints(re.search(r"emf_mc_edge_corner_init\[12\]\[2\]\s*=\s*\{(.*?)\};", text, re.S).group(1)),
                     np.int64).reshape(12, 2)
    rows = re.findall(r'"([0-9ab]*)"', re.search(r"#define EMF_MC_ROWS(.*?)\n\n", text, re.S).group(1))
    assert len(rows) == 256
    tri = np.full((256, 16), -1, np.int8)  # the layout the kernels index: edges, then -1
    for c, r in enumerate(rows):
        tri[c, :len(r)] = [int(ch, 16) for ch in r]
    return edges, tri


def active_edges(cls, edges):
    return {e for e, (a, b) in enumerate(edges) if ((cls >> a) & 1) != ((cls >> b) & 1)}


def test_rows_use_exactly_the_active_edges_and_close_up():
    edges, tri = load_tables()
    for cls in range(256):
        row = tri[cls]
        n = int((row >= 0).sum())
        assert n % 3 == 0 and np.all(row[n:] == -1) and np.all(row[:n] >= 0) and np.all(row[:n] < 12)
        used = set(int(e) for e in row[:n])
        assert used == active_edges(cls, edges), cls
        tris = row[:n].reshape(-1, 3)
        assert all(len(set(t)) == 3 for t in tris.tolist()), cls
        # inside the cube every triangle side is shared by two triangles or lies on a cube face
        # (both end points on edges of one face); sides on faces are matched by the neighbour cube
        count = {}
        for t in tris.tolist():
            for i in range(3):
                k = tuple(sorted((t[i], t[(i + 1) % 3])))
                count[k] = count.get(k, 0) + 1
        for (e0, e1), c in count.items():
            corners = set(edges[e0]) | set(edges[e1])
            on_face = any(all(CORNER[c_][ax] == v for c_ in corners) for ax in range(3) for v in (0, 1))
            assert c == 2 or (c == 1 and on_face), (cls, e0, e1, c)


def test_table_matches_the_reference_digest():
    edges, tri = load_tables()
    gold = json.loads((ROOT / "tests/golden/mc_tables_v1.json").read_text())
    assert [int((r >= 0).sum() // 3) for r in tri] == gold["tris"]
    mask = np.array([sum(1 << e for e in active_edges(c, edges)) for c in range(256)], "<i4")
    assert [bin(int(m)).count("1") for m in mask] == gold["edge_bits"]
    assert hashlib.sha256(mask.tobytes()).hexdigest() == gold["edge_sha256"]  # edgeTable = active edges
    assert hashlib.sha256(tri.tobytes()).hexdigest() == gold["tri_sha256"]


def test_union_over_random_fields_is_watertight():
    """Triangles of neighbouring cubes meet edge to edge: in a closed volume interior every mesh edge
    (identified by the two grid edges it joins) is used exactly twice."""
    edges, tri = load_tables()
    rng = np.random.default_rng(5)
    n = 7
    f = rng.standard_normal((n, n, n))
    f[0], f[-1], f[:, 0], f[:, -1], f[:, :, 0], f[:, :, -1] = 1, 1, 1, 1, 1, 1  # positive shell: closed surfaces
    neg = f < 0
    count = {}
    for z in range(n - 1):
        for y in range(n - 1):
            for x in range(n - 1):
                cls = sum(int(neg[z + dz, y + dy, x + dx]) << c for c, (dx, dy, dz) in enumerate(CORNER))
                row = tri[cls]
                m = int((row >= 0).sum())

                def gid(e):  # global id of a grid edge: its two lattice end points
                    a, b = (tuple(np.add((x, y, z), CORNER[c])) for c in edges[e])
                    return (min(a, b), max(a, b))
                for t in row[:m].reshape(-1, 3).tolist():
                    for i in range(3):
                        k = tuple(sorted((gid(t[i]), gid(t[(i + 1) % 3]))))
                        count[k] = count.get(k, 0) + 1
    assert count and all(c == 2 for c in count.values())
