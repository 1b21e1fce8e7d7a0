"""Digest of the marching-cubes lookup tables the reference indexes (src/core/cuda/TSDF.cu:31-324).

Run in the build container (needs /root/reference); writes tests/golden/mc_tables_v1.json holding only
digests and per-class counts -- expected values for tests/test_mc_tables.py, no table text:
  tri_sha256   sha256 of triTable as 256 x 16 int8, row-major
  edge_sha256  sha256 of edgeTable as 256 little-endian int32
  tris         number of triangles per cube class
  edge_bits    popcount of edgeTable per cube class
"""
import hashlib
import json
import re
import sys
from pathlib import Path

import numpy as np

SRC = Path("/root/reference/src/core/cuda/TSDF.cu")


def table(text, name):
    m = re.search(r"const\s+int\s+" + name + r"\s*(\[\d+\])+\s*=\s*\{(.*?)\};", text, re.S)
    return [int(v, 0) for v in re.findall(r"-?(?:0x[0-9a-fA-F]+|\d+)", m.group(2))]


def main():
    text = SRC.read_text()
    tri = np.array(table(text, "triTable"), np.int8).reshape(256, 16)
    edge = np.array(table(text, "edgeTable"), "<i4")
    assert edge.shape == (256,)
    out = {
        "source": "EmbodiedVision/emfusion src/core/cuda/TSDF.cu:31-324",
        "tri_sha256": hashlib.sha256(tri.tobytes()).hexdigest(),
        "edge_sha256": hashlib.sha256(edge.tobytes()).hexdigest(),
        "tris": [int((row >= 0).sum() // 3) for row in tri],
        "edge_bits": [bin(int(e)).count("1") for e in edge],
    }
    Path(__file__).with_name("mc_tables_v1.json").write_text(json.dumps(out))
    print(out["tri_sha256"], out["edge_sha256"], sum(out["tris"]))


if __name__ == "__main__":
    sys.exit(main())
