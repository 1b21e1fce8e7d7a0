"""Committed regression vectors for the "next row" kernels (SURVEY.md section 8 f-2 .. f-4): meshes,
Phong rendering, depth pre-processing.  Same provenance as make_golden.py: outputs of THIS repository's
CPU oracle on inputs taken from kernels_v1.npz, pinning the oracle against drift and the HIP path
against reviewable data -- not reference outputs.

Usage (repo root, CPU only):  python tests/golden/make_golden_next_rows.py  ->  next_rows_v1.npz"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding as orc  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    kv = np.load(OUT / "kernels_v1.npz")
    g = {}
    for tag in ("cube", "ragged"):
        tsdf, wts, voxel = kv[f"{tag}_tsdf2"], kv[f"{tag}_wts2"], float(kv[f"{tag}_voxel"])
        fg = kv[f"{tag}_fgmask"] if f"{tag}_fgmask" in kv.files else None
        v, n, t = orc.marching_cubes(tsdf, wts, voxel)
        g[f"{tag}_mesh_v"], g[f"{tag}_mesh_n"], g[f"{tag}_mesh_t"] = v, n, t
        if fg is not None:
            v, n, t = orc.marching_cubes(tsdf, wts, voxel, fg=fg)
            g[f"{tag}_fgmesh_v"], g[f"{tag}_fgmesh_n"], g[f"{tag}_fgmesh_t"] = v, n, t
    # rendering: the raycast outputs of the cube volume as vertex / normal maps, labels in stripes
    rng = np.random.default_rng(77)
    pts, nrm = kv["cube_vert0"], kv["cube_nrm0"]
    h, w = pts.shape[:2]
    seg = ((np.arange(w)[None, :] // 8 + np.arange(h)[:, None] // 8) % 5).astype(np.uint8)
    cmap = rng.integers(0, 256, (256, 3)).astype(np.uint8)
    g["render_seg"], g["render_cmap"] = seg, cmap
    g["render_rgb"] = orc.render_phong(pts, nrm, seg, cmap)
    g["render_rgb_light"] = orc.render_phong(pts, nrm, seg, cmap, (0.3, -0.2, 0.1))
    # depth pre-processing on a kernel-vector depth image with holes and an outlier
    d = kv["cube_depth0"].copy()
    d[10:14, 20:30] = 0
    d[40, 50] = np.float32(9.5)
    g["prep_in"] = d
    g["prep_out"] = orc.preprocess_depth(d, 7, 0.04, 4.5)
    np.savez_compressed(OUT / "next_rows_v1.npz", **g)
    print({k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    sys.exit(main())
