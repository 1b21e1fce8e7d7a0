"""Committed regression vectors for the tracker: inputs and the Levenberg-Marquardt
history of tests/oracle_tracking.py (the restatement of TSDF.cpp:170-344 over the CPU oracle kernels) on a small
scene -- two volumes, a 96 x 72 frame, a start pose that is off by ~1.5 cm / 0.6 degrees.  Same provenance as
make_golden.py: outputs of THIS repository's oracle, pinning it against drift and giving the HIP tracker
reviewable numbers -- not reference outputs (the reference cannot be built here: parity unpinned).

Usage (repo root, CPU only):  python tests/golden/make_golden_tracking.py  ->  tracking_v1.npz"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import binding as orc  # noqa: E402
from tests.oracle_tracking import OracleTracker, orthonormalise  # noqa: E402
from tests.scenes import Pose, camera_path, intrinsics, rel_CO, rel_OC, render_depth, rot  # noqa: E402

OUT = Path(__file__).resolve().parent
W, H = 96, 72
SPHERES = [((0.25, 0.05, 1.3), 0.22), ((-0.3, -0.1, 1.6), 0.18)]
VOLS = [dict(n=(48, 48, 48), vox=0.05, pose=Pose(t=[0, 0, 1.2])), dict(n=(32, 32, 32), vox=0.02, pose=Pose(t=SPHERES[0][0]))]
ITER = (1, 3, 40)


def build():
    K = intrinsics(W, H)
    g = {"K": np.asarray(K, np.float32)}
    vols = []
    for k, v in enumerate(VOLS):
        n = v["n"]
        tsdf, wts = np.zeros((n[2], n[1], n[0]), np.float32), np.zeros((n[2], n[1], n[0]), np.float32)
        for i in range(4):
            cam = camera_path(i)
            depth, _ = render_depth(W, H, K, cam, SPHERES, noise=0.002, dropout=0.01, seed=300 + i)
            oc = rel_OC(cam, v["pose"])
            orc.update_tsdf(depth, np.ones((H, W), np.float32), tsdf, wts, oc.R32, oc.t32, K, v["vox"], 10 * v["vox"], 64.0)
        vols.append((tsdf, wts))
        g[f"m{k}_tsdf"], g[f"m{k}_wts"], g[f"m{k}_vox"] = tsdf, wts, np.float32(v["vox"])
    cam = camera_path(5)
    depth, _ = render_depth(W, H, K, cam, SPHERES, noise=0.002, dropout=0.01, seed=305)
    g["points"] = orc.compute_points(depth, K)
    rng = np.random.default_rng(31)
    guess = cam * Pose(rot([0.2, 1.0, 0.3], 0.6), [0.012, -0.006, 0.008])
    for k, v in enumerate(VOLS):
        assoc = np.ones((H, W), np.float32) if k == 0 else rng.uniform(0.2, 1.0, (H, W)).astype(np.float32)
        g[f"m{k}_assoc"] = assoc
        co = rel_CO(guess, v["pose"])
        R0, t0 = orthonormalise(co.R32.reshape(3, 3)).reshape(-1), co.t32
        g[f"m{k}_R0"], g[f"m{k}_t0"] = R0, t0
        for n_it in ITER:
            tr = OracleTracker(orc, vols[k][0], vols[k][1], v["vox"])
            tr.prepare(R0, t0)
            for _ in range(n_it):
                tr.iterate(g["points"], assoc)
            h = tr.history[-1]
            p = f"m{k}_it{n_it}_"
            g[p + "R"], g[p + "t"] = np.asarray(tr.R, np.float32), np.asarray(tr.t, np.float32)
            g[p + "mu"] = np.float32(tr.mu)
            g[p + "counts"] = np.array([tr.iterations, tr.accepted, int(tr.converged)], np.int32)
            g[p + "A"], g[p + "b"], g[p + "x"] = h["A"], h["b"], h["x"]
            g[p + "err"] = np.array([h["err"], h["err_new"], h["rho"]], np.float64)
    return g


def main():
    g = build()
    np.savez_compressed(OUT / "tracking_v1.npz", **g)
    print({k: (v.shape, v.dtype) for k, v in g.items() if "it" in k and k.startswith("m0")})
    print("bytes", (OUT / "tracking_v1.npz").stat().st_size)


if __name__ == "__main__":
    sys.exit(main())
