"""Generator of the committed regression vectors under tests/golden/ (SURVEY.md section 8c).

PROVENANCE -- read this before trusting the numbers: these vectors are outputs of THIS repository's
CPU oracle (oracle/emf_oracle.c, built with -ffp-contract=off), not of the reference's CUDA build.
The reference cannot be built or run here (no CUDA, no OpenCV; the oracle header says "parity
unpinned"), it ships no tests and no golden data.  What the vectors pin is (a) the oracle against
drift -- any edit that changes its results fails tests/test_golden.py -- and (b) the HIP path against
a fixed, reviewable set of inputs and outputs that travels to the GPU box as plain data.

Usage (from the repo root, CPU only):  python tests/golden/make_golden.py
Writes tests/golden/kernels_v1.npz (IEEE-exact stages: points, integration, gradients, raycast,
trilinear lookups, fg/bg counts) and tests/golden/frames_v1.npz (4 frames of the full schedule).
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import binding as orc  # noqa: E402
from tests.oracle_pipeline import Affine32, OraclePipeline  # noqa: E402
from tests.scenes import Pose, camera_path, intrinsics, rel_CO, rel_OC, render_depth, rot  # noqa: E402

OUT = Path(__file__).resolve().parent
f32 = np.float32

W, H = 96, 72
K = intrinsics(W, H)
SPHERES = [((0.25, 0.05, 1.3), 0.22), ((-0.3, -0.1, 1.6), 0.18)]


def vol(n, ch=1, dt=np.float32):
    return np.zeros((n[2], n[1], n[0]) if ch == 1 else (n[2], n[1], n[0], ch), dt)


def kernels(fma=False, save=True):
    """Stages whose arithmetic is +,-,*,/,sqrt only: results are machine independent.
    fma=True: the same sequence on the oracle built with a*b+c contraction on (make_golden_fma.py)."""
    g = {"K": K.astype(f32)}
    rng = np.random.default_rng(20240917)
    for tag, n, voxel in (("cube", (32, 32, 32), 0.08), ("ragged", (30, 22, 18), 0.085)):
        pose = Pose(t=[0, 0, 1.28])
        tsdf, wts = vol(n), vol(n)
        for i in range(3):
            cam = camera_path(i)
            depth, ids = render_depth(W, H, K, cam, SPHERES, noise=0.002, dropout=0.02, seed=300 + i)
            assoc = rng.uniform(0, 1, (H, W)).astype(f32)
            assoc[rng.uniform(size=(H, W)) < 0.05] = 0  # exercises w + a == 0 on unseen voxels
            oc = rel_OC(cam, pose)
            orc.update_tsdf(depth, assoc, tsdf, wts, oc.R32, oc.t32, K, voxel, 10 * voxel, 3.0, fma=fma)
            g[f"{tag}_depth{i}"], g[f"{tag}_assoc{i}"] = depth, assoc
            g[f"{tag}_Roc{i}"], g[f"{tag}_toc{i}"] = oc.R32, oc.t32
            if i != 1:  # frame 1 is covered through frame 2 (max weight 3: the cap is reached)
                g[f"{tag}_tsdf{i}"], g[f"{tag}_wts{i}"] = tsdf.copy(), wts.copy()
        g[f"{tag}_res"] = np.array(n, np.int32)
        g[f"{tag}_voxel"] = f32(voxel)
        g[f"{tag}_trunc"] = f32(10 * voxel)  # the float32 the kernels receive
        g[f"{tag}_grads"] = orc.compute_tsdf_grads(tsdf, fma=fma)
        fg = np.zeros(tsdf.shape, np.uint8)  # foreground = the lower-x half, with a ragged edge
        fg[:, :, : n[0] // 2] = 255
        fg[rng.uniform(size=tsdf.shape) < 0.1] = 0
        g[f"{tag}_fgmask"] = fg
        for j, cam in enumerate((camera_path(3), Pose(rot([0.3, 1, 0.2], 14), [0.2, -0.1, 0.15]),
                                 Pose(rot([0, 1, 0], -8), [0.0, 0.0, 0.6]))):  # last: inside the box
            co = rel_CO(cam, pose)
            for name, mask in ((("", None), ("_fg", fg)) if j == 0 else (("", None),)):
                ray, vert, nrm, hit, steps = orc.raycast_tsdf(tsdf, None, wts, mask, W, H, co.R32, co.t32,
                                                              K, voxel, 10 * voxel, count_steps=True, fma=fma)
                g[f"{tag}_ray{j}{name}"], g[f"{tag}_vert{j}{name}"] = ray, vert
                g[f"{tag}_nrm{j}{name}"], g[f"{tag}_hit{j}{name}"] = nrm, hit
                g[f"{tag}_steps{j}{name}"] = np.int64(steps.sum())
            g[f"{tag}_Rco{j}"], g[f"{tag}_tco{j}"] = co.R32, co.t32
        # trilinear lookups of 1 and 3 channels at the points of frame 2
        pts = orc.compute_points(g[f"{tag}_depth2"], K, fma=fma)
        co = rel_CO(camera_path(2), pose)
        g[f"{tag}_points"] = pts
        g[f"{tag}_vals1"] = orc.get_volume_vals(tsdf, pts, co.R32, co.t32, voxel, fma=fma)
        g[f"{tag}_vals3"] = orc.get_volume_vals(g[f"{tag}_grads"], pts, co.R32, co.t32, voxel, fma=fma)
        # fg / bg counts and probabilities
        fgbg = vol(n, 2)
        m = (render_depth(W, H, K, camera_path(2), SPHERES, noise=0, dropout=0, seed=1)[1] == 1).astype(np.uint8)
        occ = (rng.uniform(size=(H, W)) < 0.1).astype(np.uint8) * 255
        oc = rel_OC(camera_path(2), pose)
        orc.update_fgbg_probs(m, occ, tsdf, wts, fgbg, oc.R32, oc.t32, K, voxel, fma=fma)
        g[f"{tag}_mask"], g[f"{tag}_occluded"], g[f"{tag}_fgbg"] = m, occ, fgbg
        probs, vmask = orc.compute_fg_probs(fgbg, fma=fma)
        g[f"{tag}_probs"], g[f"{tag}_vmask"] = probs, vmask
    if save:
        np.savez_compressed(OUT / "kernels_v1.npz", **g)
    return g


FR = dict(bg_res=32, bg_voxel=0.08, obj_res=16, vis=30, boundary=3, frames=4, mask_every=2)


def frames():
    """Four frames of the whole schedule (E-step x3, raycast + compositing, weighted integration,
    mask integration).  The association weights pass through expf, so consumers compare these with
    a tolerance (tests/test_golden.py)."""
    g = {"K": K.astype(f32)}
    centers = [(np.array(c, f32), f32(2 * r * 1.3)) for c, r in SPHERES]
    pipe = OraclePipeline(orc, W, H, K, FR["bg_res"], FR["bg_voxel"], [0, 0, 1.28], FR["obj_res"],
                          visibility_thresh=FR["vis"], boundary=FR["boundary"])
    ids = [pipe.add_object(c, s) for c, s in centers]
    for k, (c, s) in enumerate(centers):
        g[f"obj{k + 1}_center"], g[f"obj{k + 1}_size"] = c, s
    for f in range(FR["frames"]):
        cam = camera_path(f)
        depth, sid = render_depth(W, H, K, cam, SPHERES, noise=0.002, dropout=0.01, seed=500 + f)
        run_masks = f % FR["mask_every"] == 0
        masks = {i: (sid == i).astype(np.uint8) for i in ids} if run_masks else {}
        poses = {i: Affine32(np.eye(3, dtype=f32), centers[i - 1][0] + f32(0.004 * f)) for i in ids}
        pipe.process_frame(depth, Affine32(cam.R.astype(f32), cam.t.astype(f32)), poses, masks, run_masks)
        g[f"f{f}_depth"], g[f"f{f}_camR"], g[f"f{f}_camt"] = depth, cam.R.astype(f32), cam.t.astype(f32)
        for i in ids:
            g[f"f{f}_obj{i}_t"] = poses[i].t
            if run_masks:
                g[f"f{f}_obj{i}_mask"] = masks[i]
        g[f"f{f}_ray"], g[f"f{f}_seg"] = pipe.ray.copy(), pipe.seg.copy()
        g[f"f{f}_norm"], g[f"f{f}_bg_assoc"] = pipe.norm.copy(), pipe.bg_assoc.copy()
        g[f"f{f}_vis"] = np.array(sorted(pipe.vis), np.int32)
    g["bg_tsdf"], g["bg_wts"] = pipe.bg["tsdf"], pipe.bg["wts"]
    for v in pipe.objects:
        i = v["id"]
        g[f"obj{i}_tsdf"], g[f"obj{i}_wts"], g[f"obj{i}_probs"] = v["tsdf"], v["wts"], v["probs"]
        g[f"obj{i}_vmask"], g[f"obj{i}_assoc"] = v["vmask"], v["assoc"]
    g["points"] = pipe.points
    np.savez_compressed(OUT / "frames_v1.npz", **g)
    return g


if __name__ == "__main__":
    orc.lib()
    orc.set_threads(4)
    a, b = kernels(), frames()
    for name in ("kernels_v1.npz", "frames_v1.npz"):
        print(name, (OUT / name).stat().st_size, "bytes")
    print("kernel vectors:", len(a), "arrays; frame vectors:", len(b), "arrays")
