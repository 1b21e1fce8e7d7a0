"""The long full-size stream of VERDICT r03 "next round" item 1, and the runner that drives the HIP host
classes and the frame-level oracle (tests/oracle_pipeline.py) through it side by side.

Test infrastructure.  configs[1] of BASELINE.json (640 x 480, background 512^3 @ 1 cm + 4 objects 128^3,
maxWeight 64), 82 frames of the whole schedule with supplied poses (reference EMFusion.cpp:70-129,
integrateDepth 865-889, weight cap TSDF.cu:382-400) arranged so that every piece of state the native path
keeps BESIDE the volumes is exercised the way a long run exercises it:

  frames  0-19   camera on a 5 cm circle with a small rotation (the bench's motion);
                 object 3 slides behind object 2 (frames 4-12), stays hidden (12-22: its visibility
                 count falls under the threshold -> gated out of integrateDepth) and comes back (22-30)
  frames 20-33   the camera yaws right by 2 deg / frame to 28 deg: object 1 (at 25 deg left) leaves the
                 view, space never seen before enters it
  frames 34-45   held at 28 deg (12 frames "away", with a small wobble)
  frames 46-59   the camera swings back: it re-enters space it integrated 26-40 frames earlier,
                 object 1 re-enters the view
  frames 60-81   the first motion again; free space observed in every frame reaches the weight cap
                 (64) at frame 63: frame 65 is the first checkpoint past it
  masks          on frames 0, 30, 60 (maskRCNNFrames = 30) for the objects that cover > 1600 px

Check points (after processing the frame): 1, 40, 65, 80.
"""
from __future__ import annotations

import os
import time

import numpy as np
import xxhash

from tests.oracle_pipeline import Affine32, OraclePipeline
from tests.scenes import Pose, intrinsics, render_depth, rot

W, H = 640, 480
BG_RES, BG_VOX, OBJ_RES, NOBJ = 512, 0.01, 128, 4
NFRAMES = 82
CHECKPOINTS = (1, 40, 65, 80)
MASK_EVERY = 30

# the alternative execution paths of emf::EMFusion that must produce the same bytes
PATHS = (("per-volume launches", {"EMF_PER_VOLUME": "1"}),
         ("IEEE divisions, inline 1/lambda", {"EMF_VOXEL_RCP": "0", "EMF_LAMBDA_TABLE": "0"}),
         ("background integrated in place after the raycast", {"EMF_BG_OVERLAP": "0"}),
         ("every ray marched to the end of its range", {"EMF_FAR_BOUNDS": "0"}),
         ("every tile of the sweep loaded, one-level launch",
          {"EMF_UNSEEN_TILES": "0", "EMF_DEEP_TILES": "0", "EMF_INT_CULL": "0"}),
         ("four lanes per background ray (march_quad)", {"EMF_MARCH_ROWS": "4"}))

_REST = {1: ((-0.75, -0.10, 1.60), 0.22), 2: ((0.10, 0.00, 1.50), 0.25),
         3: ((0.65, 0.02, 2.00), 0.16), 4: ((0.70, 0.30, 1.80), 0.20)}
_HIDDEN_X = 0.133  # object 3 straight behind object 2 as seen from the origin


def yaw_deg(f: int) -> float:
    if f < 20:
        return 0.0
    if f < 34:
        return 2.0 * (f - 19)
    if f < 46:
        return 28.0
    if f < 60:
        return 28.0 - 2.0 * (f - 45)
    return 0.0


def camera_pose(f: int) -> Pose:
    a = 2 * np.pi * f / 90.0
    t = np.array([0.05 * np.cos(a) - 0.05, 0.05 * np.sin(a), 0.0])
    wobble = 0.4 * np.sin(2 * np.pi * f / 23.0)
    return Pose(rot([0.0, 1.0, 0.0], yaw_deg(f)) @ rot([0.2, 1.0, 0.1], wobble), t)


def sphere(k: int, f: int):
    """(centre (3,) float32, radius, object volume edge) of object id k at frame f."""
    c0, r = _REST[k]
    c = np.array(c0, np.float64)
    if k == 3:
        if f <= 4:
            s = 0.0
        elif f < 12:
            s = (f - 4) / 8.0
        elif f <= 22:
            s = 1.0
        elif f < 30:
            s = 1.0 - (f - 22) / 8.0
        else:
            s = 0.0
        c[0] = c0[0] + s * (_HIDDEN_X - c0[0])
    else:
        c += 0.03 * np.array([np.sin(0.07 * f + k), 0.5 * np.sin(0.05 * f + 2 * k), np.sin(0.06 * f + 3 * k)])
    return c.astype(np.float32), r, np.float32(3.2 * r)


def frame(f: int):
    """depth (H, W) f32, instance ids (H, W) u8, camera (R9, t3) f32, object poses {id: (R9, t3)}."""
    K = intrinsics(W, H)
    cam = camera_pose(f)
    sph = [(sphere(k, f)[0].astype(np.float64), sphere(k, f)[1]) for k in range(1, NOBJ + 1)]
    depth, ids = render_depth(W, H, K, cam, sph, noise=0.002, dropout=0.01, seed=1000 + f)
    poses = {k: (np.eye(3, dtype=np.float32).reshape(-1), sphere(k, f)[0]) for k in range(1, NOBJ + 1)}
    return depth, ids.astype(np.uint8), (cam.R32, cam.t32), poses


def frame_masks(f: int, ids: np.ndarray):
    if f % MASK_EVERY:
        return {}
    out = {}
    for k in range(1, NOBJ + 1):
        m = (ids == k).astype(np.uint8)
        if int(m.sum()) > 1600:
            out[k] = m
    return out


def digest(a: np.ndarray) -> str:
    return xxhash.xxh3_128(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()


def compare(got: np.ndarray, want: np.ndarray, rtol=1e-4, atol=0.0):
    """(fraction bit-identical, fraction outside |a-b| <= rtol max(|a|,|b|) + atol, worst abs diff) without
    float64 copies of 134 M-element volumes: the tolerance is evaluated on the differing elements only."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    if got.dtype.kind != "f":
        ne = got != want
        return 1.0 - float(ne.mean()), float(ne.mean()), float(ne.any())
    ne = got != want
    idx = np.flatnonzero(ne.reshape(-1))
    if idx.size == 0:
        return 1.0, 0.0, 0.0
    a = got.reshape(-1)[idx].astype(np.float64)
    b = want.reshape(-1)[idx].astype(np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    with np.errstate(invalid="ignore"):
        bad = np.abs(a - b) > rtol * np.maximum(np.abs(a), np.abs(b)) + atol
    bad |= np.isnan(a) ^ np.isnan(b)
    bad &= ~both_nan
    d = np.abs(a - b)
    worst = float(np.nanmax(d)) if np.isfinite(d).any() else 0.0
    n = got.size
    return 1.0 - float((idx.size - both_nan.sum()) / n), float(bad.sum() / n), worst


class Frames:
    """The stream, rendered once (numpy, ~0.1 s per frame) and shared by every run of a module."""

    def __init__(self, nframes=NFRAMES):
        self.items = [frame(f) for f in range(nframes)]

    def __len__(self):
        return len(self.items)


def new_fusion(env=None, comm=None):
    from emfusion_amd import pipeline
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        prm = pipeline.make_params(W, H, BG_RES, BG_VOX, OBJ_RES, mask_frames=MASK_EVERY)
        fus = pipeline.Fusion(prm, comm)
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)
    ids = [fus.add_object(sphere(k, 0)[0], float(sphere(k, 0)[2])) for k in range(1, NOBJ + 1)]
    assert ids == list(range(1, NOBJ + 1))
    return fus, prm


def hip_step(fus, frames: Frames, f: int):
    from tests.parity_util import to_dev
    from emfusion_amd.ops import image_view
    depth, ids, (R, t), poses = frames.items[f]
    masks = frame_masks(f, ids)
    d_depth = to_dev(depth)
    d_masks = {i: to_dev(m) for i, m in masks.items()}
    fus.process_frame(image_view(d_depth), R, t, poses, {i: image_view(m) for i, m in d_masks.items()}, bool(masks))
    fus.synchronize()
    return masks


def snapshot_digests(fus):
    out = {}
    for which in ("tsdf", "weights"):
        out[f"bg {which}"] = digest(fus.volume(which, 0))
        for i in range(1, NOBJ + 1):
            out[f"obj {i} {which}"] = digest(fus.volume(which, i))
    for i in range(1, NOBJ + 1):
        out[f"obj {i} fgprobs"] = digest(fus.volume("fgprobs", i))
        out[f"obj {i} assoc"] = digest(fus.image("obj_assoc", i))
        out[f"obj {i} raylengths"] = digest(fus.image("obj_raylengths", i))
    for im in ("raylengths", "segmentation", "assoc_norm", "bg_assoc", "bg_raylengths", "vertices", "normals"):
        out[im] = digest(fus.image(im))
    return out


def run_path(frames: Frames, env=None, checkpoints=CHECKPOINTS):
    """The stream through emf::EMFusion alone: visible sets per frame + digests at the check points."""
    fus, _ = new_fusion(env)
    vis, dig = [], {}
    for f in range(len(frames)):
        hip_step(fus, frames, f)
        vis.append(sorted(fus.visible_objects()))
        if f in checkpoints:
            dig[f] = snapshot_digests(fus)
    fus.close()
    return dict(visible=vis, digests=dig)


def run_against_oracle(oracle, frames: Frames, checkpoints=CHECKPOINTS, log=print):
    """HIP host classes (default path) and the frame-level oracle side by side; returns the per-frame
    visible sets, the check points' comparison records and the default path's digests."""
    oracle.set_threads(oracle.host_threads())
    fus, prm = new_fusion()
    K = np.array(prm.K, np.float32)
    orc = OraclePipeline(oracle, W, H, K, BG_RES, BG_VOX, list(prm.volume_pose_t), OBJ_RES)
    for k in range(1, NOBJ + 1):
        assert orc.add_object(sphere(k, 0)[0], sphere(k, 0)[2]) == k
    rec = dict(visible=[], oracle_visible=[], checks={}, digests={}, pixels=[], t_oracle=0.0, t_hip=0.0,
               t_checks=0.0, oracle_threads=oracle.host_threads())
    for f in range(len(frames)):
        depth, ids, (R, t), poses = frames.items[f]
        t0 = time.time()
        masks = hip_step(fus, frames, f)
        rec["t_hip"] += time.time() - t0
        t0 = time.time()
        orc.process_frame(depth, Affine32(R.reshape(3, 3), t),
                          {i: Affine32(p[0].reshape(3, 3), p[1]) for i, p in poses.items()}, masks, bool(masks))
        rec["t_oracle"] += time.time() - t0
        rec["visible"].append(sorted(fus.visible_objects()))
        rec["oracle_visible"].append(sorted(orc.vis))
        rec["pixels"].append([int((ids == k).sum()) for k in range(1, NOBJ + 1)])
        if f in checkpoints:
            c, t0 = {}, time.time()

            def put(name, got, want, rtol=1e-4, atol=0.0):
                c[name] = compare(got, want, rtol, atol)
            put("bg tsdf", fus.volume("tsdf", 0), orc.bg["tsdf"], atol=1e-6)
            w = fus.volume("weights", 0)
            put("bg weights", w, orc.bg["wts"], atol=1e-6)
            c["_bg_seen"] = int((orc.bg["wts"] > 0).sum())
            c["_bg_capped"] = int((orc.bg["wts"] >= 64.0).sum())
            c["_bg_capped_hip"] = int((w >= 64.0).sum())
            del w
            for v in orc.objects:
                i = v["id"]
                put(f"obj {i} tsdf", fus.volume("tsdf", i), v["tsdf"], atol=1e-6)
                put(f"obj {i} weights", fus.volume("weights", i), v["wts"], atol=1e-6)
                put(f"obj {i} fgprobs", fus.volume("fgprobs", i), v["probs"], atol=1e-6)
                put(f"obj {i} fgmask", fus.volume("fgmask", i), v["vmask"])
                put(f"obj {i} assoc", fus.image("obj_assoc", i), v["assoc"], atol=1e-7)
                put(f"obj {i} raylengths", fus.image("obj_raylengths", i), v["ray"])
                c[f"_obj {i} seen"] = int((v["wts"] > 0).sum())
            put("points", fus.image("points"), orc.points)
            put("assoc_norm", fus.image("assoc_norm"), orc.norm)
            put("bg_assoc", fus.image("bg_assoc"), orc.bg_assoc, atol=1e-7)
            seg = fus.image("segmentation")
            put("segmentation", seg, orc.seg)
            same = seg == orc.seg
            put("raylengths", fus.image("raylengths")[same], orc.ray[same])
            put("bg_raylengths", fus.image("bg_raylengths"), orc.bg_ray)
            hit = (orc.ray > 0) & same
            put("normals", fus.image("normals")[hit], orc.nrm[hit], rtol=1e-3, atol=1e-4)
            c["_object_pixels"] = int((orc.seg > 0).sum())
            c["_hits"] = int((orc.ray > 0).sum())
            c["_bg_hits"] = int((orc.bg_ray > 0).sum())
            total = fus.image("bg_assoc").astype(np.float64)
            for i in range(1, NOBJ + 1):
                total += fus.image("obj_assoc", i)
            valid = orc.norm != 0
            c["_assoc_sums_to_one"] = bool(np.allclose(total[valid], 1.0, atol=1e-5) and np.all(total[~valid] == 0))
            rec["checks"][f] = c
            rec["digests"][f] = snapshot_digests(fus)
            rec["t_checks"] += time.time() - t0
            log(f"frame {f}: " + ", ".join(f"{k} {v[0]:.4f}/{v[1]:.2e}" for k, v in c.items() if not k.startswith("_")))
    fus.close()
    oracle.set_threads(min(8, os.cpu_count() or 1))
    return rec
