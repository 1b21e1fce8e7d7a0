"""A full-size stand-in for BASELINE.json configs[2] (TUM fr3/walking_xyz with config/tum.cfg; the dataset cannot be
staged here): an analytic indoor scene rendered at 640 x 480 and written in the TUM RGB-D layout the reference's reader
takes (src/utils/TUMRGBDReader.cpp: associations.txt, depth PNGs x 5000) with the preprocessed masks its
`--maskdir` option takes (apps/maskrcnn.in.py:188-206: Mask%04d.plk = (boxes, masks, 81 class scores)) and a
groundtruth.txt in the benchmark's trajectory format (timestamp tx ty tz qx qy qz qw), so that the run the reference
evaluates -- run_exps.sh:31-33 then eval_tum.sh:30-35, ATE and RPE of poses-cam.txt -- can be repeated on it.

Scene (world = first camera frame, x right, y down, z forward, metres):
  * floor, back wall and side wall: three mutually non-parallel planes, none axis-aligned -> all six camera degrees
    of freedom are observable (the bench's synthetic stream, one wall + a floor + spheres, leaves one line free);
  * a desk (box), a cabinet (rotated box) and an ellipsoid, all static: background as far as the masks go --
    config/tum.cfg exports the `person` class only (FILTER_CLASSES = person);
  * one "person": a vertical capsule walking along x at ~0.45 m/s and swaying in z, whose instance mask goes into
    Mask%04d.plk every 30 frames with the class scores of `person` (index 1 of the COCO list).
Camera: fr3/walking_xyz style -- translations along all three axes (a Lissajous curve of ~15 cm amplitude, <= 9 mm per
frame) with small rotations about all three axes (<= 0.25 degrees per frame).
Depth: z in metres, multiplicative Gaussian noise sigma = 0.2 %, 1 % drop-out to 0, quantised to 1 / 5000 m by the PNG.

Test infrastructure.  Deterministic (seeded); the sequence is generated where the test runs, nothing is committed.
"""
from __future__ import annotations

import pickle
from pathlib import Path

import numpy as np

W, H = 640, 480
FX = FY = 525.0
CX, CY = W / 2 - 0.5, H / 2 - 0.5
FPS = 30.0
MASK_EVERY = 30
PERSON_CLASS = 1  # 'person' in the COCO class list of apps/maskrcnn.in.py:38


def _rot(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


TIME_SCALE = 1.0  # seconds of scene time per second of frames (a test at a coarser resolution slows the scene down)


def camera_pose(f: int):
    """camera -> world of frame f (float64); frame 0 is the identity."""
    s = TIME_SCALE * f / FPS
    t = np.array([0.16 * np.sin(2 * np.pi * 0.45 * s), 0.10 * (1 - np.cos(2 * np.pi * 0.35 * s)),
                  0.14 * np.sin(2 * np.pi * 0.30 * s)])
    R = (_rot([0, 1, 0], np.deg2rad(3.0) * np.sin(2 * np.pi * 0.40 * s)) @
         _rot([1, 0, 0], np.deg2rad(2.0) * np.sin(2 * np.pi * 0.33 * s)) @
         _rot([0, 0, 1], np.deg2rad(1.5) * np.sin(2 * np.pi * 0.25 * s)))
    return R, t


def person_position(f: int):
    """foot point of the capsule's axis on the floor plane's level (world)."""
    s = TIME_SCALE * f / FPS
    return np.array([-0.55 + 0.45 * s, 0.0, 2.05 + 0.12 * np.sin(2 * np.pi * 0.5 * s)])


# ---- primitives: each returns the ray parameter t (= camera z, rays have camera-frame z = 1) or +inf -------------------
def _plane(o, D, n, c):
    den = D @ n
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (c - o @ n) / den
    return np.where((np.abs(den) > 1e-12) & (t > 0), t, np.inf)


def _box(o, D, centre, Rb, half):
    ol = Rb.T @ (o - centre)
    Dl = D @ Rb  # rows: Rb^T d
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (-half - ol) / Dl
        t2 = (half - ol) / Dl
    tn = np.nanmax(np.minimum(t1, t2), axis=-1)
    tf = np.nanmin(np.maximum(t1, t2), axis=-1)
    return np.where((tn <= tf) & (tn > 0), tn, np.inf)


def _ellipsoid(o, D, centre, radii):
    ol = (o - centre) / radii
    Dl = D / radii
    a = np.sum(Dl * Dl, axis=-1)
    b = 2 * (Dl @ ol)
    c = ol @ ol - 1
    disc = b * b - 4 * a * c
    with np.errstate(invalid="ignore"):
        t = (-b - np.sqrt(disc)) / (2 * a)
    return np.where((disc > 0) & (t > 0), t, np.inf)


def _capsule(o, D, foot, radius, height):
    """vertical capsule: axis from foot - (0, radius, 0) up to foot - (0, height - radius, 0) (y points down)."""
    y0, y1 = foot[1] - radius, foot[1] - (height - radius)  # lower / upper sphere centres (y1 < y0)
    # cylinder around the y axis through (foot.x, *, foot.z)
    ox, oz = o[0] - foot[0], o[2] - foot[2]
    a = D[..., 0] ** 2 + D[..., 2] ** 2
    b = 2 * (ox * D[..., 0] + oz * D[..., 2])
    c = ox * ox + oz * oz - radius * radius
    disc = b * b - 4 * a * c
    with np.errstate(invalid="ignore", divide="ignore"):
        t = (-b - np.sqrt(disc)) / (2 * a)
    y = o[1] + t * D[..., 1]
    tc = np.where((disc > 0) & (t > 0) & (y <= y0) & (y >= y1), t, np.inf)
    ts0 = _ellipsoid(o, D, np.array([foot[0], y0, foot[2]]), np.full(3, radius))
    ts1 = _ellipsoid(o, D, np.array([foot[0], y1, foot[2]]), np.full(3, radius))
    return np.minimum(tc, np.minimum(ts0, ts1))


FLOOR_N = _rot([0, 0, 1], np.deg2rad(2.0)) @ _rot([1, 0, 0], np.deg2rad(-3.0)) @ np.array([0.0, 1.0, 0.0])
BACK_N = _rot([0, 1, 0], np.deg2rad(12.0)) @ _rot([1, 0, 0], np.deg2rad(4.0)) @ np.array([0.0, 0.0, 1.0])
SIDE_N = _rot([0, 1, 0], np.deg2rad(-8.0)) @ _rot([0, 0, 1], np.deg2rad(3.0)) @ np.array([-1.0, 0.0, 0.0])
FLOOR_C, BACK_C, SIDE_C = 1.05, 3.1, 1.45  # n . X = c


def intrinsics(width=W, height=H):
    """fx, fy, cx, cy of Params' default camera scaled to the image width (data.h:84-90)."""
    s = width / 640.0
    return FX * s, FY * s, width / 2 - 0.5, height / 2 - 0.5


def render(f: int, rng: np.random.Generator | None = None, noise=0.002, dropout=0.01, size=(W, H)):
    """depth (H, W) float32 metres, instance ids (H, W) uint8: 0 static scene, 1 the person."""
    R, o = camera_pose(f)
    w, h = size
    fx, fy, cx, cy = intrinsics(w, h)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)
    D = d @ R.T
    layers = [
        _plane(o, D, FLOOR_N, FLOOR_C), _plane(o, D, BACK_N, BACK_C), _plane(o, D, SIDE_N, SIDE_C),
        _box(o, D, np.array([0.75, 0.70, 2.55]), _rot([0, 1, 0], np.deg2rad(20.0)), np.array([0.45, 0.35, 0.30])),
        _box(o, D, np.array([-0.95, 0.35, 2.75]), _rot([0, 1, 0], np.deg2rad(-25.0)) @ _rot([0, 0, 1], np.deg2rad(5.0)),
             np.array([0.25, 0.70, 0.22])),
        _ellipsoid(o, D, np.array([0.15, 0.78, 1.75]), np.array([0.22, 0.26, 0.18])),
    ]
    static = np.minimum.reduce(layers)
    person = _capsule(o, D, np.array([person_position(f)[0], FLOOR_C, person_position(f)[2]]), 0.19, 1.55)
    depth = np.minimum(static, person)
    ids = (person < static).astype(np.uint8)
    depth = np.where(np.isfinite(depth) & (depth < 6.0), depth, 0.0)
    if rng is not None:
        depth = depth * (1.0 + noise * rng.standard_normal(depth.shape))
        depth = np.where(rng.random(depth.shape) < dropout, 0.0, depth)
    ids[depth == 0] = 0
    return depth.astype(np.float32), ids


def quaternion(R):
    """(qx, qy, qz, qw) of a rotation matrix."""
    q = np.empty(4)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q[:] = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q


def stage(root, frames=60, seed=0x7A5C, size=(W, H)):
    """Writes <root>/seq/{associations.txt, groundtruth.txt, depth/NNNN.png} and <root>/masks/MaskNNNN.plk.
    Returns dict(seq=..., masks=..., truth=[(R, t)], depth=[(H, W) f32 as the PNG holds it], ids=[...])."""
    from emfusion_amd import readers
    root = Path(root)
    seq, masks = root / "seq", root / "masks"
    (seq / "depth").mkdir(parents=True, exist_ok=True)
    masks.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    lines, gt, truth, depths, idmaps = [], [], [], [], []
    for f in range(frames):
        depth, ids = render(f, rng, size=size)
        q16 = np.round(depth * 5000.0).astype(np.uint16)
        readers.write_png_gray16(seq / "depth" / f"{f:04d}.png", q16, filters=np.arange(size[1]) % 5)
        depths.append(q16.astype(np.float32) / np.float32(5000.0))
        idmaps.append(ids)
        ts = f / FPS
        lines.append(f"{ts:.6f} rgb/{f:04d}.png {ts:.6f} depth/{f:04d}.png")
        R, t = camera_pose(f)
        truth.append((R, t))
        q = quaternion(R)
        gt.append(f"{ts:.6f} {t[0]:.6f} {t[1]:.6f} {t[2]:.6f} {q[0]:.6f} {q[1]:.6f} {q[2]:.6f} {q[3]:.6f}")
        if f % MASK_EVERY == 0:
            m = ids == 1
            ys, xs = np.nonzero(m)
            box = [int(ys.min()), int(xs.min()), int(ys.max()) + 1, int(xs.max()) + 1] if m.any() else [0, 0, 1, 1]
            scores = np.full(81, 0.001)
            scores[PERSON_CLASS] = 0.92
            with open(masks / f"Mask{f:04d}.plk", "wb") as fh:
                pickle.dump(([box], [m], [scores.tolist()]), fh, protocol=2)
    (seq / "associations.txt").write_text("\n".join(lines) + "\n")
    (seq / "groundtruth.txt").write_text("# timestamp tx ty tz qx qy qz qw\n" + "\n".join(gt) + "\n")
    return dict(seq=str(seq) + "/", masks=str(masks), truth=truth, depth=depths, ids=idmaps)


# ---- the benchmark's two error measures (rgbd_benchmark_tools evaluate_ate.py / evaluate_rpe.py, restated) --------------
def ate_rmse(est_t, true_t):
    """Absolute trajectory error: RMSE of the translational differences after the rigid alignment (Horn's closed form,
    no scale) of the estimated onto the true positions."""
    P, Q = np.asarray(est_t, np.float64), np.asarray(true_t, np.float64)
    pc, qc = P.mean(0), Q.mean(0)
    Wm = (Q - qc).T @ (P - pc)
    U, _, Vt = np.linalg.svd(Wm)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    Rr = U @ S @ Vt
    err = (Q - qc) - (P - pc) @ Rr.T
    return float(np.sqrt((err ** 2).sum(1).mean()))


def rpe(est, true, delta):
    """Relative pose error over pairs `delta` frames apart (evaluate_rpe.py --fixed_delta): RMSE of the translational part
    [m] and of the rotation angle [rad] of (Q_i^-1 Q_j)^-1 (P_i^-1 P_j)."""
    def T(R, t):
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = R, t
        return M
    te, re = [], []
    for i in range(len(est) - delta):
        Pi, Pj, Qi, Qj = T(*est[i]), T(*est[i + delta]), T(*true[i]), T(*true[i + delta])
        E = np.linalg.inv(np.linalg.inv(Qi) @ Qj) @ (np.linalg.inv(Pi) @ Pj)
        te.append(np.linalg.norm(E[:3, 3]))
        re.append(np.arccos(min(1.0, max(-1.0, (np.trace(E[:3, :3]) - 1) / 2))))
    return float(np.sqrt(np.mean(np.square(te)))), float(np.sqrt(np.mean(np.square(re))))
