"""Small deterministic scenes for parity tests: analytic depth renderer + pose helpers (numpy).

World frame = first camera frame (x right, y down, z forward), like the reference's default
configuration where the background volume is centred at (0, 0, volSize/2) in front of the camera
(reference include/EMFusion/core/data.h:96-103).
"""
from __future__ import annotations

import numpy as np


def intrinsics(w: int, h: int) -> np.ndarray:
    """Reference defaults scaled to the image size (data.h:84-90)."""
    f = 525.0 * w / 640.0
    return np.array([[f, 0, w / 2 - 0.5], [0, f, h / 2 - 0.5], [0, 0, 1]], np.float32)


def rot(axis, angle_deg: float) -> np.ndarray:
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    a = np.deg2rad(angle_deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


class Pose:
    """Rigid transform x_out = R x_in + t (double precision on the host, cast at the ABI)."""

    def __init__(self, R=None, t=None):
        self.R = np.eye(3) if R is None else np.asarray(R, np.float64)
        self.t = np.zeros(3) if t is None else np.asarray(t, np.float64)

    def inv(self) -> "Pose":
        return Pose(self.R.T, -self.R.T @ self.t)

    def __mul__(self, o: "Pose") -> "Pose":
        return Pose(self.R @ o.R, self.R @ o.t + self.t)

    @property
    def R32(self):
        return self.R.astype(np.float32).reshape(-1)

    @property
    def t32(self):
        return self.t.astype(np.float32)


def rel_OC(cam_pose: Pose, vol_pose: Pose) -> Pose:
    """volume -> camera, reference TSDF.cpp:112"""
    return cam_pose.inv() * vol_pose


def rel_CO(cam_pose: Pose, vol_pose: Pose) -> Pose:
    """camera -> volume, reference TSDF.cpp:141,162"""
    return vol_pose.inv() * cam_pose


def render_depth(w, h, K, cam_pose: Pose, spheres=(), wall=(0.15, -0.1, 2.2), floor_y=1.0,
                 noise=0.0, dropout=0.0, seed=0):
    """z-depth image (metres, float32) of a tilted wall z = c + a x + b y, a floor y = floor_y and
    spheres [(centre(3), radius)], all in world coordinates.  Also returns the per-pixel id of the
    nearest sphere (0 = none) for building object masks."""
    K = np.asarray(K, np.float64)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    d_cam = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], -1)
    d_w = d_cam @ cam_pose.R.T  # ray directions in world, un-normalised (z_cam = 1 per unit s)
    o = cam_pose.t
    best = np.full((h, w), np.inf)
    ids = np.zeros((h, w), np.int32)
    # wall: z - a x - b y - c = 0
    a, b, c = wall
    nrm = np.array([-a, -b, 1.0])
    denom = d_w @ nrm
    s = (c - o @ nrm) / np.where(np.abs(denom) < 1e-12, np.nan, denom)
    ok = np.isfinite(s) & (s > 0)
    best = np.where(ok & (s < best), s, best)
    # floor: y = floor_y
    if floor_y is not None:
        s = (floor_y - o[1]) / np.where(np.abs(d_w[..., 1]) < 1e-12, np.nan, d_w[..., 1])
        ok = np.isfinite(s) & (s > 0)
        best = np.where(ok & (s < best), s, best)
    for k, (cen, r) in enumerate(spheres):
        cen = np.asarray(cen, np.float64)
        oc = o - cen
        A = np.sum(d_w * d_w, -1)
        B = 2 * (d_w @ oc)
        Cc = oc @ oc - r * r
        disc = B * B - 4 * A * Cc
        s = (-B - np.sqrt(np.where(disc >= 0, disc, np.nan))) / (2 * A)
        ok = np.isfinite(s) & (s > 0) & (s < best)
        best = np.where(ok, s, best)
        ids = np.where(ok, k + 1, ids)
    depth = np.where(np.isfinite(best), best, 0.0)  # camera-frame z equals the ray parameter s
    rng = np.random.default_rng(seed)
    if noise > 0:
        depth = depth * (1.0 + noise * rng.standard_normal(depth.shape))
    if dropout > 0:
        depth = np.where(rng.random(depth.shape) < dropout, 0.0, depth)
    return depth.astype(np.float32), ids


def camera_path(frame: int) -> Pose:
    """Small deterministic camera motion: 5 cm circle, <= 0.5 deg per frame."""
    ang = 2 * np.pi * frame / 40.0
    t = np.array([0.05 * np.cos(ang) - 0.05, 0.05 * np.sin(ang), 0.0])
    R = rot([0.2, 1.0, 0.1], 0.4 * frame)
    return Pose(R, t)
