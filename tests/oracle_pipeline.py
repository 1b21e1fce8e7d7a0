"""Frame-level oracle: the per-frame schedule of the reference's EMFusion::processFrame
(src/core/EMFusion.cpp:70-129, 635-670, 726-795, 865-906) restated over the CPU oracle kernels.

Test infrastructure only.  Tracking / Mask R-CNN are replaced by supplied poses and masks exactly
as in emf::EMFusion (emfusion_amd/csrc/core/EMFusion.cpp), so the two can be compared frame by
frame.  Pose algebra is done in float32 in the same operation order as emf::Affine3f.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


class Affine32:
    """float32 rigid transform with emf::Affine3f's arithmetic (types.hpp)."""

    def __init__(self, R=None, t=None):
        self.R = np.eye(3, dtype=f32) if R is None else np.asarray(R, f32).reshape(3, 3).copy()
        self.t = np.zeros(3, f32) if t is None else np.asarray(t, f32).reshape(3).copy()

    @staticmethod
    def _mv(M, v):
        return np.array([f32(f32(f32(M[i, 0] * v[0]) + f32(M[i, 1] * v[1])) + f32(M[i, 2] * v[2]))
                         for i in range(3)], f32)

    @staticmethod
    def _mm(A, B):
        out = np.empty((3, 3), f32)
        for i in range(3):
            for j in range(3):
                out[i, j] = f32(f32(f32(A[i, 0] * B[0, j]) + f32(A[i, 1] * B[1, j])) +
                                f32(A[i, 2] * B[2, j]))
        return out

    def inv(self):
        Rt = self.R.T.copy()
        return Affine32(Rt, -self._mv(Rt, self.t))

    def __mul__(self, o):
        return Affine32(self._mm(self.R, o.R), self._mv(self.R, o.t) + self.t)


class OraclePipeline:
    def __init__(self, orc, width, height, K, bg_res, bg_voxel, volume_pose_t, obj_res,
                 rel_trunc=10.0, max_weight=64.0, sigma=0.02, alpha=0.8, prior=1.0,
                 visibility_thresh=1600, boundary=20):
        self.o = orc
        self.w, self.h, self.K = width, height, np.asarray(K, f32).reshape(3, 3)
        self.p = dict(rel_trunc=f32(rel_trunc), max_weight=max_weight, sigma=sigma, alpha=alpha,
                      prior=prior, vis=visibility_thresh, boundary=boundary)
        self.bg = self._new_volume(0, bg_res, f32(bg_voxel), Affine32(t=volume_pose_t), False)
        self.obj_res = obj_res
        self.objects = []  # creation order
        self.pose = Affine32()
        self.frame = 0
        self.vis = set()
        self.bg_assoc = np.ones((height, width), f32)
        self.diff = np.zeros((height, width), f32)
        self.seg = np.zeros((height, width), np.uint8)
        self.ray = np.zeros((height, width), f32)
        self.bg_ray = np.zeros((height, width), f32)
        self.norm = np.zeros((height, width), f32)
        self.points = None
        self.march_samples = 0

    def _new_volume(self, vid, res, vox, pose, is_obj, trunc=None):
        n = (res, res, res) if np.isscalar(res) else tuple(res)
        trunc = f32(self.p["rel_trunc"] * f32(vox)) if trunc is None else f32(trunc)
        v = dict(id=vid, n=n, vox=f32(vox), trunc=trunc, pose=pose,
                 tsdf=np.zeros((n[2], n[1], n[0]), f32), wts=np.zeros((n[2], n[1], n[0]), f32))
        if is_obj:
            v.update(fgbg=np.zeros((n[2], n[1], n[0], 2), f32),
                     probs=np.zeros((n[2], n[1], n[0]), f32),
                     vmask=np.zeros((n[2], n[1], n[0]), np.uint8),
                     assoc=np.ones((self.h, self.w), f32),
                     seg=np.zeros((self.h, self.w), np.uint8),
                     ray=np.zeros((self.h, self.w), f32))
        return v

    def add_object(self, center, vol_size):
        vid = len(self.objects) + 1
        res = self.obj_res
        vox = f32(f32(vol_size) / f32(res))  # EMFusion::addObject: volSize / float(res[0])
        trunc = f32(f32(self.p["rel_trunc"] * f32(vol_size)) / f32(res))  # (rel * volSize) / res, EMFusion.cpp:545-547
        self.objects.append(self._new_volume(vid, res, vox, Affine32(t=center), True, trunc))
        self.vis.add(vid)
        return vid

    # ---- stages -------------------------------------------------------------------------------
    def _estep(self):
        maps = []
        for v in [self.bg] + sorted(self.objects, key=lambda v: v["id"]):
            co = v["pose"].inv() * self.pose
            m = self.o.compute_association(v["tsdf"], v.get("probs"), self.points, co.R, co.t,
                                           v["vox"], v["trunc"], self.p["sigma"],
                                           self.p["alpha"], self.p["prior"])
            maps.append(m)
        self.norm = self.o.normalize_association(maps)
        self.bg_assoc = maps[0]
        for v, m in zip(sorted(self.objects, key=lambda v: v["id"]), maps[1:]):
            v["assoc"] = m

    def _raycast(self):
        self.vis = set()
        co = self.bg["pose"].inv() * self.pose
        bg = self.o.raycast_tsdf(self.bg["tsdf"], None, self.bg["wts"], None, self.w, self.h,
                                 co.R, co.t, self.K, self.bg["vox"], self.bg["trunc"],
                                 count_steps=True)
        self.march_samples += int(bg[4].sum())
        outs = []
        for v in self.objects:
            co = v["pose"].inv() * self.pose
            r = self.o.raycast_tsdf(v["tsdf"], None, v["wts"], v["vmask"], self.w, self.h, co.R,
                                    co.t, self.K, v["vox"], v["trunc"], count_steps=True)
            self.march_samples += int(r[4].sum())
            v["ray"], v["seg"] = r[0], r[3]
            outs.append(r)
        ids = [v["id"] for v in self.objects]
        comp = self.o.composite_raycast(ids, [r[0] for r in outs], [r[1] for r in outs],
                                        [r[2] for r in outs], [r[3] for r in outs], bg[0], bg[1],
                                        bg[2], bg[3], self.diff, self.p["boundary"])
        self.ray, self.vert, self.nrm, self.seg, self.no_obj, counts = comp
        self.bg_ray = bg[0]
        for i, c in zip(ids, counts):
            if c > self.p["vis"]:
                self.vis.add(i)

    def _integrate(self, depth):
        oc = self.pose.inv() * self.bg["pose"]
        self.o.update_tsdf(depth, self.bg_assoc, self.bg["tsdf"], self.bg["wts"], oc.R, oc.t,
                           self.K, self.bg["vox"], self.bg["trunc"], self.p["max_weight"])
        for v in self.objects:
            if v["id"] not in self.vis:
                continue
            oc = self.pose.inv() * v["pose"]
            self.o.update_tsdf(depth, v["assoc"], v["tsdf"], v["wts"], oc.R, oc.t, self.K,
                               v["vox"], v["trunc"], self.p["max_weight"])

    def _integrate_masks(self, masks):
        for v in self.objects:
            if v["id"] not in masks:
                continue
            occ = self.o.occluded_mask(v["seg"], self.seg, v["id"])
            oc = self.pose.inv() * v["pose"]
            self.o.update_fgbg_probs(masks[v["id"]], occ, v["tsdf"], v["wts"], v["fgbg"], oc.R,
                                     oc.t, self.K, v["vox"])
            v["probs"], v["vmask"] = self.o.compute_fg_probs(v["fgbg"])

    # ---- the schedule -------------------------------------------------------------------------
    def _track(self, vols, assoc_of, iters):
        """Lock-step LM of `vols` (EMFusion.cpp:673-684 / 692-720); returns their rel_pose_CO."""
        from tests.oracle_tracking import OracleTracker
        trackers = []
        for v in vols:
            t = OracleTracker(self.o, v["tsdf"], v["wts"], v["vox"], max_weight=self.p["max_weight"])
            co = v["pose"].inv() * self.pose
            t.prepare(co.R, co.t)
            trackers.append(t)
        for _ in range(iters):
            for v, t in zip(vols, trackers):
                t.iterate(self.points, assoc_of(v))
        return trackers

    def process_frame(self, depth, cam_pose: Affine32, obj_poses=None, masks=None,
                      run_masks=False, track_camera=False, track_objects=False, track_iters=100):
        obj_poses = obj_poses or {}
        self.points = self.o.compute_points(depth, self.K)

        def apply_obj():
            for v in self.objects:
                if v["id"] in obj_poses:
                    v["pose"] = obj_poses[v["id"]]

        if self.frame > 0:
            self._estep()
            if track_camera:
                t = self._track([self.bg], lambda v: self.bg_assoc, track_iters)[0]
                self.pose = self.bg["pose"] * Affine32(t.R, t.t)  # TSDF::syncTrack
                self.track = {0: t}
            else:
                self.pose = cam_pose
            self._estep()
            if track_objects:
                objs = sorted(self.objects, key=lambda v: v["id"])
                for v, t in zip(objs, self._track(objs, lambda v: v["assoc"], track_iters)):
                    v["pose"] = self.pose * Affine32(t.R, t.t).inv()  # ObjTSDF::syncTrack
                    self.track[v["id"]] = t
            else:
                apply_obj()
            self._estep()
            self._raycast()
        else:
            self.pose = cam_pose
            apply_obj()
        self._integrate(depth)
        if run_masks and masks:
            self._integrate_masks(masks)
        self.frame += 1
