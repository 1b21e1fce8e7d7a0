"""Frame-level oracle of the tracker: TSDF::prepareTracking ... computePoseUpdate / syncTrack
(reference src/core/TSDF.cpp:170-344, 375-395) restated over the CPU oracle kernels, host
arithmetic in float32.  Test infrastructure only.

Third-party pieces the reference calls and this file restates (all absent from the reference tree,
versions unpinned, hence "parity unpinned"): Eigen's Householder QR with the sign fix of
TSDF.cpp:176-183 (= Gram-Schmidt with positive diagonal), Sophus::SE3f::exp / log, cv::solve
(DECOMP_LU, float), cv::cuda::normalize(NORM_INF), cv::cuda::reduce / sum.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def orthonormalise(R):
    """Q of the QR decomposition with a positive diagonal of R (TSDF.cpp:176-183)."""
    R = np.asarray(R, np.float64).reshape(3, 3)
    q, r = np.linalg.qr(R)
    q = q * np.sign(np.diag(r))[None, :]
    return q.astype(f32)


def _hat(o):
    return np.array([[0, -o[2], o[1]], [o[2], 0, -o[0]], [-o[1], o[0], 0]], f32)


def se3_exp(x):
    """exp of the twist (upsilon, omega), float32 (Sophus::SE3f::exp)."""
    x = np.asarray(x, f32)
    u, o = x[:3], x[3:]
    th2 = f32(o @ o)
    th = f32(np.sqrt(th2))
    if th < 1e-4:
        A, B, C = f32(1) - th2 / f32(6), f32(0.5) - th2 / f32(24), f32(1) / f32(6) - th2 / f32(120)
    else:
        A, B, C = f32(np.sin(th)) / th, (f32(1) - f32(np.cos(th))) / th2, (th - f32(np.sin(th))) / (th2 * th)
    O = _hat(o)
    O2 = (O @ O).astype(f32)
    R = (np.eye(3, dtype=f32) + A * O + B * O2).astype(f32)
    V = (np.eye(3, dtype=f32) + B * O + C * O2).astype(f32)
    return R, (V @ u).astype(f32)


def se3_log_norm(R, t):
    R = np.asarray(R, np.float64).reshape(3, 3)
    c = min(1.0, max(-1.0, (np.trace(R) - 1.0) * 0.5))
    th = np.arccos(c)
    a = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    o = a * (0.5 * (1 + th * th / 6) if th < 1e-4 else th / (2 * np.sin(th)))
    th2 = float(o @ o)
    if th2 < 1e-8:
        D = 1.0 / 12.0
    else:
        hf = 0.5 * np.sqrt(th2)
        D = (1 - hf * np.cos(hf) / np.sin(hf)) / th2
    O = _hat(o).astype(np.float64)
    u = np.asarray(t, np.float64) - 0.5 * (O @ t) + D * (O @ (O @ t))
    return float(np.sqrt(u @ u + th2))


class OracleTracker:
    """One volume's LM state (the members TSDF.h:295-327 keeps for tracking)."""

    def __init__(self, orc, tsdf, weights, voxel, huber=0.2, max_weight=64.0, tau=1e3, eps1=1e-8,
                 eps2=1e-8, nu_init=2.0):
        self.o, self.tsdf, self.wts, self.vox = orc, tsdf, weights, f32(voxel)
        self.p = dict(huber=f32(huber), maxw=f32(max_weight), tau=f32(tau), eps1=eps1, eps2=eps2,
                      nu=f32(nu_init))

    def prepare(self, R_CO, t_CO):  # TSDF.cpp:170-192 (the caller forms pose.inv() * cam_pose)
        self.R, self.t = orthonormalise(R_CO), np.asarray(t_CO, f32).copy()
        self.nu = self.p["nu"]
        self.converged, self.first, self.eval_grad = False, True, True
        self.mu = f32(0)
        self.A, self.b = np.zeros((6, 6), f32), np.zeros(6, f32)
        self.iterations = self.accepted = 0
        self.history = []

    def _vals(self, points, R, t):
        return self.o.get_volume_vals(self.tsdf, points, R.reshape(-1), t, self.vox).reshape(-1)

    def iterate(self, points, assoc):
        """One pass of the loop body of EMFusion::performTracking (EMFusion.cpp:674-684)."""
        if self.converged:
            return
        o, p = self.o, self.p
        if self.eval_grad:
            self.g6 = o.compute_pose_gradients(self.tsdf, None, points, self.R.reshape(-1), self.t, self.vox)
        self.vals = self._vals(points, self.R, self.t)  # computeTSDFVals: every iteration
        if self.eval_grad:
            raw = o.get_volume_vals(self.wts, points, self.R.reshape(-1), self.t, self.vox).reshape(-1)
            _, self.w = o.tracking_weights(self.vals, raw, np.asarray(assoc, f32).reshape(-1),
                                           p["huber"], p["maxw"])
            self.A, self.b = o.reduce_ab(self.g6, self.vals, self.w)  # reduceHessians
            self.converged = bool(np.abs(self.b).max() < p["eps1"])
            if self.converged:
                return
        # ---- computePoseUpdate (TSDF.cpp:281-337) ----
        if self.first:
            self.mu = f32(p["tau"] * f32(np.diag(self.A).max()))
            self.first = False
        M = (self.A + self.mu * np.eye(6, dtype=f32)).astype(f32)
        x = np.linalg.solve(M.astype(np.float64), self.b.astype(np.float64)).astype(f32)
        if float(np.linalg.norm(x)) < p["eps2"] * (se3_log_norm(self.R, self.t) + p["eps2"]):
            self.converged = True
            return
        err = f32(o.tracking_error(self.vals, self.w))
        Ri, ti = se3_exp(-x)
        Rn = (Ri @ self.R).astype(f32)
        tn = (Ri @ self.t + ti).astype(f32)
        vals_new = self._vals(points, Rn, tn)
        err_new = f32(o.tracking_error(vals_new, self.w))
        gain = f32(0.5) * f32(np.dot(-x, self.mu * -x - self.b))
        rho = (err - err_new) / gain if gain != 0 else f32(np.nan)
        self.iterations += 1
        self.history.append(dict(x=x.copy(), err=float(err), err_new=float(err_new), rho=float(rho),
                                 mu=float(self.mu), A=self.A.copy(), b=self.b.copy(), Rtrial=Rn, ttrial=tn))
        if rho > 0:
            self.R, self.t = Rn, tn
            c = f32(2) * rho - f32(1)
            self.mu = f32(self.mu * max(f32(1.0 / 3.0), f32(1) - c * c * c))
            self.nu = p["nu"]
            self.eval_grad = True
            self.accepted += 1
        else:
            self.mu = f32(self.mu * self.nu)
            self.nu = f32(self.nu * p["nu"])
            self.eval_grad = False
