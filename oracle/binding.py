"""numpy binding of the CPU oracle (oracle/emf_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by emfusion_amd/.  PARITY UNPINNED (see emf_oracle.h).

Arrays follow the reference layout: images (H, W[, C]) float32/uint8 C-contiguous, volumes
(Nz, Ny, Nx[, C]).  Functions that the reference runs in place modify their array arguments.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_libs: dict[str, C.CDLL] = {}

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(target: str = "all") -> None:
    subprocess.run(["make", "-s", "-C", str(HERE), target], check=True)


_default = False  # what lib(False) means: the portable -O2 build, or (use_native) the -O3 -march=native one


def use_native(on: bool = True) -> bool:
    """bench.py's cpu_baseline leg: build libemf_oracle_native.so (-O3 -march=native, still
    -ffp-contract=off: same bits) ON THIS HOST and make it what every binding call uses.  Returns
    whether the native build is in use (False: the compiler refused, the portable build stays)."""
    global _default
    _default = False
    if on:
        try:
            subprocess.run(["make", "-s", "-B", "-C", str(HERE), "native"], check=True)  # always for THIS host
            C.CDLL(str(HERE / "libemf_oracle_native.so"))
            _default = "native"
        except (subprocess.CalledProcessError, OSError):
            _default = False
    return _default == "native"


def lib(fma=False) -> C.CDLL:
    """The oracle library; ``fma=True`` loads the build with a*b+c contraction enabled."""
    variant = fma or _default
    name = {False: "libemf_oracle.so", True: "libemf_oracle_fma.so",
            "native": "libemf_oracle_native.so"}[variant]
    if name not in _libs:
        path = HERE / name
        if not path.exists():
            if variant == "native":
                raise RuntimeError("call use_native() first: the native oracle is built per host")
            build()
        _libs[name] = C.CDLL(str(path))
        _libs[name].orc_set_threads.restype = C.c_int
    return _libs[name]


def host_threads() -> int:
    """Threads this process may really use: its CPU affinity mask, capped by the cgroup's CPU quota
    (cpu.max of cgroup v2, cfs quota of v1) -- omp_get_num_procs() sees only the former."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, math.ceil(q / p)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _c(a, dtype=np.float32):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _farr(v, n):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(-1))
    assert a.size == n
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _res(vol):
    nz, ny, nx = vol.shape[:3]
    return (C.c_int * 3)(nx, ny, nz)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_threads(n: int, fma: bool = False) -> int:
    return lib(fma).orc_set_threads(int(n))


def compute_points(depth, K, fma=False):
    depth = _c(depth)
    h, w = depth.shape
    pts = np.zeros((h, w, 3), np.float32)
    lib(fma).orc_computePoints(_p(depth), _p(pts), w, h, _farr(K, 9))
    return pts


def update_tsdf(depth, assoc, tsdf, weights, R_OC, t_OC, K, voxel_size, truncdist, max_weight,
                fma=False):
    depth, assoc = _c(depth), _c(assoc)
    assert tsdf.flags.c_contiguous and weights.flags.c_contiguous
    h, w = depth.shape
    lib(fma).orc_updateTSDF(_p(depth), _p(assoc), w, h, _p(tsdf), _p(weights), _farr(R_OC, 9),
                            _farr(t_OC, 3), _farr(K, 9), _res(tsdf), C.c_float(voxel_size),
                            C.c_float(truncdist), C.c_float(max_weight))


def compute_tsdf_grads(tsdf, fma=False):
    tsdf = _c(tsdf)
    grads = np.empty(tsdf.shape + (3,), np.float32)
    lib(fma).orc_computeTSDFGrads(_p(tsdf), _p(grads), _res(tsdf))
    return grads


def raycast_tsdf(tsdf, grads, weights, fg_mask, w, h, R_CO, t_CO, K, voxel_size, truncdist,
                 raylengths=None, count_steps=False, fma=False):
    tsdf, weights = _c(tsdf), _c(weights)
    grads = None if grads is None else _c(grads)
    fg_mask = None if fg_mask is None else _c(fg_mask, np.uint8)
    ray = np.zeros((h, w), np.float32) if raylengths is None else _c(raylengths).copy()
    vert = np.zeros((h, w, 3), np.float32)
    nrm = np.zeros((h, w, 3), np.float32)
    mask = np.zeros((h, w), np.uint8)
    steps = np.zeros((h, w), np.uint32) if count_steps else None
    lib(fma).orc_raycastTSDF(_p(tsdf), _p(grads), _p(weights), _p(fg_mask), _p(ray), _p(vert),
                             _p(nrm), _p(mask), w, h, _farr(R_CO, 9), _farr(t_CO, 3), _farr(K, 9),
                             _res(tsdf), C.c_float(voxel_size), C.c_float(truncdist), _p(steps))
    out = (ray, vert, nrm, mask)
    return out + (steps,) if count_steps else out


def get_volume_vals(vol, points, R_CO, t_CO, voxel_size, fma=False):
    vol, points = _c(vol), _c(points)
    ch = 1 if vol.ndim == 3 else vol.shape[3]
    h, w = points.shape[:2]
    vals = np.empty((h, w) if ch == 1 else (h, w, ch), np.float32)
    lib(fma).orc_getVolumeVals(_p(vol), ch, _p(points), w, h, _farr(R_CO, 9), _farr(t_CO, 3),
                               _res(vol), C.c_float(voxel_size), _p(vals))
    return vals


def update_fgbg_probs(mask, occluded, tsdf, weights, fgbg, R_OC, t_OC, K, voxel_size, fma=False):
    mask, occluded = _c(mask, np.uint8), _c(occluded, np.uint8)
    tsdf, weights = _c(tsdf), _c(weights)
    assert fgbg.flags.c_contiguous and fgbg.dtype == np.float32
    h, w = mask.shape
    lib(fma).orc_updateFgBgProbs(_p(mask), _p(occluded), w, h, _p(tsdf), _p(weights), _p(fgbg),
                                 _farr(R_OC, 9), _farr(t_OC, 3), _farr(K, 9), _res(tsdf),
                                 C.c_float(voxel_size))


def compute_fg_probs(fgbg, fma=False):
    fgbg = _c(fgbg)
    probs = np.empty(fgbg.shape[:3], np.float32)
    vmask = np.empty(fgbg.shape[:3], np.uint8)
    lib(fma).orc_computeFgProbs(_p(fgbg), _p(probs), _p(vmask), _res(fgbg))
    return probs, vmask


def mask_raycast_weights(weights, fg_vol_mask, fma=False):
    weights, fg_vol_mask = _c(weights), _c(fg_vol_mask, np.uint8)
    out = np.empty_like(weights)
    lib(fma).orc_maskRaycastWeights(_p(weights), _p(fg_vol_mask), _p(out), _res(weights))
    return out


def compute_association(tsdf, fg_probs, points, R_CO, t_CO, voxel_size, truncdist, sigma, alpha,
                        uni_prior, fma=False):
    tsdf, points = _c(tsdf), _c(points)
    fg_probs = None if fg_probs is None else _c(fg_probs)
    h, w = points.shape[:2]
    out = np.empty((h, w), np.float32)
    lib(fma).orc_computeAssociation(_p(tsdf), _p(fg_probs), _p(points), w, h, _farr(R_CO, 9),
                                    _farr(t_CO, 3), _res(tsdf), C.c_float(voxel_size),
                                    C.c_float(truncdist), C.c_float(sigma), C.c_float(alpha),
                                    C.c_float(uni_prior), _p(out))
    return out


def normalize_association(maps, fma=False):
    """In place on the list of (H, W) float32 arrays; returns the normaliser."""
    for m in maps:
        assert m.flags.c_contiguous and m.dtype == np.float32
    h, w = maps[0].shape
    ptrs = (C.c_void_p * len(maps))(*[m.ctypes.data for m in maps])
    norm = np.empty((h, w), np.float32)
    lib(fma).orc_normalizeAssociation(ptrs, len(maps), w, h, _p(norm))
    return norm


def composite_raycast(ids, obj_ray, obj_vert, obj_norm, obj_seg, bg_ray, bg_vert, bg_norm, bg_mask,
                      diff, boundary, fma=False):
    """Returns (ray, vert, norm, seg, no_obj, vis_counts); ``diff`` is updated in place."""
    n = len(ids)
    h, w = bg_ray.shape

    def tab(arrs, dt):
        keep = [_c(a, dt) for a in arrs]
        return keep, (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in keep])

    k1, pr = tab(obj_ray, np.float32)
    k2, pv = tab(obj_vert, np.float32)
    k3, pn = tab(obj_norm, np.float32)
    k4, ps = tab(obj_seg, np.uint8)
    bg_ray, bg_vert, bg_norm = _c(bg_ray), _c(bg_vert), _c(bg_norm)
    bg_mask = _c(bg_mask, np.uint8)
    assert diff.flags.c_contiguous and diff.dtype == np.float32
    ray = np.empty((h, w), np.float32)
    vert = np.empty((h, w, 3), np.float32)
    nrm = np.empty((h, w, 3), np.float32)
    seg = np.empty((h, w), np.uint8)
    no_obj = np.empty((h, w), np.uint8)
    vis = np.zeros(max(n, 1), np.int32)
    ids_arr = (C.c_int * max(n, 1))(*[int(i) for i in ids])
    lib(fma).orc_compositeRaycast(n, ids_arr, pr, pv, pn, ps, _p(bg_ray), _p(bg_vert), _p(bg_norm),
                                  _p(bg_mask), _p(ray), _p(vert), _p(nrm), _p(seg), _p(diff),
                                  _p(no_obj), w, h, int(boundary), _p(vis))
    return ray, vert, nrm, seg, no_obj, vis[:n]


def occluded_mask(obj_seg, seg, obj_id, fma=False):
    obj_seg, seg = _c(obj_seg, np.uint8), _c(seg, np.uint8)
    h, w = seg.shape
    occ = np.empty((h, w), np.uint8)
    lib(fma).orc_occludedMask(_p(obj_seg), _p(seg), int(obj_id), _p(occ), w, h)
    return occ


# ---- f-1: tracking ------------------------------------------------------------------------------

def compute_pose_gradients(tsdf, grads_vol, points, R_CO, t_CO, voxel_size, fma=False):
    tsdf, points = _c(tsdf), _c(points)
    h, w = points.shape[:2]
    out = np.empty((h * w, 6), np.float32)
    gv = _c(grads_vol) if grads_vol is not None else None
    lib(fma).orc_computePoseGradients(_p(tsdf), _p(gv) if gv is not None else None, _p(points), w, h,
                                      _farr(R_CO, 9), _farr(t_CO, 3), _res(tsdf),
                                      C.c_float(voxel_size), _p(out))
    return out


def tracking_weights(tsdf_vals, int_weights_raw, assoc, huber, max_weight, fma=False):
    tv, iw, a = _c(tsdf_vals), _c(int_weights_raw), _c(assoc)
    tw, out = np.empty_like(tv), np.empty_like(tv)
    lib(fma).orc_trackingWeights(_p(tv), _p(iw), _p(a), tv.size, C.c_float(huber),
                                 C.c_float(max_weight), _p(tw), _p(out))
    return tw, out


def reduce_ab(grads6, tsdf_vals, int_weights, fma=False):
    g, tv, iw = _c(grads6), _c(tsdf_vals), _c(int_weights)
    A, b = np.empty(36, np.float32), np.empty(6, np.float32)
    lib(fma).orc_reduceAb(_p(g), _p(tv), _p(iw), tv.size, _p(A), _p(b))
    return A.reshape(6, 6), b


def tracking_error(tsdf_vals, int_weights, fma=False):
    tv, iw = _c(tsdf_vals), _c(int_weights)
    f = lib(fma).orc_trackingError
    f.restype = C.c_double
    return float(f(_p(tv), _p(iw), tv.size))


# ---- f-2: depth pre-processing ------------------------------------------------------------------

def preprocess_depth(raw, ksz=7, sigma_depth=0.04, sigma_spatial=4.5, fma=False):
    raw = _c(raw)
    h, w = raw.shape
    out = np.empty_like(raw)
    lib(fma).orc_preprocessDepth(_p(raw), w, h, int(ksz), C.c_float(sigma_depth),
                                 C.c_float(sigma_spatial), _p(out))
    return out


# ---- f-4: marching cubes -----------------------------------------------------------------------

def marching_cubes(tsdf, weights, voxel_size, fg=None, grads=None, fma=False):
    """(vertices (n, 3), normals (n, 3), triangles (m, 4)) of the reference's mesh."""
    t, w = _c(tsdf), _c(weights)
    nz, ny, nx = t.shape
    res = (C.c_int * 3)(nx, ny, nz)
    fgp = None if fg is None else _c(fg, np.uint8)
    gp = None if grads is None else _c(grads)
    nv, nt = C.c_int(), C.c_int()
    L = lib(fma)
    L.orc_marchingCubesCount(_p(t), _p(w), None if fgp is None else _p(fgp), res, C.byref(nv), C.byref(nt))
    v = np.empty((nv.value, 3), np.float32)
    n = np.empty((nv.value, 3), np.float32)
    tri = np.empty((nt.value, 4), np.int32)
    L.orc_marchingCubes(_p(t), None if gp is None else _p(gp), _p(w), None if fgp is None else _p(fgp),
                        res, C.c_float(voxel_size), _p(v), _p(n), _p(tri))
    return v, n, tri


def render_phong(points, normals, seg, color_map, light=(0.0, 0.0, 0.0), fma=False):
    p, n = _c(points), _c(normals)
    s, cm = _c(seg, np.uint8), _c(color_map, np.uint8)
    h, w = s.shape
    assert cm.size == 768
    out = np.empty((h, w, 3), np.uint8)
    lib(fma).orc_renderPhong(_p(p), _p(n), _p(s), _p(cm), _farr(light, 3), w, h, _p(out))
    return out
