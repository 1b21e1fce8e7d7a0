"""Thin wrappers that hand device arrays (emfusion_amd.devmem.DeviceArray) to the emf_hip_* C ABI.

Harness-side plumbing only: all arithmetic happens inside libemf_hip.so.  Every wrapper enqueues
on the null stream unless ``stream`` (a raw hipStream_t integer of the product's HIP runtime) is
given, and never synchronises.

Layout conventions (see include/emf_hip.h): images are (H, W) or (H, W, C) arrays, rows possibly
padded (pitch); volumes are contiguous arrays of shape (Nz, Ny, Nx) or (Nz, Ny, Nx, C);
``res`` is (Nx, Ny, Nz).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import EmfImage, check
from .devmem import DeviceArray

_L = _lib.load()


def _stream(stream: Optional[int]) -> C.c_void_p:
    return C.c_void_p(stream or 0)


def _f(values, n: int):
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
    assert a.size == n, f"expected {n} floats, got {a.size}"
    return (C.c_float * n)(*a.tolist())


def _res(t: DeviceArray):
    nz, ny, nx = t.shape[0], t.shape[1], t.shape[2]
    return (C.c_int32 * 3)(nx, ny, nz)


def _ptr(t: Optional[DeviceArray]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.ptr)


def _vol(t: DeviceArray, dtype, channels: int = 1) -> DeviceArray:
    assert isinstance(t, DeviceArray) and t.dtype == np.dtype(dtype) and not t.padded, \
        "volume must be a contiguous device array"
    assert len(t.shape) == (3 if channels == 1 else 4), f"volume rank for {channels} channel(s)"
    if channels > 1:
        assert t.shape[3] == channels
    return t


def image_view(t: DeviceArray) -> EmfImage:
    """emf_image_t over an (H, W[, C]) device array; rows may be padded."""
    assert isinstance(t, DeviceArray) and len(t.shape) in (2, 3)
    return EmfImage(t.ptr, t.pitch, t.shape[1], t.shape[0])


def _views(ts: Sequence[DeviceArray]):
    arr = (EmfImage * max(len(ts), 1))()
    for i, t in enumerate(ts):
        arr[i] = image_view(t)
    return arr


def compute_points(depth, K, points, stream=None):
    check("emf_hip_computePoints",
          _L.emf_hip_computePoints(C.byref(image_view(depth)), C.byref(image_view(points)),
                                   _f(K, 9), _stream(stream)))
    return points


def brick_shape(vol_shape):
    """Shape (2, Bz, By, Bx) of the brick flag buffer of a (Nz, Ny, Nx) volume: [0] raw flags,
    [1] dilated flags."""
    return (2,) + tuple((n + 3) // 4 for n in vol_shape[:3])


def reset_brick_flags(tsdf_like, flags, stream=None):
    check("emf_hip_resetBrickFlags",
          _L.emf_hip_resetBrickFlags(_ptr(flags), _res(tsdf_like), _stream(stream)))
    return flags


def compute_inv_lambda(K, inv_lambda, stream=None):
    """inv_lambda: f32 H x W DeviceArray, written with the per-pixel 1 / lambda table."""
    check("emf_hip_computeInvLambda",
          _L.emf_hip_computeInvLambda(_f(K, 9), C.byref(image_view(inv_lambda)), _stream(stream)))


def _opt_view(img):
    return C.byref(image_view(img)) if img is not None else None


def update_tsdf(depth, assoc, tsdf, weights, R_OC, t_OC, K, voxel_size, truncdist, max_weight,
                brick_flags=None, stream=None, inv_lambda=None):
    _vol(tsdf, np.float32)
    _vol(weights, np.float32)
    if brick_flags is not None:
        assert brick_flags.dtype == np.dtype(np.uint8) and brick_flags.shape == brick_shape(tsdf.shape)
    check("emf_hip_updateTSDF",
          _L.emf_hip_updateTSDF(C.byref(image_view(depth)), C.byref(image_view(assoc)), _ptr(tsdf),
                                _ptr(weights), _ptr(brick_flags), _f(R_OC, 9), _f(t_OC, 3), _f(K, 9), _res(tsdf),
                                voxel_size, truncdist, max_weight, _opt_view(inv_lambda),
                                _stream(stream)))


def compute_tsdf_grads(tsdf, grads, stream=None):
    _vol(tsdf, np.float32)
    _vol(grads, np.float32, 3)
    check("emf_hip_computeTSDFGrads",
          _L.emf_hip_computeTSDFGrads(_ptr(tsdf), _ptr(grads), _res(tsdf), _stream(stream)))


def raycast_tsdf(tsdf, grads, weights, fg_mask, raylengths, vertices, normals, mask, R_CO, t_CO, K,
                 voxel_size, truncdist, stats=None, brick_flags=None, stream=None, rcp_voxel=0.0):
    _vol(tsdf, np.float32)
    _vol(weights, np.float32)
    if grads is not None:
        _vol(grads, np.float32, 3)
    if fg_mask is not None:
        _vol(fg_mask, np.uint8)
    if stats is not None:
        assert stats.dtype == np.dtype(np.uint64) and stats.shape[0] >= 4
    check("emf_hip_raycastTSDF",
          _L.emf_hip_raycastTSDF(_ptr(tsdf), _ptr(grads), _ptr(weights), _ptr(fg_mask),
                                 _ptr(brick_flags), C.byref(image_view(raylengths)), C.byref(image_view(vertices)),
                                 C.byref(image_view(normals)), C.byref(image_view(mask)),
                                 _f(R_CO, 9), _f(t_CO, 3), _f(K, 9), _res(tsdf), voxel_size,
                                 truncdist, float(rcp_voxel), _ptr(stats), _stream(stream)))


def get_volume_vals(vol, points, R_CO, t_CO, voxel_size, vals, stream=None):
    channels = 1 if len(vol.shape) == 3 else vol.shape[3]
    _vol(vol, np.float32, channels)
    check("emf_hip_getVolumeVals",
          _L.emf_hip_getVolumeVals(_ptr(vol), channels, C.byref(image_view(points)), _f(R_CO, 9),
                                   _f(t_CO, 3), _res(vol), voxel_size, C.byref(image_view(vals)),
                                   _stream(stream)))
    return vals


def update_fgbg_probs(mask, occluded, tsdf, weights, fgbg, R_OC, t_OC, K, voxel_size, stream=None):
    _vol(tsdf, np.float32)
    _vol(weights, np.float32)
    _vol(fgbg, np.float32, 2)
    check("emf_hip_updateFgBgProbs",
          _L.emf_hip_updateFgBgProbs(C.byref(image_view(mask)), C.byref(image_view(occluded)),
                                     _ptr(tsdf), _ptr(weights), _ptr(fgbg), _f(R_OC, 9),
                                     _f(t_OC, 3), _f(K, 9), _res(tsdf), voxel_size,
                                     _stream(stream)))


def compute_fg_probs(fgbg, fg_probs, fg_vol_mask, stream=None):
    _vol(fgbg, np.float32, 2)
    _vol(fg_probs, np.float32)
    _vol(fg_vol_mask, np.uint8)
    check("emf_hip_computeFgProbs",
          _L.emf_hip_computeFgProbs(_ptr(fgbg), _ptr(fg_probs), _ptr(fg_vol_mask), _res(fg_probs),
                                    _stream(stream)))


def mask_raycast_weights(weights, fg_vol_mask, out, stream=None):
    _vol(weights, np.float32)
    _vol(fg_vol_mask, np.uint8)
    _vol(out, np.float32)
    check("emf_hip_maskRaycastWeights",
          _L.emf_hip_maskRaycastWeights(_ptr(weights), _ptr(fg_vol_mask), _ptr(out), _res(weights),
                                        _stream(stream)))


def compute_association(tsdf, fg_probs, points, R_CO, t_CO, voxel_size, truncdist, sigma, alpha,
                        uni_prior, out, stream=None):
    _vol(tsdf, np.float32)
    if fg_probs is not None:
        _vol(fg_probs, np.float32)
    check("emf_hip_computeAssociation",
          _L.emf_hip_computeAssociation(_ptr(tsdf), _ptr(fg_probs), C.byref(image_view(points)),
                                        _f(R_CO, 9), _f(t_CO, 3), _res(tsdf), voxel_size,
                                        truncdist, sigma, alpha, uni_prior,
                                        C.byref(image_view(out)), _stream(stream)))
    return out


def normalize_association(maps, extra_sum=None, norm=None, nsum=None, stream=None):
    views = _views(maps)
    nsum = len(maps) if nsum is None else int(nsum)
    ex = C.byref(image_view(extra_sum)) if extra_sum is not None else None
    nr = C.byref(image_view(norm)) if norm is not None else None
    check("emf_hip_normalizeAssociation",
          _L.emf_hip_normalizeAssociation(views, len(maps), nsum, ex, nr, _stream(stream)))


def normalize_association_table(models_dev, nmodels, width, height, norm=None, stream=None):
    """normalize_association(nsum = all) over the `assoc` maps of a device model table, one launch."""
    check("emf_hip_normalizeAssociationTable",
          _L.emf_hip_normalizeAssociationTable(_ptr(models_dev), int(nmodels), int(width), int(height), _ptr(norm),
                                               _stream(stream)))


def sum_association(maps, out, stream=None):
    views = _views(maps)
    check("emf_hip_sumAssociation",
          _L.emf_hip_sumAssociation(views, len(maps), C.byref(image_view(out)), _stream(stream)))
    return out


def composite_visibility(ids, obj_ray, obj_vert, obj_norm, obj_seg, bg_ray, bg_vert, bg_norm, bg_mask,
                         ray, vert, norm, seg, diff, no_obj, boundary, vis_counts, thresh, visible, mirror=None,
                         stream=None):
    """emf_hip_compositeVisibility: vis_counts must hold zeros and holds zeros again afterwards; the numbers go
    to `mirror` (int32 device array of len(ids)) and into `visible` (int32, len(ids) + 1)."""
    n = len(ids)
    ids_arr = (C.c_int32 * max(n, 1))(*[int(i) for i in ids])
    check("emf_hip_compositeVisibility",
          _L.emf_hip_compositeVisibility(n, ids_arr, _views(obj_ray), _views(obj_vert),
                                         _views(obj_norm), _views(obj_seg),
                                         C.byref(image_view(bg_ray)), C.byref(image_view(bg_vert)),
                                         C.byref(image_view(bg_norm)), C.byref(image_view(bg_mask)),
                                         C.byref(image_view(ray)), C.byref(image_view(vert)),
                                         C.byref(image_view(norm)), C.byref(image_view(seg)),
                                         C.byref(image_view(diff)), C.byref(image_view(no_obj)),
                                         boundary, _ptr(vis_counts), int(thresh), _ptr(visible),
                                         _ptr(mirror) if mirror is not None else None, _stream(stream)))


def composite_raycast(ids, obj_ray, obj_vert, obj_norm, obj_seg, bg_ray, bg_vert, bg_norm, bg_mask,
                      ray, vert, norm, seg, diff, no_obj, boundary, vis_counts, stream=None):
    n = len(ids)
    ids_arr = (C.c_int32 * max(n, 1))(*[int(i) for i in ids])
    if n:
        assert vis_counts.dtype == np.dtype(np.int32) and vis_counts.shape[0] >= n
    check("emf_hip_compositeRaycast",
          _L.emf_hip_compositeRaycast(n, ids_arr, _views(obj_ray), _views(obj_vert),
                                      _views(obj_norm), _views(obj_seg),
                                      C.byref(image_view(bg_ray)), C.byref(image_view(bg_vert)),
                                      C.byref(image_view(bg_norm)), C.byref(image_view(bg_mask)),
                                      C.byref(image_view(ray)), C.byref(image_view(vert)),
                                      C.byref(image_view(norm)), C.byref(image_view(seg)),
                                      C.byref(image_view(diff)), C.byref(image_view(no_obj)),
                                      boundary, _ptr(vis_counts if n else None), _stream(stream)))


def occluded_mask(obj_seg, seg, obj_id, occluded, stream=None):
    check("emf_hip_occludedMask",
          _L.emf_hip_occludedMask(C.byref(image_view(obj_seg)), C.byref(image_view(seg)),
                                  int(obj_id), C.byref(image_view(occluded)), _stream(stream)))
    return occluded


def stream_copy(dst, src, stream=None):
    """dst <- src with the plain copy kernel (both DeviceArrays of equal byte size)."""
    assert dst.nbytes == src.nbytes
    check("emf_hip_streamCopy", _L.emf_hip_streamCopy(_ptr(dst), _ptr(src), dst.nbytes, _stream(stream)))


def l1_gather_probe(buf, footprint_bytes, lines, iterations, workgroups, sink, stream=None):
    """emf_hip_l1GatherProbe: the vector L1's gather rate on a resident footprint (bench.py's calibration)."""
    check("emf_hip_l1GatherProbe", _L.emf_hip_l1GatherProbe(_ptr(buf), int(footprint_bytes), int(lines), int(iterations),
                                                            int(workgroups), _ptr(sink), _stream(stream)))


def device_info():
    name = C.create_string_buffer(256)
    arch = C.create_string_buffer(256)
    cus = C.c_int(0)
    check("emf_hip_device_info", _L.emf_hip_device_info(name, 256, arch, 256, C.byref(cus)))
    return name.value.decode(), arch.value.decode(), cus.value


# ---- level 3: batched, model-table driven launches ----------------------------------------------

def make_model(tsdf, weights, assoc, raylengths, vertices, normals, hit_mask, voxel_size,
               truncdist, max_weight, sigma, alpha, uni_prior, model_id=0, grads=None,
               fg_probs=None, fg_mask=None, brick_flags=None, rcp_voxel=0.0, sign_maps=None, relevant_tiles=None,
               unseen_tiles=None) -> "_lib.EmfModel":
    """Fill an emf_model_t from device arrays (images must be unpadded)."""
    f32 = np.float32
    m = _lib.EmfModel()
    m.tsdf, m.weights = tsdf.ptr, weights.ptr
    m.grads = grads.ptr if grads is not None else None
    m.fgProbs = fg_probs.ptr if fg_probs is not None else None
    m.fgVolMask = fg_mask.ptr if fg_mask is not None else None
    m.brickFlags = brick_flags.ptr if brick_flags is not None else None
    m.signMaps = sign_maps.ptr if sign_maps is not None else None
    m.relevantTiles = relevant_tiles.ptr if relevant_tiles is not None else None
    m.unseenTiles = unseen_tiles.ptr if unseen_tiles is not None else None
    for name, im in (("assoc", assoc), ("raylengths", raylengths), ("vertices", vertices),
                     ("normals", normals), ("hitMask", hit_mask)):
        assert not im.padded
        setattr(m, name, im.ptr)
    nz, ny, nx = tsdf.shape
    m.res[:] = [nx, ny, nz]
    m.id = model_id
    m.voxelSize, m.truncdist, m.maxWeight = voxel_size, truncdist, max_weight
    m.assocC1 = float(-f32(truncdist) / f32(sigma))
    m.assocC2 = float(f32(1) / (f32(2) * f32(sigma)))
    m.alpha = alpha
    m.assocC3 = float((f32(1) - f32(alpha)) * f32(uni_prior))
    m.rcpVoxel = rcp_voxel  # 0, or ops.voxel_reciprocal(voxel_size)
    return m


def upload_models(models) -> DeviceArray:
    arr = (_lib.EmfModel * len(models))(*models)
    raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
    return DeviceArray.from_numpy(raw)


def _poses(poses):
    arr = (_lib.EmfPose * max(len(poses), 1))()
    for i, (R, t) in enumerate(poses):
        arr[i].R[:] = np.asarray(R, np.float32).reshape(-1).tolist()
        arr[i].t[:] = np.asarray(t, np.float32).reshape(-1).tolist()
    return arr


def estep_batched(models_dev, poses_co, points, normalize=True, norm=None, obj_sum=None,
                  stream=None):
    check("emf_hip_estepBatched",
          _L.emf_hip_estepBatched(_ptr(models_dev), _poses(poses_co), len(poses_co),
                                  C.byref(image_view(points)), int(normalize),
                                  C.byref(image_view(norm)) if norm is not None else None,
                                  C.byref(image_view(obj_sum)) if obj_sum is not None else None,
                                  _stream(stream)))


def estep_batched_from_depth(models_dev, poses_co, depth, K, points, normalize=True, norm=None, obj_sum=None,
                             stream=None):
    """compute_points + estep_batched in one launch; `points` is written."""
    check("emf_hip_estepBatchedFromDepth",
          _L.emf_hip_estepBatchedFromDepth(_ptr(models_dev), _poses(poses_co), len(poses_co),
                                           C.byref(image_view(depth)), _f(K, 9), C.byref(image_view(points)),
                                           int(normalize),
                                           C.byref(image_view(norm)) if norm is not None else None,
                                           C.byref(image_view(obj_sum)) if obj_sum is not None else None,
                                           _stream(stream)))


def voxel_reciprocal(voxel_size) -> float:
    """1 / voxel_size if the device check finds it usable in place of x / voxel_size, else 0."""
    r = C.c_float(0.0)
    check("emf_hip_voxelReciprocal", _L.emf_hip_voxelReciprocal(float(voxel_size), C.byref(r)))
    return float(r.value)


def voxel_reciprocal_exhaustive(voxel_size) -> int:
    """Test aid: the number of floats x, 1e-30 <= |x| <= 1e30, on which the reciprocal form and the division differ,
    counted over all 2^32 bit patterns (2.3 ms of the whole chip; nothing is cached)."""
    n = C.c_ulonglong(0)
    check("emf_hip_voxelReciprocalExhaustive", _L.emf_hip_voxelReciprocalExhaustive(float(voxel_size), C.byref(n)))
    return int(n.value)


def raycast_batched(models_dev, poses_co, res_list, width, height, K, stats=None,
                    use_brick_flags=False, stream=None, bg_band=(0, 0), far_bounds=None, voxel_sizes=None, lanes=1,
                    objects_only=False):
    """bg_band = (row0, rows): march only that row band of table slot 0 (multi-GPU background split).
    far_bounds: raycast_far_bounds()'s array for the same table, poses and image (same results, shorter marches).
    lanes: 1 / 2 / 4 lanes per background ray.  objects_only: the table (chunk) holds no background in slot 0."""
    res = (C.c_int32 * (3 * len(poses_co)))(*[int(v) for r in res_list for v in r])
    vox = None if voxel_sizes is None else _f(voxel_sizes, len(poses_co))
    if objects_only:
        check("emf_hip_raycastBatchedObjects",
              _L.emf_hip_raycastBatchedObjects(_ptr(models_dev), _poses(poses_co), res, len(poses_co), width, height,
                                               _f(K, 9), int(use_brick_flags), _ptr(far_bounds), vox, _ptr(stats),
                                               _stream(stream)))
        return
    check("emf_hip_raycastBatchedLanes",
          _L.emf_hip_raycastBatchedLanes(_ptr(models_dev), _poses(poses_co), res, len(poses_co), width,
                                         height, _f(K, 9), int(use_brick_flags), int(bg_band[0]),
                                         int(bg_band[1]), _ptr(far_bounds), vox, int(lanes), _ptr(stats),
                                         _stream(stream)))


def sign_map_bytes(res) -> int:
    return int(_L.emf_hip_signMapBytes((C.c_int32 * 3)(*[int(v) for v in res])))


def rebuild_sign_maps(tsdf, sign_maps, stream=None):
    nz, ny, nx = tsdf.shape
    check("emf_hip_rebuildSignMaps",
          _L.emf_hip_rebuildSignMaps(_ptr(tsdf), (C.c_int32 * 3)(nx, ny, nz), _ptr(sign_maps), _stream(stream)))


def unseen_tile_bytes(res) -> int:
    return int(_L.emf_hip_unseenTileBytes((C.c_int32 * 3)(*[int(v) for v in res])))


def rebuild_unseen_tiles(tsdf, weights, unseen_tiles, stream=None):
    nz, ny, nx = tsdf.shape
    check("emf_hip_rebuildUnseenTiles",
          _L.emf_hip_rebuildUnseenTiles(_ptr(tsdf), _ptr(weights), (C.c_int32 * 3)(nx, ny, nz), _ptr(unseen_tiles),
                                        _stream(stream)))


def raycast_far_bounds(models_dev, poses_co, res_list, width, height, K, bounds=None, stream=None, scan_mask=0xffffffff):
    """emf_hip_raycastFarBounds -> float32 (nmodels, cellsY, cellsX) device array."""
    n = len(poses_co)
    res = (C.c_int32 * (3 * n))(*[int(v) for r in res_list for v in r])
    if bounds is None:
        cy, cx = 2 * ((height + 15) // 16), 2 * ((width + 15) // 16)
        assert int(_L.emf_hip_raycastFarBoundBytes(n, width, height)) == 4 * n * cy * cx
        bounds = DeviceArray.zeros((n, cy, cx), np.float32)
    check("emf_hip_raycastFarBounds",
          _L.emf_hip_raycastFarBounds(_ptr(models_dev), _poses(poses_co), res, n, width, height, _f(K, 9), int(scan_mask),
                                      _ptr(bounds), _stream(stream)))
    return bounds


def relevant_tile_words(res) -> int:
    return int(_L.emf_hip_relevantTileBytes((C.c_int32 * 3)(*[int(v) for v in res]))) // 4


def update_relevant_tiles(models_dev, res_list, stream=None):
    n = len(res_list)
    res = (C.c_int32 * (3 * n))(*[int(v) for r in res_list for v in r])
    check("emf_hip_updateRelevantTiles", _L.emf_hip_updateRelevantTiles(_ptr(models_dev), res, n, _stream(stream)))


def integrate_batched(models_dev, poses_oc, res_list, visible, depth, K, stats=None, stream=None,
                      inv_lambda=None):
    res = (C.c_int32 * (3 * len(poses_oc)))(*[int(v) for r in res_list for v in r])
    check("emf_hip_integrateBatched",
          _L.emf_hip_integrateBatched(_ptr(models_dev), _poses(poses_oc), res, len(poses_oc),
                                      _ptr(visible), C.byref(image_view(depth)),
                                      _opt_view(inv_lambda), _f(K, 9), 1, _ptr(stats),
                                      _stream(stream)))


def integrate_batched_culled(models_dev, poses_oc, res_list, visible, depth, K, launch_boxes=0, survivors=None,
                             stats=None, stream=None, inv_lambda=None, scratch=None):
    """emf_hip_integrateBatchedCulled; returns the scratch buffer (reusable)."""
    res = (C.c_int32 * (3 * len(poses_oc)))(*[int(v) for r in res_list for v in r])
    if scratch is None:
        scratch = DeviceArray.zeros((int(_L.emf_hip_integrateCullScratchBytes(res, len(poses_oc))) // 4,), np.uint32)
    check("emf_hip_integrateBatchedCulled",
          _L.emf_hip_integrateBatchedCulled(_ptr(models_dev), _poses(poses_oc), res, len(poses_oc), _ptr(visible),
                                            C.byref(image_view(depth)), _opt_view(inv_lambda), _f(K, 9),
                                            _ptr(scratch), int(launch_boxes), _ptr(survivors), _ptr(stats),
                                            _stream(stream)))
    return scratch


def integrate_dirty_map_bytes(res) -> int:
    return int(_L.emf_hip_integrateDirtyMapBytes((C.c_int32 * 3)(*[int(v) for v in res])))


def integrate_prepare_out(outs, res_list, scratch, stream=None):
    """emf_hip_integratePrepareOut: clear the survivor counter and the dirtyNext maps of outs ahead of time."""
    from ._lib import EmfVolumeOut
    res = (C.c_int32 * (3 * len(res_list)))(*[int(v) for r in res_list for v in r])
    table = (EmfVolumeOut * len(outs))()
    for o, (t, w, dp, dn) in zip(table, outs):
        o.tsdf, o.weights, o.dirtyPrev, o.dirtyNext = t.ptr, w.ptr, dp.ptr, dn.ptr
    check("emf_hip_integratePrepareOut",
          _L.emf_hip_integratePrepareOut(C.cast(table, C.c_void_p), res, len(outs), _ptr(scratch), _stream(stream)))


def integrate_batched_culled_out(models_dev, poses_oc, res_list, visible, depth, K, outs, launch_boxes=0,
                                 stats=None, stream=None, inv_lambda=None, scratch=None, prepared=False):
    """emf_hip_integrateBatchedCulledOut: model m is read from the table and written to outs[m] =
    (tsdf_back, weights_back, dirty_prev, dirty_next) device arrays; returns the scratch buffer."""
    from ._lib import EmfVolumeOut
    res = (C.c_int32 * (3 * len(poses_oc)))(*[int(v) for r in res_list for v in r])
    if scratch is None:
        scratch = DeviceArray.zeros((int(_L.emf_hip_integrateCullScratchBytes(res, len(poses_oc))) // 4,), np.uint32)
    table = (EmfVolumeOut * len(outs))()
    for o, (t, w, dp, dn) in zip(table, outs):
        o.tsdf, o.weights, o.dirtyPrev, o.dirtyNext = t.ptr, w.ptr, dp.ptr, dn.ptr
    check("emf_hip_integrateBatchedCulledOut",
          _L.emf_hip_integrateBatchedCulledOut(_ptr(models_dev), _poses(poses_oc), res, len(poses_oc), _ptr(visible),
                                               C.byref(image_view(depth)), _opt_view(inv_lambda), _f(K, 9),
                                               C.cast(table, C.c_void_p), int(prepared), _ptr(scratch), int(launch_boxes), None,
                                               _ptr(stats), _stream(stream)))
    return scratch


def visibility_flags(vis_counts, nmodels, thresh, visible, stream=None):
    check("emf_hip_visibilityFlags",
          _L.emf_hip_visibilityFlags(_ptr(vis_counts), nmodels, thresh, _ptr(visible), None,
                                     _stream(stream)))


# ---- cross-GPU compositing ------------------------------------------------------------------------

def pack_hit_keys(list_pos, obj_ray, obj_seg, keys, width, height, stream=None):
    n = len(list_pos)
    pos = (C.c_int32 * max(n, 1))(*[int(p) for p in list_pos])
    assert keys.dtype == np.dtype(np.uint64)
    check("emf_hip_packHitKeys",
          _L.emf_hip_packHitKeys(n, pos, _views(obj_ray), _views(obj_seg), _ptr(keys), width,
                                 height, _stream(stream)))
    return keys


def composite_from_keys(keys, ids_all, list_pos, obj_ray, obj_vert, obj_norm, bg_ray, bg_vert,
                        bg_norm, bg_mask, ray, vert, norm, seg, diff, no_obj, boundary, vis_counts,
                        stream=None):
    nall, n = len(ids_all), len(list_pos)
    ids = (C.c_int32 * max(nall, 1))(*[int(i) for i in ids_all])
    pos = (C.c_int32 * max(n, 1))(*[int(p) for p in list_pos])
    check("emf_hip_compositeFromKeys",
          _L.emf_hip_compositeFromKeys(_ptr(keys), nall, ids, n, pos, _views(obj_ray),
                                       _views(obj_vert), _views(obj_norm),
                                       C.byref(image_view(bg_ray)), C.byref(image_view(bg_vert)),
                                       C.byref(image_view(bg_norm)), C.byref(image_view(bg_mask)),
                                       C.byref(image_view(ray)), C.byref(image_view(vert)),
                                       C.byref(image_view(norm)), C.byref(image_view(seg)),
                                       C.byref(image_view(diff)), C.byref(image_view(no_obj)),
                                       boundary, _ptr(vis_counts if nall else None),
                                       _stream(stream)))


def visibility_flags_indexed(vis_counts, count_index, thresh, visible, stream=None):
    n = len(count_index)
    idx = (C.c_int32 * max(n, 1))(*[int(i) for i in count_index])
    check("emf_hip_visibilityFlagsIndexed",
          _L.emf_hip_visibilityFlagsIndexed(_ptr(vis_counts), n, idx, thresh, _ptr(visible),
                                            _stream(stream)))


# ---- tracking (SURVEY f-1) ----------------------------------------------------------------------

def compute_pose_gradients(tsdf, grads, points, R_CO, t_CO, voxel_size, out, stream=None):
    """out: float32 DeviceArray (H * W, 6); grads: gradient volume or None (on the fly)."""
    _vol(tsdf, np.float32)
    check("emf_hip_computePoseGradients",
          _L.emf_hip_computePoseGradients(_ptr(tsdf), _ptr(grads), C.byref(image_view(points)),
                                          _f(R_CO, 9), _f(t_CO, 3), _res(tsdf), voxel_size,
                                          _ptr(out), _stream(stream)))


def track_scratch_bytes(width, height) -> int:
    return int(_L.emf_hip_trackScratchBytes(int(width), int(height)))


def track_prepare(states_dev, poses_co, nu_init=2.0, stream=None):
    """states_dev: uint8 DeviceArray of len(poses_co) * sizeof(EmfTrackState) bytes."""
    check("emf_hip_trackPrepare",
          _L.emf_hip_trackPrepare(_ptr(states_dev), _poses(poses_co), len(poses_co), nu_init,
                                  _stream(stream)))


def track_iterate(models_dev, states_dev, nmodels, points, params, scratch, scratch_per_model,
                  iterations=1, stream=None):
    check("emf_hip_trackIterate",
          _L.emf_hip_trackIterate(_ptr(models_dev), _ptr(states_dev), nmodels,
                                  C.byref(image_view(points)), C.byref(params), _ptr(scratch),
                                  scratch_per_model, iterations, _stream(stream)))


def track_step(models_dev, states_dev, nmodels, points, params, scratch, scratch_per_model, launch,
               iterations, watch=None, seq=0, stream=None, final_states=None):
    """One launch of the LM step kernel (emf_hip_trackStep); watch: address of host-pinned uint32s or None;
    final_states: address of nmodels EmfTrackState the device can write (a done model's state, ahead of its word) or None."""
    check("emf_hip_trackStep",
          _L.emf_hip_trackStep(_ptr(models_dev), _ptr(states_dev), nmodels, C.byref(image_view(points)),
                               C.byref(params), _ptr(scratch), scratch_per_model, launch, iterations,
                               watch, seq, final_states, _stream(stream)))


def track_weight_images(models_dev, states_dev, nmodels, points, params, scratch, scratch_per_model,
                        huber=None, track=None, stream=None):
    """Huber and combined tracking weights of a finished stage at its final pose (emf_hip_trackWeightImages);
    huber / track: device arrays of nmodels x H x W floats or None."""
    check("emf_hip_trackWeightImages",
          _L.emf_hip_trackWeightImages(_ptr(models_dev), _ptr(states_dev), nmodels, C.byref(image_view(points)),
                                       C.byref(params), _ptr(scratch), scratch_per_model,
                                       _ptr(huber) if huber is not None else None,
                                       _ptr(track) if track is not None else None, _stream(stream)))


def read_track_states(states_dev, nmodels):
    """Synchronise and return the device LM states as a list of EmfTrackState."""
    raw = states_dev.numpy().tobytes()
    n = C.sizeof(_lib.EmfTrackState)
    return [_lib.EmfTrackState.from_buffer_copy(raw[i * n:(i + 1) * n]) for i in range(nmodels)]


# ---- depth pre-processing (SURVEY f-2) -----------------------------------------------------------

def preprocess_depth(raw, out, ksz=7, sigma_depth=0.04, sigma_spatial=4.5, stream=None):
    check("emf_hip_preprocessDepth",
          _L.emf_hip_preprocessDepth(C.byref(image_view(raw)), C.byref(image_view(out)), int(ksz),
                                     sigma_depth, sigma_spatial, _stream(stream)))


# ---- object creation / matching from masks (SURVEY f-3) ------------------------------------------

def masked_point_stats(points, mask, R, t, stream=None):
    """(count, p10[3], p90[3]) of the valid masked points after x' = R x + t (synchronises)."""
    scratch = DeviceArray.zeros((int(_L.emf_hip_pointStatsScratchBytes()) // 4,), np.uint32)
    out = DeviceArray.zeros((7,), np.float32)
    check("emf_hip_maskedPointStats",
          _L.emf_hip_maskedPointStats(C.byref(image_view(points)), C.byref(image_view(mask)), _f(R, 9),
                                      _f(t, 3), _ptr(scratch), _ptr(out), _stream(stream)))
    raw = out.numpy()
    return int(raw.view(np.uint32)[0]), raw[1:4].copy(), raw[4:7].copy()


def mask_overlap(seg, model_seg, stream=None):
    """(mask pixels, intersection[256], area[256]) against every id of the model segmentation."""
    counts = DeviceArray.zeros((513,), np.uint32)
    check("emf_hip_maskOverlap",
          _L.emf_hip_maskOverlap(C.byref(image_view(seg)), C.byref(image_view(model_seg)),
                                 _ptr(counts), _stream(stream)))
    c = counts.numpy()
    return int(c[0]), c[1:257].copy(), c[257:513].copy()


def mask_association_mass(obj_seg, match_mask, assoc, stream=None):
    """(count, sum) of cleanUpObjs' association test; match_mask may be None (synchronises)."""
    out = DeviceArray.zeros((int(_L.emf_hip_maskAssociationMassBytes()) // 8,), np.float64)
    check("emf_hip_maskAssociationMass",
          _L.emf_hip_maskAssociationMass(C.byref(image_view(obj_seg)), _opt_view(match_mask),
                                         C.byref(image_view(assoc)), _ptr(out), _stream(stream)))
    raw = out.numpy()
    return int(raw.view(np.uint32)[2]), float(raw[0])


def carve_mask(seg, model_seg, obj_id, match_mask=None, stream=None):
    """seg &= !((model_seg == obj_id) | match_mask) in place; returns (pixels before, after)."""
    counts = DeviceArray.zeros((2,), np.uint32)
    check("emf_hip_carveMask",
          _L.emf_hip_carveMask(C.byref(image_view(seg)), C.byref(image_view(model_seg)), int(obj_id),
                               _opt_view(match_mask), _ptr(counts), _stream(stream)))
    c = counts.numpy()
    return int(c[0]), int(c[1])


def object_extent_stats(points, mask, R, t, tsdf, weights, fg_mask, voxel_size, stream=None):
    """updateObj's statistics: masked points (R x + t) plus the object's iso-surface vertex cloud."""
    scratch = DeviceArray.zeros((int(_L.emf_hip_pointStatsScratchBytes()) // 4,), np.uint32)
    out = DeviceArray.zeros((7,), np.float32)
    check("emf_hip_objectExtentStats",
          _L.emf_hip_objectExtentStats(C.byref(image_view(points)), C.byref(image_view(mask)), _f(R, 9),
                                       _f(t, 3), _ptr(tsdf), _ptr(weights), _ptr(fg_mask), _res(tsdf),
                                       voxel_size, _ptr(scratch), _ptr(out), _stream(stream)))
    raw = out.numpy()
    return int(raw.view(np.uint32)[0]), raw[1:4].copy(), raw[4:7].copy()


def render_phong(vertices, normals, segmentation, color_map, image, light=(0.0, 0.0, 0.0), stream=None):
    """renderGPU: Phong-shaded RGB image (H, W, 3) u8 of the composited raycast; color_map (256, 3) u8 host."""
    cm = np.ascontiguousarray(color_map, np.uint8)
    assert cm.size == 768
    check("emf_hip_renderPhong",
          _L.emf_hip_renderPhong(C.byref(image_view(vertices)), C.byref(image_view(normals)),
                                 C.byref(image_view(segmentation)), cm.ctypes.data, _f(light, 3),
                                 C.byref(image_view(image)), _stream(stream)))
    return image


def extract_mesh(tsdf, weights, voxel_size, fg_mask=None, grads=None, stream=None):
    """TSDF::getMesh / ObjTSDF::getMesh: (vertices (n, 3) f32, normals (n, 3) f32, triangles (m, 4) i32)
    as numpy arrays; two launches to count, one read-back, one launch to emit."""
    res = _res(tsdf)
    scratch = DeviceArray.zeros((max(int(_L.emf_hip_meshScratchBytes(res)) // 4, 2),), np.uint32)
    counts = DeviceArray.zeros((2,), np.uint32)
    check("emf_hip_meshCount",
          _L.emf_hip_meshCount(_ptr(tsdf), _ptr(weights), _ptr(fg_mask), res, _ptr(scratch), _ptr(counts),
                               _stream(stream)))
    nv, nt = (int(v) for v in counts.numpy())
    verts = DeviceArray.zeros((max(nv, 1), 3), np.float32)
    norms = DeviceArray.zeros((max(nv, 1), 3), np.float32)
    tris = DeviceArray.zeros((max(nt, 1), 4), np.int32)
    if nv:
        check("emf_hip_meshEmit",
              _L.emf_hip_meshEmit(_ptr(tsdf), _ptr(grads), _ptr(weights), _ptr(fg_mask), res, voxel_size,
                                  _ptr(scratch), _ptr(verts), _ptr(norms), _ptr(tris), _stream(stream)))
    return verts.numpy()[:nv], norms.numpy()[:nv], tris.numpy()[:nt]


def copy_values(src, dst, offset, stream=None):
    """dst(v) = src(v + offset) inside src, else 0 (kernel_copyValues); volumes (Nz, Ny, Nx[, C])."""
    ch = 1 if len(src.shape) == 3 else src.shape[3]
    check("emf_hip_copyValues",
          _L.emf_hip_copyValues(_ptr(src), _ptr(dst), ch, (C.c_int32 * 3)(*[int(v) for v in offset]),
                                _res(src), _res(dst), _stream(stream)))
