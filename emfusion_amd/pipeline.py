"""ctypes binding of libemf_fusion.so (include/emf_fusion.h): the C++ host classes
emf::EMFusion / TSDF / ObjTSDF, the RCCL communicator and the synthetic RGB-D stream.

Harness-side only (tests/, bench.py).  All per-frame work happens in C++/HIP; this module moves
pointers.  Fails loudly if the library is missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path
from typing import Dict, Optional, Sequence

import numpy as np

from . import devmem
from ._lib import EmfImage, PKG_DIR, REPO_ROOT

# EMF_FUSION_VARIANT=_dbg loads libemf_fusion_dbg.so (the host classes built with -DEMF_DEBUG_SWITCHES: test infrastructure)
LIB_PATH = PKG_DIR / ("libemf_fusion%s.so" % os.environ.get("EMF_FUSION_VARIANT", ""))
HEADER_PATH = REPO_ROOT / "include" / "emf_fusion.h"


class FusionParams(C.Structure):
    """Mirror of emf_fusion_params_t."""

    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("K", C.c_float * 9),
                ("bg_res", C.c_int32 * 3), ("bg_voxel_size", C.c_float),
                ("bg_rel_truncdist", C.c_float), ("volume_pose_t", C.c_float * 3),
                ("obj_res", C.c_int32 * 3), ("obj_rel_truncdist", C.c_float),
                ("max_tsdf_weight", C.c_float), ("assoc_sigma", C.c_float), ("alpha", C.c_float),
                ("uni_prior", C.c_float), ("visibility_thresh", C.c_int32),
                ("boundary", C.c_int32), ("mask_frames", C.c_int32),
                ("materialize_gradients", C.c_int32), ("max_tracking_iter", C.c_int32)]


class FrameTimings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("points", "estep", "raycast", "composite", "integrate",
                                         "masks", "total")]

    def as_dict(self) -> Dict[str, float]:
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


class KernelSummary(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("total_ms", C.c_double), ("units", C.c_double)]


KERNEL_KINDS = ("points", "assoc", "normalize", "raycast", "composite", "integrate", "grads", "fgbg",
                "track", "integrate_bg")

IMG = dict(points=0, bg_assoc=1, obj_assoc=2, assoc_norm=3, raylengths=4, vertices=5, normals=6,
           segmentation=7, bg_raylengths=8, obj_raylengths=9)
_IMG_DTYPE = {0: ("float32", 3), 1: ("float32", 1), 2: ("float32", 1), 3: ("float32", 1),
              4: ("float32", 1), 5: ("float32", 3), 6: ("float32", 3), 7: ("uint8", 1),
              8: ("float32", 1), 9: ("float32", 1)}
VOL = dict(tsdf=0, weights=1, fgprobs=2, fgmask=3, bricks=4)

_lib = None


class FusionError(RuntimeError):
    def __init__(self, fn, code, msg):
        super().__init__(f"{fn} failed with {code}: {msg}")
        self.code = code


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER_PATH.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(emf_(?:fusion|comm|synth)_\w+)\s*\(", text)))


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C {PKG_DIR / 'csrc'}` "
                           "(or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(os.fspath(LIB_PATH))
    lib.emf_fusion_last_error_string.restype = C.c_char_p
    lib.emf_fusion_default_params.restype = None
    lib.emf_fusion_destroy.restype = None
    lib.emf_comm_destroy.restype = None
    lib.emf_synth_destroy.restype = None
    vp = C.c_void_p
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int32)
    img = C.POINTER(EmfImage)
    sigs = {
        "emf_fusion_default_params": [C.POINTER(FusionParams)],
        "emf_fusion_create": [C.POINTER(FusionParams), vp, C.POINTER(vp)],
        "emf_fusion_destroy": [vp],
        "emf_fusion_reset": [vp],
        "emf_fusion_trim_pool": [C.POINTER(C.c_uint64)],
        "emf_fusion_process_rgbd": [vp, fp, C.c_int32, C.c_int32],
        "emf_fusion_use_preproc_masks": [vp, C.c_char_p],
        "emf_fusion_get_last_masks": [vp, C.c_void_p, C.c_size_t, ip],
        "emf_io_read_depth_png": [C.c_char_p, C.c_float, fp, C.c_size_t, ip, ip],
        "emf_io_read_exr": [C.c_char_p, C.c_char_p, fp, C.c_size_t, ip, ip],
        "emf_io_load_config": [C.c_char_p, C.c_char_p, C.c_void_p, C.c_char_p, C.c_size_t],
        "emf_io_image_reader": [C.c_char_p, C.c_char_p, C.c_char_p, ip, ip],
        "emf_io_tum_associations": [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), ip],
        "emf_io_load_preproc_masks": [C.c_char_p, ip, ip, ip, C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.c_size_t,
                                      C.POINTER(C.c_double), C.c_size_t, ip],
        "emf_fusion_create_from_config": [C.c_char_p, C.c_char_p, C.c_int, vp, C.c_void_p, C.POINTER(vp)],
        "emf_fusion_add_object": [vp, fp, C.c_float, ip],
        "emf_fusion_process_frame": [vp, img, fp, fp, C.c_int, ip, fp, fp, C.c_int, ip, img,
                                     C.c_int],
        "emf_fusion_set_tracking": [vp, C.c_int, C.c_int],
        "emf_fusion_set_preprocess": [vp, C.c_int],
        "emf_fusion_set_cleanup": [vp, C.c_int],
        "emf_fusion_enable_pose_log": [vp, C.c_int],
        "emf_fusion_setup_output": [vp, C.c_int, C.c_int],
        "emf_fusion_write_results": [vp, C.c_char_p, C.c_int],
        "emf_io_png_unfilter": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
        "emf_io_write_volume": [C.c_char_p, fp, ip, C.c_float],
        "emf_io_write_pose_file": [C.c_char_p, C.c_int, ip, fp, fp],
        "emf_fusion_last_deleted": [vp, ip, C.c_int, ip],
        "emf_fusion_create_object_from_mask": [vp, img, ip],
        "emf_fusion_match_mask": [vp, img, ip, fp],
        "emf_fusion_update_object": [vp, C.c_int, img, fp],
        "emf_fusion_set_depth_broadcast": [vp, C.c_int],
        "emf_fusion_queue_instance_scores": [vp, C.c_int, C.c_int, C.c_void_p],
        "emf_fusion_object_class": [vp, C.c_int, ip],
        "emf_fusion_set_ignore_person": [vp, C.c_int],
        "emf_fusion_object_info": [vp, C.c_int, ip, fp, fp, fp],
        "emf_fusion_render": [vp, C.c_void_p, C.c_void_p],
        "emf_fusion_extract_mesh": [vp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
        "emf_fusion_copy_mesh": [vp, C.c_void_p, C.c_void_p, C.c_void_p],
        "emf_io_write_mesh": [C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p],
        "emf_fusion_queue_new_object_masks": [vp, C.c_int, img],
        "emf_fusion_last_created": [vp, ip, C.c_int, ip],
        "emf_fusion_queue_instance_masks": [vp, C.c_int, img],
        "emf_fusion_last_mask_assignment": [vp, ip, C.c_int, ip],
        "emf_fusion_get_pose": [vp, C.c_int, fp, fp],
        "emf_fusion_track_result": [vp, C.c_int, ip, ip, ip, fp],
        "emf_fusion_stage_estep": [vp],
        "emf_fusion_stage_raycast": [vp],
        "emf_fusion_stage_integrate": [vp],
        "emf_fusion_synchronize": [vp],
        "emf_fusion_enable_timings": [vp, C.c_int],
        "emf_fusion_last_timings": [vp, C.POINTER(FrameTimings)],
        "emf_fusion_enable_raycast_stats": [vp, C.c_int],
        "emf_fusion_raycast_stats": [vp, C.POINTER(C.c_uint64)],
        "emf_fusion_kernel_timers_enable": [vp, C.c_uint64],
        "emf_fusion_kernel_timers_clear": [vp],
        "emf_fusion_kernel_timers_select": [vp, C.c_uint32],
        "emf_fusion_kernel_timers_stride": [vp, C.c_uint32],
        "emf_fusion_kernel_timers_collect": [vp, C.POINTER(KernelSummary), C.POINTER(C.c_uint64)],
        "emf_fusion_get_image": [vp, C.c_int, C.c_int, img],
        "emf_fusion_get_volume": [vp, C.c_int, C.c_int, C.POINTER(vp), ip],
        "emf_fusion_visible_objects": [vp, ip, C.c_int, C.POINTER(C.c_int)],
        "emf_fusion_object_ids": [vp, ip, C.c_int, C.POINTER(C.c_int)],
        "emf_fusion_frame_index": [vp],
        "emf_fusion_background_overlap": [vp],
        "emf_fusion_batched_chunks": [vp],
        "emf_fusion_upload_host_time": [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)],
        "emf_fusion_owns_object": [vp, C.c_int],
        "emf_comm_describe": [vp, C.c_char_p, C.c_size_t],
        "emf_comm_unique_id": [vp],
        "emf_comm_create": [vp, C.c_int, C.c_int, C.POINTER(vp)],
        "emf_comm_destroy": [vp],
        "emf_comm_create_local_group": [C.c_int, C.POINTER(vp)],
        "emf_comm_create_delayed": [vp, C.c_int, C.POINTER(vp)],
        "emf_comm_all_reduce_sum_f32": [vp, vp, C.c_size_t, vp],
        "emf_comm_all_reduce_min_u64": [vp, vp, C.c_size_t, vp],
        "emf_comm_broadcast": [vp, vp, C.c_size_t, C.c_int, vp],
        "emf_comm_gather_row_bands": [vp, vp, C.c_size_t, C.c_int, C.c_int, vp],
        "emf_comm_create_peer_local_group": [C.c_int, C.c_size_t, C.POINTER(vp)],
        "emf_comm_create_peer": [C.c_int, C.c_int, C.c_size_t, vp, vp, C.POINTER(vp)],
        "emf_comm_exchanges": [vp, C.POINTER(C.c_uint64)],
        "emf_comm_create_host_staged": [vp, C.POINTER(vp)],
        "emf_synth_create": [C.c_int, C.c_int, fp, C.c_int, C.c_uint64, C.c_float, C.c_float,
                             C.POINTER(vp)],
        "emf_synth_destroy": [vp],
        "emf_synth_render": [vp, C.c_int, vp, vp],
        "emf_synth_camera_pose": [vp, C.c_int, fp, fp],
        "emf_synth_sphere": [vp, C.c_int, C.c_int, fp, fp, fp],
    }
    for name, argtypes in sigs.items():
        getattr(lib, name).argtypes = argtypes
    lib._emf_sigs = sigs
    _lib = lib
    return lib


def _check(fn: str, rc: int):
    if rc != 0:
        raise FusionError(fn, rc, load().emf_fusion_last_error_string().decode(errors="replace"))


def _farr(v, n):
    a = np.ascontiguousarray(np.asarray(v, np.float32).reshape(-1))
    assert a.size == n
    return (C.c_float * n)(*a.tolist())


def default_params() -> FusionParams:
    p = FusionParams()
    load().emf_fusion_default_params(C.byref(p))
    return p


def make_params(width=640, height=480, bg_res=512, bg_voxel=0.01, obj_res=128,
                materialize_gradients=False, **overrides) -> FusionParams:
    """Reference defaults (config/default.cfg) with the BASELINE.json volume sizes."""
    p = default_params()
    p.width, p.height = width, height
    f = 525.0 * width / 640.0
    p.K[:] = [f, 0, width / 2 - 0.5, 0, f, height / 2 - 0.5, 0, 0, 1]
    p.bg_res[:] = [bg_res] * 3
    p.bg_voxel_size = bg_voxel
    p.volume_pose_t[:] = [0, 0, bg_res * bg_voxel / 2]
    p.obj_res[:] = [obj_res] * 3
    p.materialize_gradients = int(materialize_gradients)
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


class Communicator:
    """RCCL communicator (one process per GPU).  The unique id travels over torch.distributed."""

    def __init__(self, unique_id: bytes, rank: int, world: int):
        self._h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, 128)
        _check("emf_comm_create", load().emf_comm_create(buf, rank, world, C.byref(self._h)))
        self.rank, self.world = rank, world

    @classmethod
    def local_group(cls, world: int, transport: str = "host", max_bytes: int = 0):
        """`world` communicators of THIS process for `world` Fusion objects on `world` threads sharing one
        GPU (rehearsal of the multi-GPU code path).  transport "host": collectives staged through host
        memory; "peer": the direct peer-write exchanges (max_bytes = the slot size: W * H * 16 holds the raycast's fused exchange -- keys, background band, mask; with less the frame falls back to unfused exchanges)."""
        handles = (C.c_void_p * world)()
        if transport == "peer":
            _check("emf_comm_create_peer_local_group",
                   load().emf_comm_create_peer_local_group(world, int(max_bytes), handles))
        else:
            _check("emf_comm_create_local_group", load().emf_comm_create_local_group(world, handles))
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c._h, c.rank, c.world = C.c_void_p(handles[r]), r, world
            out.append(c)
        return out

    @classmethod
    def host_staged(cls, dist):
        """One process per rank, collectives staged through host memory and carried by the given
        torch.distributed module (gloo): rehearsal of the N-rank job on fewer than N GPUs."""
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()

        def view(ptr, count, dtype):
            return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dtype)), shape=(count,)))

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:  # noqa: BLE001 - reported through the C++ exception
                    print("host-staged collective failed:", repr(e), flush=True)
                    return 1
            return wrapped

        def sum_f32(user, ptr, count):
            dist.all_reduce(view(ptr, count, C.c_float), op=dist.ReduceOp.SUM)

        def min_u64(user, ptr, count):
            t = view(ptr, count, C.c_int64)
            t ^= -(2 ** 63)          # unsigned order -> signed order
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            t ^= -(2 ** 63)

        def bcast(user, ptr, nbytes, root):
            dist.broadcast(view(ptr, nbytes, C.c_uint8), src=root)

        f_red = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
        f_bc = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)

        class Callbacks(C.Structure):
            _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("sum", f_red), ("min", f_red),
                        ("bcast", f_bc), ("user", C.c_void_p)]
        c = cls.__new__(cls)
        c._keep = (f_red(guard(sum_f32)), f_red(guard(min_u64)), f_bc(guard(bcast)))
        cb = Callbacks(rank, world, c._keep[0], c._keep[1], c._keep[2], None)
        c._h = C.c_void_p()
        _check("emf_comm_create_host_staged", load().emf_comm_create_host_staged(C.byref(cb), C.byref(c._h)))
        c.rank, c.world = rank, world
        return c

    @classmethod
    def peer(cls, dist, max_bytes: int):
        """One process per rank, direct peer-write exchanges: the ranks' receive buffers are mapped into each
        other with hipIpc*, the handles travel through the given torch.distributed module (gloo)."""
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()

        def gather(user, mine, nbytes, allp):
            try:
                t = torch.from_numpy(np.ctypeslib.as_array(C.cast(mine, C.POINTER(C.c_uint8)), shape=(nbytes,)).copy())
                outs = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(outs, t)
                dst = np.ctypeslib.as_array(C.cast(allp, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
                for r, o in enumerate(outs):
                    dst[r * nbytes:(r + 1) * nbytes] = o.numpy()
                return 0
            except Exception as e:  # noqa: BLE001 - reported through the C++ exception
                print("peer bootstrap all-gather failed:", repr(e), flush=True)
                return 1
        f_ag = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
        c = cls.__new__(cls)
        c._keep = (f_ag(gather),)
        c._h = C.c_void_p()
        _check("emf_comm_create_peer", load().emf_comm_create_peer(rank, world, int(max_bytes),
                                                                   C.cast(c._keep[0], C.c_void_p), None, C.byref(c._h)))
        c.rank, c.world = rank, world
        return c

    # the four exchanges on their own (device arrays of emfusion_amd.devmem, stream = devmem.Stream or None)
    def all_reduce_sum_f32(self, arr, stream=None):
        _check("emf_comm_all_reduce_sum_f32", load().emf_comm_all_reduce_sum_f32(
            self._h, C.c_void_p(arr.ptr), arr.nbytes // 4, stream.handle if stream else None))

    def all_reduce_min_u64(self, arr, stream=None):
        _check("emf_comm_all_reduce_min_u64", load().emf_comm_all_reduce_min_u64(
            self._h, C.c_void_p(arr.ptr), arr.nbytes // 8, stream.handle if stream else None))

    def broadcast(self, arr, root, stream=None):
        _check("emf_comm_broadcast", load().emf_comm_broadcast(
            self._h, C.c_void_p(arr.ptr), arr.nbytes, int(root), stream.handle if stream else None))

    def gather_row_bands(self, arr, band_rows, stream=None):
        _check("emf_comm_gather_row_bands", load().emf_comm_gather_row_bands(
            self._h, C.c_void_p(arr.ptr), arr.pitch, int(band_rows), arr.shape[0], stream.handle if stream else None))

    def delayed(self, microseconds: int) -> "Communicator":
        """Latency model around this communicator: every exchange (a grouped one counts once) first keeps its
        stream busy for `microseconds`.  Keep `self` alive as long as the result is used."""
        c = Communicator.__new__(Communicator)
        c._h = C.c_void_p()
        _check("emf_comm_create_delayed", load().emf_comm_create_delayed(self._h, int(microseconds), C.byref(c._h)))
        c.rank, c.world, c._inner = self.rank, self.world, self
        return c

    def describe(self) -> dict:
        """What the transport itself reports about this rank: ranks it sees, device ordinal, PCI bus id, version."""
        import json as _json
        buf = C.create_string_buffer(1024)
        _check("emf_comm_describe", load().emf_comm_describe(self._h, buf, len(buf)))
        return _json.loads(buf.value.decode())

    def exchanges(self) -> int:
        """Exchanges issued so far through a delayed() communicator (0 for the others)."""
        n = C.c_uint64(0)
        _check("emf_comm_exchanges", load().emf_comm_exchanges(self._h, C.byref(n)))
        return int(n.value)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check("emf_comm_unique_id", load().emf_comm_unique_id(buf))
        return buf.raw

    def close(self):
        if self._h:
            load().emf_comm_destroy(self._h)
            self._h = C.c_void_p()


class SyntheticStream:
    """Deterministic synthetic RGB-D stream (emf::SyntheticScene)."""

    def __init__(self, width, height, K, num_spheres, seed=0xE3F5, noise=0.002, dropout=0.01):
        self._h = C.c_void_p()
        self.width, self.height, self.n = width, height, num_spheres
        _check("emf_synth_create",
               load().emf_synth_create(width, height, _farr(K, 9), num_spheres, seed, noise,
                                       dropout, C.byref(self._h)))

    def render(self, frame: int):
        depth = np.empty((self.height, self.width), np.float32)
        ids = np.empty((self.height, self.width), np.uint8)
        _check("emf_synth_render",
               load().emf_synth_render(self._h, frame, depth.ctypes.data, ids.ctypes.data))
        return depth, ids

    def camera_pose(self, frame: int):
        R = (C.c_float * 9)()
        t = (C.c_float * 3)()
        _check("emf_synth_camera_pose", load().emf_synth_camera_pose(self._h, frame, R, t))
        return np.array(R, np.float32), np.array(t, np.float32)

    def sphere(self, k: int, frame: int):
        c = (C.c_float * 3)()
        r = C.c_float()
        v = C.c_float()
        _check("emf_synth_sphere",
               load().emf_synth_sphere(self._h, k, frame, c, C.byref(r), C.byref(v)))
        return np.array(c, np.float32), float(r.value), float(v.value)

    def close(self):
        if self._h:
            load().emf_synth_destroy(self._h)
            self._h = C.c_void_p()


class Fusion:
    """One emf::EMFusion instance (background + object volumes) on the current device."""

    def __init__(self, params: FusionParams, comm: Optional[Communicator] = None):
        self._h = C.c_void_p()
        self.params = params
        self._comm = comm
        _check("emf_fusion_create",
               load().emf_fusion_create(C.byref(params), comm._h if comm else None,
                                        C.byref(self._h)))

    @classmethod
    def from_config(cls, path=None, calibration=None, comm: Optional[Communicator] = None, materialize_gradients=False):
        """The instance `apps/emfusion_synth --configfile` builds: every key of one of the reference's configuration
        files (ignore_person, LM / Huber / bilateral / lifecycle thresholds included -- FusionParams holds a subset)."""
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.params = FusionParams()
        self._comm = comm
        _check("emf_fusion_create_from_config",
               load().emf_fusion_create_from_config(os.fspath(path).encode() if path else None,
                                                    os.fspath(calibration).encode() if calibration else None,
                                                    int(materialize_gradients), comm._h if comm else None,
                                                    C.byref(self.params), C.byref(self._h)))
        return self

    def close(self):
        if self._h:
            load().emf_fusion_destroy(self._h)
            self._h = C.c_void_p()

    def reset(self):
        _check("emf_fusion_reset", load().emf_fusion_reset(self._h))

    def add_object(self, center, vol_size: float) -> int:
        out = C.c_int32()
        _check("emf_fusion_add_object",
               load().emf_fusion_add_object(self._h, _farr(center, 3), vol_size, C.byref(out)))
        return out.value

    def process_frame(self, depth_view: EmfImage, cam_R, cam_t, obj_poses=None, masks=None,
                      run_masks=False):
        """obj_poses: {id: (R9, t3)}; masks: {id: EmfImage (device u8 0/1)}."""
        obj_poses = obj_poses or {}
        masks = masks or {}
        n = len(obj_poses)
        ids = (C.c_int32 * max(n, 1))(*obj_poses.keys())
        Rs = (C.c_float * (9 * max(n, 1)))()
        ts = (C.c_float * (3 * max(n, 1)))()
        for i, (R, t) in enumerate(obj_poses.values()):
            Rs[9 * i:9 * i + 9] = np.asarray(R, np.float32).reshape(-1).tolist()
            ts[3 * i:3 * i + 3] = np.asarray(t, np.float32).reshape(-1).tolist()
        m = len(masks)
        mids = (C.c_int32 * max(m, 1))(*masks.keys())
        mviews = (EmfImage * max(m, 1))(*masks.values())
        _check("emf_fusion_process_frame",
               load().emf_fusion_process_frame(self._h, C.byref(depth_view), _farr(cam_R, 9),
                                               _farr(cam_t, 3), n, ids, Rs, ts, m, mids, mviews,
                                               int(run_masks)))

    def process_rgbd(self, depth: np.ndarray):
        """EMFusion::processFrame(const RGBD&): a host depth image in metres (uploaded, filtered, fused)."""
        d = np.ascontiguousarray(depth, np.float32)
        _check("emf_fusion_process_rgbd",
               load().emf_fusion_process_rgbd(self._h, d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[1], d.shape[0]))

    def use_preproc_masks(self, path):
        """EMFusion::usePreprocMasks: <path>/Mask%04d.plk on every mask frame of process_rgbd."""
        _check("emf_fusion_use_preproc_masks", load().emf_fusion_use_preproc_masks(self._h, os.fspath(path).encode()))

    def last_masks(self):
        """EMFusion::getLastMasks: (instances, (H, W, 3) u8 image or None before the first mask frame)."""
        n = C.c_int32(0)
        img = np.zeros((self.params.height, self.params.width, 3), np.uint8)
        _check("emf_fusion_get_last_masks", load().emf_fusion_get_last_masks(self._h, img.ctypes.data, img.nbytes, C.byref(n)))
        return n.value, img

    def set_tracking(self, camera=True, objects=True):
        """From the next frame on, track the camera / object poses instead of taking them as inputs."""
        _check("emf_fusion_set_tracking", load().emf_fusion_set_tracking(self._h, int(camera), int(objects)))

    def create_object_from_mask(self, mask_view: EmfImage) -> int:
        """EMFusion::initNewObjVolume on the current frame's points; -1 if no object is created."""
        i = C.c_int32(-1)
        _check("emf_fusion_create_object_from_mask",
               load().emf_fusion_create_object_from_mask(self._h, C.byref(mask_view), C.byref(i)))
        return i.value

    def queue_new_object_masks(self, mask_views):
        """Masks for in-frame object creation by the next process_frame (initOrMatchObjs)."""
        arr = (EmfImage * max(len(mask_views), 1))(*mask_views)
        _check("emf_fusion_queue_new_object_masks",
               load().emf_fusion_queue_new_object_masks(self._h, len(mask_views), arr))

    def queue_instance_masks(self, mask_views):
        """The instance masks of a Mask R-CNN frame for the next process_frame (initOrMatchObjs);
        the masks are modified in place by the carving step."""
        arr = (EmfImage * max(len(mask_views), 1))(*mask_views)
        _check("emf_fusion_queue_instance_masks",
               load().emf_fusion_queue_instance_masks(self._h, len(mask_views), arr))

    def last_mask_assignment(self):
        ids, n = (C.c_int32 * 64)(), C.c_int32(0)
        _check("emf_fusion_last_mask_assignment",
               load().emf_fusion_last_mask_assignment(self._h, ids, 64, C.byref(n)))
        return [ids[i] for i in range(min(n.value, 64))]

    def last_created(self):
        ids, n = (C.c_int32 * 64)(), C.c_int32(0)
        _check("emf_fusion_last_created", load().emf_fusion_last_created(self._h, ids, 64, C.byref(n)))
        return [ids[i] for i in range(min(n.value, 64))]

    def match_mask(self, mask_view: EmfImage):
        """EMFusion::matchSegmentation: (object id or -1, best IoU)."""
        i, iou = C.c_int32(-1), C.c_float(0.0)
        _check("emf_fusion_match_mask",
               load().emf_fusion_match_mask(self._h, C.byref(mask_view), C.byref(i), C.byref(iou)))
        return i.value, iou.value

    def update_object(self, obj_id: int, mask_view: EmfImage):
        """EMFusion::updateObj + ObjTSDF::resize; returns the centre shift (zeros: unchanged)."""
        off = (C.c_float * 3)()
        _check("emf_fusion_update_object",
               load().emf_fusion_update_object(self._h, int(obj_id), C.byref(mask_view), off))
        return np.array(list(off), np.float32)

    def queue_instance_scores(self, scores):
        """Class scores (n, num_classes) that go with the masks of queue_instance_masks."""
        s = np.ascontiguousarray(scores, np.float64)
        s = s.reshape(len(s), -1) if s.size else np.zeros((0, 81))
        _check("emf_fusion_queue_instance_scores",
               load().emf_fusion_queue_instance_scores(self._h, s.shape[0], s.shape[1], s.ctypes.data))

    def object_class(self, obj_id: int) -> int:
        c = C.c_int32()
        _check("emf_fusion_object_class", load().emf_fusion_object_class(self._h, int(obj_id), C.byref(c)))
        return c.value

    def object_info(self, obj_id: int) -> Dict[str, float]:
        """resolution, voxel size, truncation distance and existence probability of an object volume as it is now."""
        res = (C.c_int32 * 3)()
        vox, trunc, ex = C.c_float(), C.c_float(), C.c_float()
        _check("emf_fusion_object_info",
               load().emf_fusion_object_info(self._h, int(obj_id), res, C.byref(vox), C.byref(trunc), C.byref(ex)))
        return dict(res=tuple(res), voxel_size=vox.value, truncdist=trunc.value, existence=ex.value)

    def set_ignore_person(self, on=True):
        """Params.ignore_person: objects classified as person stay out of renderings and mesh files."""
        _check("emf_fusion_set_ignore_person", load().emf_fusion_set_ignore_person(self._h, int(on)))

    def set_depth_broadcast(self, root: int = 0):
        """Multi-GPU: every frame's depth image is broadcast from rank `root` first (-1: off)."""
        _check("emf_fusion_set_depth_broadcast", load().emf_fusion_set_depth_broadcast(self._h, int(root)))

    def render(self):
        """EMFusion::render: (image (H, W, 3) u8 RGB, colour map (256, 3) u8)."""
        rgb = np.empty((self.params.height, self.params.width, 3), np.uint8)
        cmap = np.empty((256, 3), np.uint8)
        _check("emf_fusion_render", load().emf_fusion_render(self._h, rgb.ctypes.data, cmap.ctypes.data))
        return rgb, cmap

    def mesh(self, obj_id: int = 0):
        """TSDF::getMesh / ObjTSDF::getMesh: (vertices (n, 3), normals (n, 3), triangles (m, 4))."""
        nv, nt = C.c_uint32(), C.c_uint32()
        _check("emf_fusion_extract_mesh",
               load().emf_fusion_extract_mesh(self._h, int(obj_id), C.byref(nv), C.byref(nt)))
        v = np.empty((nv.value, 3), np.float32)
        n = np.empty((nv.value, 3), np.float32)
        t = np.empty((nt.value, 4), np.int32)
        _check("emf_fusion_copy_mesh",
               load().emf_fusion_copy_mesh(self._h, v.ctypes.data, n.ctypes.data, t.ctypes.data))
        return v, n, t

    def enable_pose_log(self, on=True):
        _check("emf_fusion_enable_pose_log", load().emf_fusion_enable_pose_log(self._h, int(on)))

    def setup_output(self, exp_frame_meshes=False, exp_vols=False):
        """Reference EMFusion::setupOutput: log on; exp_vols keeps deleted objects' volumes too."""
        _check("emf_fusion_setup_output",
               load().emf_fusion_setup_output(self._h, int(exp_frame_meshes), int(exp_vols)))

    def write_results(self, directory: str, volumes: bool = True):
        """poses-*.txt, mesh_bg.ply, mesh_<id>.ply always; tsdfs/*.bin with `volumes` (reference formats)."""
        _check("emf_fusion_write_results",
               load().emf_fusion_write_results(self._h, os.fspath(directory).encode(), int(volumes)))

    def set_cleanup(self, on=True):
        """Run the reference's cleanUpObjs at the end of every frame."""
        _check("emf_fusion_set_cleanup", load().emf_fusion_set_cleanup(self._h, int(on)))

    def last_deleted(self):
        ids, n = (C.c_int32 * 64)(), C.c_int32(0)
        _check("emf_fusion_last_deleted", load().emf_fusion_last_deleted(self._h, ids, 64, C.byref(n)))
        return [ids[i] for i in range(min(n.value, 64))]

    def set_preprocess(self, on=True):
        """Filter incoming depth maps as the reference's preprocessDepth does (bilateral + patches)."""
        _check("emf_fusion_set_preprocess", load().emf_fusion_set_preprocess(self._h, int(on)))

    def pose(self, obj_id: int = 0):
        """(R 3x3, t 3): camera -> world for id 0, object volume -> world otherwise."""
        R, t = (C.c_float * 9)(), (C.c_float * 3)()
        _check("emf_fusion_get_pose", load().emf_fusion_get_pose(self._h, int(obj_id), R, t))
        return np.array(R, np.float32).reshape(3, 3), np.array(t, np.float32)

    def track_result(self, obj_id: int = 0) -> Dict[str, float]:
        it, acc, conv, err = C.c_int32(), C.c_int32(), C.c_int32(), C.c_float()
        _check("emf_fusion_track_result",
               load().emf_fusion_track_result(self._h, int(obj_id), C.byref(it), C.byref(acc),
                                              C.byref(conv), C.byref(err)))
        return dict(iterations=it.value, accepted=acc.value, converged=bool(conv.value), error=err.value)

    def stage_estep(self):
        _check("emf_fusion_stage_estep", load().emf_fusion_stage_estep(self._h))

    def stage_raycast(self):
        _check("emf_fusion_stage_raycast", load().emf_fusion_stage_raycast(self._h))

    def stage_integrate(self):
        _check("emf_fusion_stage_integrate", load().emf_fusion_stage_integrate(self._h))

    def synchronize(self):
        _check("emf_fusion_synchronize", load().emf_fusion_synchronize(self._h))

    def enable_timings(self, on=True):
        _check("emf_fusion_enable_timings", load().emf_fusion_enable_timings(self._h, int(on)))

    def last_timings(self) -> Dict[str, float]:
        t = FrameTimings()
        _check("emf_fusion_last_timings", load().emf_fusion_last_timings(self._h, C.byref(t)))
        return t.as_dict()

    def enable_raycast_stats(self, on=True):
        _check("emf_fusion_enable_raycast_stats",
               load().emf_fusion_enable_raycast_stats(self._h, int(on)))

    def raycast_stats(self):
        c = (C.c_uint64 * 4)()
        _check("emf_fusion_raycast_stats", load().emf_fusion_raycast_stats(self._h, c))
        return tuple(int(v) for v in c)

    def kernel_timers_enable(self, max_launches: int):
        _check("emf_fusion_kernel_timers_enable",
               load().emf_fusion_kernel_timers_enable(self._h, int(max_launches)))

    def kernel_timers_select(self, kinds):
        """Bracket only these kernel kinds (names from KERNEL_KINDS) with event pairs."""
        mask = 0
        for k in kinds:
            mask |= 1 << KERNEL_KINDS.index(k)
        _check("emf_fusion_kernel_timers_select", load().emf_fusion_kernel_timers_select(self._h, mask))

    def kernel_timers_stride(self, every: int):
        """Bracket only every `every`-th launch of a kind with an event pair (1 = all)."""
        _check("emf_fusion_kernel_timers_stride", load().emf_fusion_kernel_timers_stride(self._h, int(every)))

    def kernel_timers_clear(self):
        _check("emf_fusion_kernel_timers_clear", load().emf_fusion_kernel_timers_clear(self._h))

    def kernel_timers_collect(self) -> Dict[str, Dict[str, float]]:
        """Synchronises, then returns {kind: {launches, total_ms, units}} (+ '_dropped')."""
        arr = (KernelSummary * len(KERNEL_KINDS))()
        dropped = C.c_uint64()
        _check("emf_fusion_kernel_timers_collect",
               load().emf_fusion_kernel_timers_collect(self._h, arr, C.byref(dropped)))
        out = {k: dict(launches=int(arr[i].launches), total_ms=float(arr[i].total_ms),
                       units=float(arr[i].units)) for i, k in enumerate(KERNEL_KINDS)}
        out["_dropped"] = int(dropped.value)
        return out

    def image_view(self, which: str, obj_id: int = 0) -> EmfImage:
        v = EmfImage()
        _check("emf_fusion_get_image",
               load().emf_fusion_get_image(self._h, IMG[which], obj_id, C.byref(v)))
        return v

    def image(self, which: str, obj_id: int = 0) -> np.ndarray:
        """Synchronise and copy an image of the last frame to the host."""
        self.synchronize()
        v = self.image_view(which, obj_id)
        dt, ch = _IMG_DTYPE[IMG[which]]
        out = np.empty((v.height, v.width, ch) if ch > 1 else (v.height, v.width), dt)
        assert v.pitch == v.width * ch * out.itemsize
        devmem.memcpy_d2h(out, v.data)
        return out

    def volume(self, which: str, obj_id: int = 0) -> np.ndarray:
        self.synchronize()
        ptr = C.c_void_p()
        res = (C.c_int32 * 3)()
        _check("emf_fusion_get_volume",
               load().emf_fusion_get_volume(self._h, VOL[which], obj_id, C.byref(ptr), res))
        dt = np.uint8 if which in ("fgmask", "bricks") else np.float32
        out = np.empty((res[2], res[1], res[0]), dt)
        devmem.memcpy_d2h(out, ptr.value)
        return out

    def background_overlap(self) -> bool:
        return load().emf_fusion_background_overlap(self._h) == 1

    def upload_host_time(self):
        """(seconds, frames): host time process_rgbd has spent handing depth maps to the device so far."""
        s, n = C.c_double(0), C.c_uint64(0)
        _check("emf_fusion_upload_host_time", load().emf_fusion_upload_host_time(self._h, C.byref(s), C.byref(n)))
        return float(s.value), int(n.value)

    def batched_chunks(self) -> int:
        """0: per-volume path; k >= 1: batched path with k launches per stage (one per <= 32 table slots)."""
        return int(load().emf_fusion_batched_chunks(self._h))

    def object_ids(self):
        """Live objects of the job in creation order."""
        ids = (C.c_int32 * 256)()
        n = C.c_int()
        _check("emf_fusion_object_ids", load().emf_fusion_object_ids(self._h, ids, 256, C.byref(n)))
        return [ids[i] for i in range(n.value)]

    def visible_objects(self):
        ids = (C.c_int32 * 256)()
        n = C.c_int()
        _check("emf_fusion_visible_objects",
               load().emf_fusion_visible_objects(self._h, ids, 256, C.byref(n)))
        return [ids[i] for i in range(n.value)]

    def frame_index(self) -> int:
        return load().emf_fusion_frame_index(self._h)

    def owns_object(self, obj_id: int) -> bool:
        return bool(load().emf_fusion_owns_object(self._h, obj_id))


def read_depth_png(path, scale=1.0 / 5000.0) -> np.ndarray:
    """core/Readers.cpp readPngGray through the C API: float32 (H, W) = raw * scale."""
    w, h = C.c_int32(), C.c_int32()
    _check("emf_io_read_depth_png", load().emf_io_read_depth_png(os.fspath(path).encode(), scale, None, 0, C.byref(w), C.byref(h)))
    out = np.empty((h.value, w.value), np.float32)
    _check("emf_io_read_depth_png", load().emf_io_read_depth_png(os.fspath(path).encode(), scale,
                                                               out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(w), C.byref(h)))
    return out


def load_config(path=None, calibration=None):
    """The reference's configuration file (config/*.cfg, apps/EM-Fusion.cpp:268-371) and / or a Co-Fusion
    calibration.txt applied to the reference defaults (core/Config.cpp): (FusionParams for Fusion(...), {key: value
    string or list of strings} of every configurable field)."""
    prm = FusionParams()
    buf = C.create_string_buffer(1 << 14)
    _check("emf_io_load_config",
           load().emf_io_load_config(os.fspath(path).encode() if path else None,
                                     os.fspath(calibration).encode() if calibration else None,
                                     C.byref(prm), buf, len(buf)))
    fields: Dict[str, object] = {}
    for line in buf.value.decode().splitlines():
        k, v = [t.strip() for t in line.split("=", 1)]
        if k.startswith("Params.MaskRCNNParams."):
            fields.setdefault(k, []).append(v)
        else:
            fields[k] = v
    return prm, fields


def read_exr(path, channel=None) -> np.ndarray:
    """core/Readers.cpp readExr through the C API: one channel of a scan-line OpenEXR file as float32 (H, W)."""
    w, h = C.c_int32(), C.c_int32()
    ch = channel.encode() if channel else None
    _check("emf_io_read_exr", load().emf_io_read_exr(os.fspath(path).encode(), ch, None, 0, C.byref(w), C.byref(h)))
    out = np.empty((h.value, w.value), np.float32)
    _check("emf_io_read_exr", load().emf_io_read_exr(os.fspath(path).encode(), ch, out.ctypes.data_as(C.POINTER(C.c_float)),
                                                     out.size, C.byref(w), C.byref(h)))
    return out


def image_reader(base, colordir="colour", depthdir="depth"):
    """(number of frames, first index) of a Co-Fusion style dataset, as the C++ emf::ImageReader sees it."""
    n, first = C.c_int32(), C.c_int32()
    _check("emf_io_image_reader", load().emf_io_image_reader((os.fspath(base) + os.sep).encode(), colordir.encode(),
                                                            depthdir.encode(), C.byref(n), C.byref(first)))
    return n.value, first.value


def tum_associations(path):
    """[(depth file name, time stamp)] of a TUM associations.txt, as the C++ TUMRGBDReader parses it."""
    n, out = C.c_int32(), []
    _check("emf_io_tum_associations", load().emf_io_tum_associations(os.fspath(path).encode(), -1, None, 0, None, C.byref(n)))
    for i in range(n.value):
        name, stamp = C.create_string_buffer(512), C.c_double()
        _check("emf_io_tum_associations", load().emf_io_tum_associations(os.fspath(path).encode(), i, name, 512, C.byref(stamp), C.byref(n)))
        out.append((name.value.decode(), stamp.value))
    return out


def load_preproc_masks(path):
    """core/Readers.cpp loadPreprocessedMasks through the C API: (boxes (N, 4), masks (N, H, W) u8, scores (N, S))."""
    n, w, h, ns = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    f = load().emf_io_load_preproc_masks
    _check("emf_io_load_preproc_masks", f(os.fspath(path).encode(), C.byref(n), C.byref(w), C.byref(h), None, 0, None, 0, None, 0, C.byref(ns)))
    masks = np.zeros((n.value, h.value, w.value), np.uint8)
    boxes = np.zeros((n.value, 4), np.float64)
    scores = np.zeros((n.value, ns.value), np.float64)
    _check("emf_io_load_preproc_masks", f(os.fspath(path).encode(), C.byref(n), C.byref(w), C.byref(h), masks.ctypes.data, masks.nbytes,
                                          boxes.ctypes.data_as(C.POINTER(C.c_double)), boxes.size, scores.ctypes.data_as(C.POINTER(C.c_double)),
                                          scores.size, C.byref(ns)))
    return boxes, masks, scores


def trim_pool() -> int:
    """Really free the device buffers the host classes keep pooled (waits for the device); bytes freed."""
    n = C.c_uint64(0)
    _check("emf_fusion_trim_pool", load().emf_fusion_trim_pool(C.byref(n)))
    return int(n.value)


def write_volume(filename, volume: np.ndarray, voxel_size: float):
    """Reference volume dump (EMFusion::writeVolume): volume is float32 (Nz, Ny, Nx)."""
    v = np.ascontiguousarray(volume, np.float32)
    nz, ny, nx = v.shape
    _check("emf_io_write_volume",
           load().emf_io_write_volume(os.fspath(filename).encode(), v.ctypes.data_as(C.POINTER(C.c_float)),
                                      (C.c_int32 * 3)(nx, ny, nz), float(voxel_size)))


def write_mesh(filename, vertices, normals, triangles):
    """ASCII PLY of the reference (EMFusion::writeMesh)."""
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(triangles, np.int32).reshape(-1, 4)
    assert len(v) == len(n)
    _check("emf_io_write_mesh",
           load().emf_io_write_mesh(os.fspath(filename).encode(), len(v), v.ctypes.data, n.ctypes.data,
                                    len(t), t.ctypes.data))


def write_pose_file(filename, poses: Dict[int, tuple]):
    """TUM-style pose file from {frame: (R 3x3, t 3)} (EMFusion::writePoseFile)."""
    frames = sorted(poses)
    R = np.ascontiguousarray([np.asarray(poses[f][0], np.float32).reshape(9) for f in frames], np.float32)
    t = np.ascontiguousarray([np.asarray(poses[f][1], np.float32).reshape(3) for f in frames], np.float32)
    _check("emf_io_write_pose_file",
           load().emf_io_write_pose_file(os.fspath(filename).encode(), len(frames),
                                         (C.c_int32 * max(len(frames), 1))(*frames),
                                         R.ctypes.data_as(C.POINTER(C.c_float)),
                                         t.ctypes.data_as(C.POINTER(C.c_float))))
