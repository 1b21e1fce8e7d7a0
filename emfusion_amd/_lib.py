"""ctypes binding of libemf_hip.so (the emf_hip_* C ABI declared in include/emf_hip.h).

Fails loudly when the library is missing or lacks a declared symbol -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
LIB_PATH = PKG_DIR / "libemf_hip.so"
HEADER_PATH = REPO_ROOT / "include" / "emf_hip.h"

EMF_OK = 0


class EmfImage(C.Structure):
    """Mirror of emf_image_t: device pointer + byte pitch + size."""

    _fields_ = [("data", C.c_void_p), ("pitch", C.c_size_t), ("width", C.c_int32),
                ("height", C.c_int32)]


class EmfHipError(RuntimeError):
    def __init__(self, fn: str, code: int, msg: str):
        super().__init__(f"{fn} failed with {code}: {msg}")
        self.code = code


_F9 = C.POINTER(C.c_float)
_I3 = C.POINTER(C.c_int32)
_IMG = C.POINTER(EmfImage)
_FP = C.c_void_p  # device float*/u8* passed as integers
_STREAM = C.c_void_p

# name -> argtypes; every entry returns int.  Must cover every function in include/emf_hip.h.
SIGNATURES = {
    "emf_hip_abi_version": [],
    "emf_hip_device_info": [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)],
    "emf_hip_computePoints": [_IMG, _IMG, _F9, _STREAM],
    "emf_hip_updateTSDF": [_IMG, _IMG, _FP, _FP, _FP, _F9, _F9, _F9, _I3, C.c_float, C.c_float,
                           C.c_float, _IMG, _STREAM],
    "emf_hip_computeInvLambda": [_F9, _IMG, _STREAM],
    "emf_hip_computeTSDFGrads": [_FP, _FP, _I3, _STREAM],
    "emf_hip_raycastTSDF": [_FP, _FP, _FP, _FP, _FP, _IMG, _IMG, _IMG, _IMG, _F9, _F9, _F9, _I3,
                            C.c_float, C.c_float, C.c_float, _FP, _STREAM],
    "emf_hip_getVolumeVals": [_FP, C.c_int, _IMG, _F9, _F9, _I3, C.c_float, _IMG, _STREAM],
    "emf_hip_updateFgBgProbs": [_IMG, _IMG, _FP, _FP, _FP, _F9, _F9, _F9, _I3, C.c_float, _STREAM],
    "emf_hip_computeFgProbs": [_FP, _FP, _FP, _I3, _STREAM],
    "emf_hip_maskRaycastWeights": [_FP, _FP, _FP, _I3, _STREAM],
    "emf_hip_computeAssociation": [_FP, _FP, _IMG, _F9, _F9, _I3, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_float, _IMG, _STREAM],
    "emf_hip_normalizeAssociation": [_IMG, C.c_int, C.c_int, _IMG, _IMG, _STREAM],
    "emf_hip_sumAssociation": [_IMG, C.c_int, _IMG, _STREAM],
    "emf_hip_compositeRaycast": [C.c_int, _I3, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG,
                                 _IMG, _IMG, _IMG, _IMG, _IMG, C.c_int, _FP, _STREAM],
    "emf_hip_compositeVisibility": [C.c_int, _I3, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG,
                                    _IMG, _IMG, _IMG, _IMG, _IMG, C.c_int, _FP, C.c_int, _FP, _FP, _STREAM],
    "emf_hip_occludedMask": [_IMG, _IMG, C.c_int, _IMG, _STREAM],
    "emf_hip_estepBatched": [_FP, _FP, C.c_int, _IMG, C.c_int, _IMG, _IMG, _STREAM],
    "emf_hip_estepBatchedFromDepth": [_FP, _FP, C.c_int, _IMG, _F9, _IMG, C.c_int, _IMG, _IMG, _STREAM],
    "emf_hip_raycastBatched": [_FP, _FP, _I3, C.c_int, C.c_int, C.c_int, _F9, C.c_int, C.c_int,
                               C.c_int, _FP, _FP, _FP, _STREAM],
    "emf_hip_normalizeAssociationTable": [_FP, C.c_int, C.c_int, C.c_int, _FP, _STREAM],
    "emf_hip_raycastBatchedLanes": [_FP, _FP, _I3, C.c_int, C.c_int, C.c_int, _F9, C.c_int, C.c_int,
                                    C.c_int, _FP, _FP, C.c_int, _FP, _STREAM],
    "emf_hip_raycastBatchedObjects": [_FP, _FP, _I3, C.c_int, C.c_int, C.c_int, _F9, C.c_int, _FP, _FP, _FP,
                                      _STREAM],
    "emf_hip_signMapBytes": [_I3],
    "emf_hip_rebuildSignMaps": [_FP, _I3, _FP, _STREAM],
    "emf_hip_unseenTileBytes": [_I3],
    "emf_hip_rebuildUnseenTiles": [_FP, _FP, _I3, _FP, _STREAM],
    "emf_hip_raycastFarBoundBytes": [C.c_int, C.c_int, C.c_int],
    "emf_hip_raycastFarBounds": [_FP, _FP, _I3, C.c_int, C.c_int, C.c_int, _F9, C.c_uint32, _FP, _STREAM],
    "emf_hip_relevantTileBytes": [_I3],
    "emf_hip_updateRelevantTiles": [_FP, _I3, C.c_int, _STREAM],
    "emf_hip_voxelReciprocal": [C.c_float, C.POINTER(C.c_float)],
    "emf_hip_voxelReciprocalCached": [C.c_float, C.POINTER(C.c_float)],
    "emf_hip_voxelReciprocalBegin": [C.c_float, C.c_void_p, _STREAM],
    "emf_hip_voxelReciprocalEnd": [C.c_float, C.c_ulonglong, C.POINTER(C.c_float)],
    "emf_hip_voxelReciprocalExhaustive": [C.c_float, C.POINTER(C.c_ulonglong)],
    "emf_hip_spinProbe": [C.c_void_p, C.c_uint32, _STREAM],
    "emf_hip_spinDelay": [C.c_uint32, _STREAM],
    "emf_hip_l1GatherProbe": [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, _STREAM],
    "emf_hip_peerBufferBytes": [C.c_int, C.c_size_t],
    "emf_hip_peerScatter": [C.c_void_p, _FP, C.c_size_t, C.c_size_t, C.c_uint32, _STREAM],
    "emf_hip_peerSignalWait": [C.c_void_p, C.c_uint32, C.c_uint32, _STREAM],
    "emf_hip_peerReduceSumF32": [C.c_void_p, C.c_uint32, C.c_size_t, _FP, _STREAM],
    "emf_hip_peerReduceMinU64": [C.c_void_p, C.c_uint32, C.c_size_t, _FP, _STREAM],
    "emf_hip_peerCopyFromSlot": [C.c_void_p, C.c_uint32, C.c_int, C.c_size_t, _FP, C.c_size_t, _STREAM],
    "emf_hip_peerWaitReduceSumF32": [C.c_void_p, C.c_uint32, C.c_size_t, _FP, _STREAM],
    "emf_hip_peerWaitReduceMinU64": [C.c_void_p, C.c_uint32, C.c_size_t, _FP, _STREAM],
    "emf_hip_peerWaitCopyFromSlots": [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      _STREAM],
    "emf_hip_estepBatchedPeer": [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_uint32, _STREAM],
    "emf_hip_peerNormalizeAssociation": [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _STREAM],
    "emf_hip_peerRaycastSlotBytes": [C.c_int, C.c_int],
    "emf_hip_packHitKeysPeer": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                C.c_void_p, C.c_uint32, _STREAM],
    "emf_hip_compositeFromKeysPeer": [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
                                     + [C.c_void_p] * 13 + [C.c_int, C.c_void_p, _STREAM],
    "emf_hip_visibilityFlagsMirror": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _STREAM],
    "emf_hip_sweepFastPathPremises": [C.c_void_p, _STREAM],
    "emf_hip_debugPixelRounding": [_FP, _FP, C.c_int, _FP, _FP, _STREAM],
    "emf_hip_debugBandDecision": [_FP, _FP, _FP, C.c_int, C.c_float, _FP, _FP, _FP, _FP, _STREAM],
    "emf_hip_streamCopy": [_FP, _FP, C.c_size_t, _STREAM],
    "emf_hip_preprocessDepth": [_IMG, _IMG, C.c_int, C.c_float, C.c_float, _STREAM],
    "emf_hip_pointStatsScratchBytes": [],
    "emf_hip_maskedPointStats": [_IMG, _IMG, _F9, _F9, _FP, _FP, _STREAM],
    "emf_hip_maskOverlap": [_IMG, _IMG, _FP, _STREAM],
    "emf_hip_maskAssociationMassBytes": [],
    "emf_hip_maskAssociationMass": [_IMG, _IMG, _IMG, _FP, _STREAM],
    "emf_hip_carveMask": [_IMG, _IMG, C.c_int, _IMG, _FP, _STREAM],
    "emf_hip_objectExtentStats": [_IMG, _IMG, _F9, _F9, _FP, _FP, _FP, _I3, C.c_float, _FP, _FP, _STREAM],
    "emf_hip_copyValues": [_FP, _FP, C.c_int, _I3, _I3, _I3, _STREAM],
    "emf_hip_meshScratchBytes": [_I3],
    "emf_hip_hideLabel": [_IMG, C.c_int, _IMG, _IMG, _IMG, _IMG, _STREAM],
    "emf_hip_renderPhong": [_IMG, _IMG, _IMG, C.c_void_p, _F9, _IMG, _STREAM],
    "emf_hip_meshCount": [_FP, _FP, _FP, _I3, _FP, _FP, _STREAM],
    "emf_hip_meshEmit": [_FP, _FP, _FP, _FP, _I3, C.c_float, _FP, _FP, _FP, _FP, _STREAM],
    "emf_hip_trackScratchBytes": [C.c_int, C.c_int],
    "emf_hip_trackPrepare": [_FP, _FP, C.c_int, C.c_float, _STREAM],
    "emf_hip_trackIterate": [_FP, _FP, C.c_int, _IMG, C.c_void_p, _FP, C.c_size_t,
                             C.c_int, _STREAM],
    "emf_hip_trackStep": [_FP, _FP, C.c_int, _IMG, C.c_void_p, _FP, C.c_size_t,
                          C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, _STREAM],
    "emf_hip_trackWeightImages": [_FP, _FP, C.c_int, _IMG, C.c_void_p, _FP, C.c_size_t, _FP, _FP, _STREAM],
    "emf_hip_computePoseGradients": [_FP, _FP, _IMG, _F9, _F9, _I3, C.c_float, _FP, _STREAM],
    "emf_hip_integrateBatched": [_FP, _FP, _I3, C.c_int, _FP, _IMG, _IMG, _F9, C.c_int, _FP,
                                 _STREAM],
    "emf_hip_integrateCullScratchBytes": [_I3, C.c_int],
    "emf_hip_integrateBatchedCulled": [_FP, _FP, _I3, C.c_int, _FP, _IMG, _IMG, _F9, _FP, C.c_uint32, _FP, _FP,
                                       _STREAM],
    "emf_hip_integrateDirtyMapBytes": [_I3],
    "emf_hip_integrateBatchedCulledOut": [_FP, _FP, _I3, C.c_int, _FP, _IMG, _IMG, _F9, C.c_void_p, C.c_int, _FP,
                                          C.c_uint32, _FP, _FP, _STREAM],
    "emf_hip_integratePrepareOut": [C.c_void_p, _I3, C.c_int, _FP, _STREAM],
    "emf_hip_visibilityFlags": [_FP, C.c_int, C.c_int, _FP, _FP, _STREAM],
    "emf_hip_resetBrickFlags": [_FP, _I3, _STREAM],
    "emf_hip_packHitKeys": [C.c_int, _I3, _IMG, _IMG, _FP, C.c_int, C.c_int, _STREAM],
    "emf_hip_compositeFromKeys": [_FP, C.c_int, _I3, C.c_int, _I3, _IMG, _IMG, _IMG, _IMG, _IMG,
                                  _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, _IMG, C.c_int, _FP,
                                  _STREAM],
    "emf_hip_visibilityFlagsIndexed": [_FP, C.c_int, _I3, C.c_int, _FP, _STREAM],
}


class EmfVolumeOut(C.Structure):
    """Mirror of emf_volume_out_t (second copy of a double-buffered volume + its dirty maps)."""
    _fields_ = [("tsdf", C.c_void_p), ("weights", C.c_void_p), ("dirtyPrev", C.c_void_p), ("dirtyNext", C.c_void_p)]


class EmfModel(C.Structure):
    """Mirror of emf_model_t (device model table entry)."""

    _fields_ = [("tsdf", C.c_void_p), ("weights", C.c_void_p), ("grads", C.c_void_p),
                ("fgProbs", C.c_void_p), ("fgVolMask", C.c_void_p), ("brickFlags", C.c_void_p),
                ("assoc", C.c_void_p), ("raylengths", C.c_void_p), ("vertices", C.c_void_p),
                ("normals", C.c_void_p), ("hitMask", C.c_void_p), ("signMaps", C.c_void_p),
                ("relevantTiles", C.c_void_p), ("unseenTiles", C.c_void_p), ("res", C.c_int32 * 3),
                ("id", C.c_int32), ("voxelSize", C.c_float), ("truncdist", C.c_float),
                ("maxWeight", C.c_float), ("assocC1", C.c_float), ("assocC2", C.c_float),
                ("alpha", C.c_float), ("assocC3", C.c_float), ("reserved", C.c_int32),
                ("rcpVoxel", C.c_float), ("pad_", C.c_int32)]


class EmfPose(C.Structure):
    _fields_ = [("R", C.c_float * 9), ("t", C.c_float * 3)]


class EmfTrackParams(C.Structure):
    """Mirror of emf_track_params_t (reference defaults, data.h:36-43)."""

    _fields_ = [("huberThresh", C.c_float), ("maxWeight", C.c_float), ("tau", C.c_float),
                ("eps1", C.c_float), ("eps2", C.c_float), ("nuInit", C.c_float)]

    @classmethod
    def defaults(cls):
        return cls(0.2, 64.0, 1e3, 1e-8, 1e-8, 2.0)


class EmfTrackState(C.Structure):
    """Mirror of emf_track_state_t (device-resident Levenberg-Marquardt state of one model)."""

    _fields_ = [("R", C.c_float * 9), ("t", C.c_float * 3), ("Rtrial", C.c_float * 9),
                ("ttrial", C.c_float * 3), ("A", C.c_float * 36), ("b", C.c_float * 6),
                ("x", C.c_float * 6), ("mu", C.c_float), ("nu", C.c_float), ("rho", C.c_float),
                ("err", C.c_float), ("errNew", C.c_float), ("maxIwBits", C.c_uint32),
                ("maxIwTrialBits", C.c_uint32),
                ("converged", C.c_int32), ("firstIteration", C.c_int32),
                ("evaluateGradient", C.c_int32), ("haveTrial", C.c_int32),
                ("iterations", C.c_int32), ("accepted", C.c_int32), ("iwSel", C.c_int32),
                ("wSel", C.c_int32), ("needAccum", C.c_int32), ("haveSpec", C.c_int32),
                ("spec", C.c_float * 28), ("checkB", C.c_int32),
                ("pending", C.c_int32), ("body", C.c_int32), ("iterTarget", C.c_int32),
                ("logCur", C.c_float), ("logTrial", C.c_float), ("wFac", C.c_float)]

_lib = None


def declared_symbols() -> list[str]:
    """Function names declared in include/emf_hip.h (what the library must export)."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(emf_hip_\w+)\s*\(", text)))


def load_variant(suffix: str) -> C.CDLL:
    """A second build of the kernel library beside the product's, e.g. "_fma" = libemf_hip_fma.so
    (`make -C csrc fma`: a*b+c contraction on).  For the tests that measure what contraction alone
    changes; nothing in the product loads it."""
    path = LIB_PATH.with_name(LIB_PATH.stem + suffix + LIB_PATH.suffix)
    if not path.exists():
        raise RuntimeError(f"{path} is missing: build it with `make -C {PKG_DIR / 'csrc'} fma`")
    return _bind(C.CDLL(os.fspath(path)))


def load() -> C.CDLL:
    """Load libemf_hip.so and bind every declared entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP library has not been built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C {PKG_DIR / 'csrc'}`."
            " There is no CPU fallback.")
    _lib = _bind(C.CDLL(os.fspath(LIB_PATH)))
    return _lib


def _bind(lib: C.CDLL) -> C.CDLL:
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.emf_hip_trackScratchBytes.restype = C.c_size_t
    lib.emf_hip_unseenTileBytes.restype = C.c_size_t
    lib.emf_hip_pointStatsScratchBytes.restype = C.c_size_t
    lib.emf_hip_meshScratchBytes.restype = C.c_size_t
    lib.emf_hip_integrateCullScratchBytes.restype = C.c_size_t
    lib.emf_hip_integrateDirtyMapBytes.restype = C.c_size_t
    lib.emf_hip_signMapBytes.restype = C.c_size_t
    lib.emf_hip_raycastFarBoundBytes.restype = C.c_size_t
    lib.emf_hip_relevantTileBytes.restype = C.c_size_t
    lib.emf_hip_peerBufferBytes.restype = C.c_size_t
    lib.emf_hip_peerRaycastSlotBytes.restype = C.c_size_t
    lib.emf_hip_maskAssociationMassBytes.restype = C.c_size_t
    lib.emf_hip_last_error_string.argtypes = []
    lib.emf_hip_last_error_string.restype = C.c_char_p
    return lib


def check(fn_name: str, rc: int) -> None:
    if rc != EMF_OK:
        msg = load().emf_hip_last_error_string().decode(errors="replace")
        raise EmfHipError(fn_name, rc, msg)
