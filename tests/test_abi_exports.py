"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/emf_hip.h declares, and rejects bad arguments without touching a device."""
import ctypes as C
import subprocess

import pytest

from emfusion_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _lib.declared_symbols()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in emf_hip.h but not exported"
    # and the python binding types every one of them
    untyped = [n for n in declared if n not in _lib.SIGNATURES and n != "emf_hip_last_error_string"]
    assert not untyped, f"no ctypes signature for {untyped}"


def test_dynamic_symbol_table_matches_header():
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], check=True,
                         capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert set(_lib.declared_symbols()) <= exported


def test_abi_version():
    import re
    declared = int(re.search(r"#define\s+EMF_HIP_ABI_VERSION\s+(\d+)", _lib.HEADER_PATH.read_text()).group(1))
    assert declared == 8  # bump together with the struct mirrors in emfusion_amd/_lib.py
    assert _lib.load().emf_hip_abi_version() == declared


def test_null_and_shape_arguments_are_rejected_before_any_launch():
    lib = _lib.load()
    res = (C.c_int32 * 3)(8, 8, 8)
    assert lib.emf_hip_computeTSDFGrads(None, None, res, None) == -1  # EMF_E_NULL
    assert b"tsdf is NULL" in lib.emf_hip_last_error_string()
    bad = (C.c_int32 * 3)(8, 1, 8)
    assert lib.emf_hip_computeTSDFGrads(C.c_void_p(16), C.c_void_p(16), bad, None) == -2  # SHAPE
    img = _lib.EmfImage(C.c_void_p(256), 4, 8, 8)  # pitch smaller than a row
    K = (C.c_float * 9)(*([0.0] * 9))
    assert lib.emf_hip_computePoints(C.byref(img), C.byref(img), K, None) == -3  # EMF_E_PITCH
    ok = _lib.EmfImage(C.c_void_p(256), 32, 8, 8)
    assert lib.emf_hip_getVolumeVals(C.c_void_p(16), 4, C.byref(ok), K, K, res, 0.01, C.byref(ok),
                                     None) == -4  # EMF_E_ARG: channels
    assert lib.emf_hip_normalizeAssociation(C.byref(ok), 0, 0, None, None, None) == -5  # EMF_E_LIMIT


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libemf_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_struct_mirrors_have_the_sizes_the_library_asserts():
    """emf_model_t / emf_track_state_t are mirrored by ctypes structures; csrc/tracking.hip holds the
    matching static_asserts (168 and 496 bytes)."""
    import ctypes as C
    assert C.sizeof(_lib.EmfModel) == 168
    assert C.sizeof(_lib.EmfTrackState) == 496
    assert C.sizeof(_lib.EmfVolumeOut) == 32
