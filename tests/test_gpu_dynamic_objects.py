"""Objects are created, matched, grown and re-centred INSIDE frames (reference EMFusion.cpp:329-372,
495-560, 827-863: initOrMatchObjs -> initNewObjVolume / updateObj -> ObjTSDF::resize, between the raycast
and the integration of processFrame).  None of that may stall the device: no hipDeviceSynchronize, no
hipFree (which synchronises implicitly), no blocking reciprocal check (emf_hip_voxelReciprocal used to
sweep 2^32 inputs on the null stream from every TSDF constructor).

The probe: one wave spinning on a stream of its own (emf_hip_spinProbe) while process_frame() runs.  A
host call that waits for the whole device cannot return before that wave ends, so if the probe's stream
is still busy when process_frame() returns, the frame contained no device-wide synchronisation."""
import numpy as np
import pytest

from tests.parity_util import to_dev

pytestmark = pytest.mark.gpu


def _spin(lib, stream, word, ms=4000):
    import ctypes as C
    word.set(0)
    rc = lib.emf_hip_spinProbe(word.ptr, C.c_uint32(ms), stream.handle)
    assert rc == 0, rc


def test_reciprocal_verdicts_are_cached_and_can_be_had_without_waiting(dev):
    import ctypes as C
    from emfusion_amd import _lib, devmem
    lib = _lib.load()
    size = C.c_float(0.0123456)  # a size nothing else in the suite uses
    r = C.c_float(-1)
    assert lib.emf_hip_voxelReciprocalCached(size, C.byref(r)) == -7 and r.value == 0  # EMF_E_NOTREADY
    probe, word = devmem.Stream(non_blocking=True), devmem.HostWord()
    check = devmem.Stream(non_blocking=True)
    counter = devmem.DeviceArray.full((2,), 7, np.uint64)  # Begin must reset it
    _spin(lib, probe, word)
    assert lib.emf_hip_voxelReciprocalBegin(size, C.c_void_p(counter.ptr), check.handle) == 0
    check.synchronize()  # waits for the check alone ...
    assert probe.busy()  # ... not for the device
    word.set(1)
    probe.synchronize()
    bad = int(counter.numpy()[0])
    assert lib.emf_hip_voxelReciprocalEnd(size, C.c_ulonglong(bad), C.byref(r)) == 0
    assert (r.value == np.float32(1) / np.float32(size.value)) == (bad == 0)
    cached = C.c_float(-1)
    assert lib.emf_hip_voxelReciprocalCached(size, C.byref(cached)) == 0 and cached.value == r.value
    # the blocking form agrees, and answers a known size without touching the device
    _spin(lib, probe, word)
    again = C.c_float(-1)
    assert lib.emf_hip_voxelReciprocal(size, C.byref(again)) == 0 and again.value == r.value
    fresh = C.c_float(-1)
    assert lib.emf_hip_voxelReciprocal(C.c_float(0.0234567), C.byref(fresh)) == 0  # a new size: runs the check ...
    assert probe.busy()                                                             # ... on its own stream
    word.set(1)
    probe.synchronize()


def test_spawning_and_resizing_objects_never_synchronises_the_device(dev):
    """30 frames (tests/dynamic_probe.py, in a process of its own so that every stream has a hardware queue
    to itself): object 1 is spawned from its mask; from frame 3 on its instance mask is reported too
    generously (dilated onto the wall behind it), still matches, and the percentile box of the matched
    points outgrows the volume: updateObj -> ObjTSDF::resize inside the frame; object 2 appears at frame
    8.  Every process_frame() returns within milliseconds while the probe's wave -- resident for three
    seconds unless released -- is still spinning: nothing in the frame waited for the device."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    run = subprocess.run([sys.executable, str(root / "tests" / "dynamic_probe.py")], cwd=root, env=env,
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    line = [ln for ln in run.stdout.splitlines() if ln.startswith("PROBE_RESULT ")][-1]
    out = json.loads(line[len("PROBE_RESULT "):])
    frames = out["frames"]
    assert len(frames) == 30
    stalled = [fr for fr in frames if not fr["probe_resident"] or fr["host_ms"] > 0.25 * out["probe_ms"]]
    assert not stalled, f"process_frame() waited for the device: {stalled}"
    # frames that spawn or resize allocate and clear volumes; none of them takes anywhere near a probe period
    assert max(fr["host_ms"] for fr in frames) < 200, frames
    assert out["created"][:2] == [1, 2] and len(out["created"]) <= 3, out["created"]
    assert out["sizes1"][0] == 32 and max(out["sizes1"]) > 32, out["sizes1"]  # grown inside a frame
    assert out["visible"][:2] == [1, 2] and all(v > 500 for v in out["seen"].values()), out
