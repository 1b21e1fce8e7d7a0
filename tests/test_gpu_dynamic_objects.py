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
    """30 frames: object 1 is spawned from its mask; from frame 3 on its instance mask is reported too
    generously (dilated onto the wall behind it), still matches, and the percentile box of the matched
    points outgrows the volume: updateObj -> ObjTSDF::resize inside the frame; object 2 appears at frame
    8.  Every process_frame() returns while the probe wave is still spinning."""
    from scipy.ndimage import binary_dilation
    from emfusion_amd import _lib, devmem, pipeline
    from emfusion_amd.ops import image_view
    lib = _lib.load()
    Wf, Hf = 320, 240
    prm = pipeline.make_params(Wf, Hf, 128, 0.04, 32, visibility_thresh=100, boundary=10, mask_frames=1)
    synth = pipeline.SyntheticStream(Wf, Hf, np.array(prm.K, np.float32), 2, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    probe, word = devmem.Stream(non_blocking=True), devmem.HostWord()
    disc = np.hypot(*np.mgrid[-9:10, -9:10]) <= 9.0
    keep, centres, created, res_history, stalled = [], {}, [], [], []
    for f in range(30):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        d = to_dev(depth)
        m1 = sid == 1
        if f >= 3:
            m1 = binary_dilation(m1, disc) & (sid != 2)
        inst = [to_dev(m1.astype(np.uint8))]
        if f >= 8:
            inst.append(to_dev((sid == 2).astype(np.uint8)))
        keep += [d, inst]
        fus.queue_instance_masks([image_view(m) for m in inst])
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), c) for i, c in centres.items()}
        devmem.synchronize()  # uploads above are the harness's, not the frame's
        _spin(lib, probe, word)
        fus.process_frame(image_view(d), R, t, poses, {}, True)
        if not probe.busy():
            stalled.append(f)
        word.set(1)
        probe.synchronize()
        fus.synchronize()
        created += [i for i in fus.last_created() if i > 0]
        for i in fus.object_ids():
            centres[i] = fus.pose(i)[1]  # a resize moves the volume's centre
        res_history.append({i: fus.volume("tsdf", i).shape[0] for i in fus.object_ids()})
    assert not stalled, f"process_frame() synchronised the device in frames {stalled}"
    assert created[:2] == [1, 2] and len(created) <= 3, created
    sizes1 = [r[1] for r in res_history]
    assert sizes1[0] == 32 and max(sizes1) > 32, sizes1   # grown inside a frame
    assert res_history[-1][2] >= 32 and sorted(fus.visible_objects()) == [1, 2]
    assert (fus.volume("weights", 1) > 0).sum() > 500 and (fus.volume("weights", 2) > 0).sum() > 500
    fus.close()
    synth.close()
