"""Scenario of tests/test_gpu_dynamic_objects.py, run in a process of its own.

Why its own process: HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4).  The
probe below keeps one wave spinning on a stream for seconds; with four hardware queues one of the frame's
streams shares the probe's queue and simply queues BEHIND the spinning wave -- which looks like a
device-wide synchronisation and is not one.  With GPU_MAX_HW_QUEUES=16 (set by the test before HIP starts)
every stream of the scenario has a queue to itself.

Prints one JSON object: per frame, how long process_frame() took on the host and whether the probe's wave
was still resident when it returned; what was created; the resolution history of object 1."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

PROBE_MS = 3000


def main():
    from scipy.ndimage import binary_dilation

    from emfusion_amd import _lib, devmem, pipeline
    from emfusion_amd.devmem import DeviceArray
    from emfusion_amd.ops import image_view
    lib = _lib.load()
    devmem.set_device(0)
    Wf, Hf = 320, 240
    prm = pipeline.make_params(Wf, Hf, 128, 0.04, 32, visibility_thresh=100, boundary=10, mask_frames=1)
    synth = pipeline.SyntheticStream(Wf, Hf, np.array(prm.K, np.float32), 2, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    probe, word = devmem.Stream(non_blocking=True), devmem.HostWord()
    disc = np.hypot(*np.mgrid[-9:10, -9:10]) <= 9.0
    keep, centres = [], {}
    out = dict(frames=[], created=[], sizes1=[], probe_ms=PROBE_MS)
    for f in range(30):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        d = DeviceArray.from_numpy(depth)
        m1 = sid == 1
        if f >= 3:  # reported too generously: onto the wall behind the sphere
            m1 = binary_dilation(m1, disc) & (sid != 2)
        inst = [DeviceArray.from_numpy(m1.astype(np.uint8))]
        if f >= 8:
            inst.append(DeviceArray.from_numpy((sid == 2).astype(np.uint8)))
        keep += [d, inst]
        fus.queue_instance_masks([image_view(m) for m in inst])
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), c) for i, c in centres.items()}
        devmem.synchronize()  # the uploads above are the harness's, not the frame's
        word.set(0)
        assert lib.emf_hip_spinProbe(word.ptr, C.c_uint32(PROBE_MS), probe.handle) == 0
        t0 = time.perf_counter()
        fus.process_frame(image_view(d), R, t, poses, {}, True)
        host_ms = (time.perf_counter() - t0) * 1e3
        resident = probe.busy()
        word.set(1)
        probe.synchronize()
        fus.synchronize()
        out["frames"].append(dict(frame=f, host_ms=round(host_ms, 2), probe_resident=bool(resident)))
        out["created"] += [i for i in fus.last_created() if i > 0]
        for i in fus.object_ids():
            centres[i] = fus.pose(i)[1]  # a resize moves the volume's centre
        out["sizes1"].append(int(fus.volume("tsdf", 1).shape[0]))
    out["visible"] = sorted(fus.visible_objects())
    out["seen"] = {str(i): int((fus.volume("weights", i) > 0).sum()) for i in fus.object_ids()}
    fus.close()
    synth.close()
    print("PROBE_RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
