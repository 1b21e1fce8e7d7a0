import ctypes as C, time, sys
import numpy as np
sys.path.insert(0, '.')
from scipy.ndimage import binary_dilation
from emfusion_amd import _lib, devmem, pipeline
from emfusion_amd.ops import image_view
from tests.parity_util import to_dev
lib = _lib.load()
devmem.set_device(0)
Wf, Hf = 320, 240
prm = pipeline.make_params(Wf, Hf, 128, 0.04, 32, visibility_thresh=100, boundary=10, mask_frames=1)
synth = pipeline.SyntheticStream(Wf, Hf, np.array(prm.K, np.float32), 2, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
probe, word = devmem.Stream(non_blocking=True), devmem.HostWord()
disc = np.hypot(*np.mgrid[-9:10, -9:10]) <= 9.0
keep, centres = [], {}
for f in range(12):
    T = [time.perf_counter()]
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
    devmem.synchronize()
    T.append(time.perf_counter())
    word.set(0)
    lib.emf_hip_spinProbe(word.ptr, C.c_uint32(4000), probe.handle)
    T.append(time.perf_counter())
    fus.process_frame(image_view(d), R, t, poses, {}, True)
    T.append(time.perf_counter())
    busy = probe.busy()
    word.set(1)
    probe.synchronize()
    T.append(time.perf_counter())
    fus.synchronize()
    T.append(time.perf_counter())
    created = fus.last_created()
    for i in fus.object_ids():
        centres[i] = fus.pose(i)[1]
    sizes = {i: fus.volume("tsdf", i).shape[0] for i in fus.object_ids()}
    T.append(time.perf_counter())
    print(f, "busy", busy, "created", created, "sizes", sizes, "assign", fus.last_mask_assignment(), " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip(T, T[1:])), flush=True)
