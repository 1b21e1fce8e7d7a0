import ctypes as C, time, sys
import numpy as np
sys.path.insert(0, '.')
from emfusion_amd import _lib, devmem, pipeline
from emfusion_amd.ops import image_view
from tests.parity_util import to_dev, dev_full, to_np
from tests.test_gpu_fast_paths import _pixels
f32 = np.float32
lib = _lib.load()
devmem.set_device(0)
sp = np.array([0.0, -0.0, 1e-45, 1e-38, 1.17549435e-38, 3e38, np.inf, -np.inf, np.nan, 0.5, 1.5, 2.5, -0.5, 2 ** 20,
               2 ** 20 + 0.5, 2 ** 23, 2 ** 31, -2 ** 31, 1e30], f32)
a, b = np.meshgrid(sp, sp)
num, den = a.reshape(-1).copy(), np.abs(b.reshape(-1)).copy()
fast, exact = _pixels(lib, num, den)
for i in np.flatnonzero(fast != exact):
    print("mismatch num=%r den=%r fast=%d exact=%d" % (num[i], den[i], fast[i], exact[i]))
rng = np.random.default_rng(9)
n = 1 << 22
num = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 6, n)).astype(f32)
den = (10 ** rng.uniform(-3, 3, n)).astype(f32)
fast, exact = _pixels(lib, num, den)
print("random mismatches", int((fast != exact).sum()))

# timing of the probe
probe, word = devmem.Stream(non_blocking=True), devmem.HostWord()
for k in range(3):
    word.set(0)
    t0 = time.perf_counter()
    lib.emf_hip_spinProbe(word.ptr, C.c_uint32(3000), probe.handle)
    t1 = time.perf_counter()
    busy = probe.busy()
    word.set(1)
    probe.synchronize()
    t2 = time.perf_counter()
    print("spin launch %.3f ms, busy=%s, release->sync %.3f ms" % ((t1 - t0) * 1e3, busy, (t2 - t1) * 1e3))
Wf, Hf = 320, 240
prm = pipeline.make_params(Wf, Hf, 128, 0.04, 32, visibility_thresh=100, boundary=10, mask_frames=1)
synth = pipeline.SyntheticStream(Wf, Hf, np.array(prm.K, np.float32), 2, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
keep = []
for f in range(4):
    depth, sid = synth.render(f)
    R, t = synth.camera_pose(f)
    d = to_dev(depth)
    inst = [to_dev((sid == 1).astype(np.uint8))]
    keep += [d, inst]
    fus.queue_instance_masks([image_view(m) for m in inst])
    devmem.synchronize()
    word.set(0)
    lib.emf_hip_spinProbe(word.ptr, C.c_uint32(3000), probe.handle)
    t0 = time.perf_counter()
    fus.process_frame(image_view(d), R, t, {}, {}, True)
    t1 = time.perf_counter()
    busy = probe.busy()
    word.set(1)
    probe.synchronize()
    t2 = time.perf_counter()
    fus.synchronize()
    t3 = time.perf_counter()
    print("frame %d: process_frame %.1f ms busy=%s release %.1f ms sync %.1f ms" % (f, (t1 - t0) * 1e3, busy, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
