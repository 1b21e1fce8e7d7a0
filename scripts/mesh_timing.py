"""Marching cubes on the bench workload (background 512^3 + 4 objects 128^3 after a few frames):
device time of count+scan and emit (HIP events), algorithmic bytes, end-to-end getMesh time.
python scripts/mesh_timing.py [frames]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from emfusion_amd import ops, pipeline
from emfusion_amd.devmem import DeviceArray, Event, synchronize
from emfusion_amd._lib import load as load_hip
import ctypes as C
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H = 640, 480
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, 4, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
for f in range(frames):
    depth, sid = synth.render(f); R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
    rm = f % prm.mask_frames == 0
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if rm else {}
    d = DeviceArray.from_numpy(depth)
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, rm)
fus.synchronize()
L = ops._L
for name, mid, vox in [("background 512^3", 0, 0.01), ("object 128^3", ids[0], None)]:
    tsdf = DeviceArray.from_numpy(fus.volume("tsdf", mid)); wts = DeviceArray.from_numpy(fus.volume("weights", mid))
    fg = None if mid == 0 else DeviceArray.from_numpy(fus.volume("fgmask", mid))
    if vox is None:
        vox = float(np.float32(synth.sphere(0, 0)[2]) / np.float32(128))
    res = ops._res(tsdf)
    nvox = int(np.prod(tsdf.shape))
    scratch = DeviceArray.zeros((max(int(L.emf_hip_meshScratchBytes(res)) // 4, 2),), np.uint32)
    counts = DeviceArray.zeros((2,), np.uint32)
    def count():
        ops.check("meshCount", L.emf_hip_meshCount(ops._ptr(tsdf), ops._ptr(wts), ops._ptr(fg), res, ops._ptr(scratch), ops._ptr(counts), None))
    count(); synchronize()
    nv, nt = (int(v) for v in counts.numpy())
    verts = DeviceArray.zeros((max(nv, 1), 3)); norms = DeviceArray.zeros((max(nv, 1), 3)); tris = DeviceArray.zeros((max(nt, 1), 4), np.int32)
    def emit():
        ops.check("meshEmit", L.emf_hip_meshEmit(ops._ptr(tsdf), None, ops._ptr(wts), ops._ptr(fg), res, vox, ops._ptr(scratch), ops._ptr(verts), ops._ptr(norms), ops._ptr(tris), None))
    def timed(fn, reps=10):
        fn(); synchronize()
        a, b = Event(), Event()
        a.record()
        for _ in range(reps): fn()
        b.record(); b.synchronize()
        return a.elapsed_ms(b) / reps
    tc, te = timed(count), timed(emit)
    per_vox = 8 + (1 if fg is not None else 0)
    out_bytes = nv * 24 + nt * 16
    print(f"{name}: {nv} vertices, {nt} triangles; count+scan {tc:.3f} ms ({nvox * per_vox / tc / 1e6:.0f} GB/s of {per_vox} B/voxel), "
          f"emit {te:.3f} ms ({(nvox * per_vox + out_bytes) / te / 1e6:.0f} GB/s incl. {out_bytes / 1e6:.1f} MB out)")
    t0 = time.perf_counter(); v, n, t = fus.mesh(mid); dt = time.perf_counter() - t0
    print(f"  getMesh end to end (count, read-back, emit, download {out_bytes / 1e6:.1f} MB): {1e3 * dt:.1f} ms; "
          f"reference-style per-cube buffers would add {9 * nvox / 1e6:.0f} MB")
