"""Tracking on the bench workload: per-stage time, iterations, camera error.  python scripts/track_timing.py [frames]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from emfusion_amd import ops, pipeline
from emfusion_amd.devmem import DeviceArray
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 25
W, H = 640, 480
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, 4, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
fus.set_tracking(True, True)
fus.kernel_timers_enable(4000)
rows = []
for f in range(frames):
    depth, sid = synth.render(f); R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
    rm = f % prm.mask_frames == 0
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if rm else {}
    d = DeviceArray.from_numpy(depth)
    t0 = time.perf_counter()
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, rm)
    fus.synchronize()
    dt = time.perf_counter() - t0
    rows.append((f, 1e3 * np.linalg.norm(fus.pose(0)[1] - t), 1e3 * dt,
                 {i: (r["iterations"], r["accepted"], int(r["converged"])) for i in [0] + ids
                  for r in [fus.track_result(i)]} if f else None))
for e in rows[::4]:
    print("frame %d cam err %.2f mm  wall %.2f ms  %s" % e)
print("mean wall per frame (frames 1..): %.2f ms" % np.mean([r[2] for r in rows[1:]]))
k = fus.kernel_timers_collect()
print({n: (v["launches"], round(v["total_ms"] / max(v["launches"], 1), 3)) for n, v in k.items() if not n.startswith("_") and v["launches"]})
