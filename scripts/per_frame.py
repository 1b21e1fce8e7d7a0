"""Per-frame stage times (HIP events between stages) and march samples on the bench scene.
Usage (GPU box): python scripts/per_frame.py [frames]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from emfusion_amd import ops, pipeline
from emfusion_amd.devmem import DeviceArray

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 130
W, H = 640, 480
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, 4, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
fus.enable_timings(True)
fus.enable_raycast_stats(True)
keep, rows, last = [], [], np.zeros(4, np.uint64)
for f in range(frames):
    depth, sid = synth.render(f)
    R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
    run_masks = f % prm.mask_frames == 0
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if run_masks else {}
    d = DeviceArray.from_numpy(depth)
    keep = [(d, masks)]
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, run_masks)
    fus.synchronize()
    tm = fus.last_timings()
    st = np.array(fus.raycast_stats(), np.uint64)
    rows.append((f, tm["raycast"], tm["integrate"], tm["estep"], tm["total"], int(st[0] - last[0]), int(st[1] - last[1])))
    last = st
print("frame raycast_ms integrate_ms estep_ms total_ms samples hits")
for r in rows:
    if r[0] % 5 == 0 or r[1] > 0.8:
        print("%4d %.3f %.3f %.3f %.3f %d %d" % r)
a = np.array([r[1] for r in rows[30:]])
print("raycast over frames 30..: mean %.3f min %.3f max %.3f p50 %.3f" % (a.mean(), a.min(), a.max(), np.median(a)))
fus.close(); synth.close()
