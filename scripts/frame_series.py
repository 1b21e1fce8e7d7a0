"""ms per frame in chunks of 10 frames over a bench-like run (configs[1]; masks on frames 0, 30, 60, ... unless
MASKS=first): shows whether a run is slow from the start or switches to a slow mode at some frame.
Usage (GPU box): [MASKS=first] python scripts/frame_series.py [frames]"""
import os
import sys
import time
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
inp = []
for f in range(frames):
    depth, sid = synth.render(f)
    R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
    run_masks = (f % prm.mask_frames == 0) if os.environ.get("MASKS") != "first" else f == 0
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if run_masks else {}
    inp.append((DeviceArray.from_numpy(depth), R, t, poses, masks, run_masks))
out = []
for c in range(0, frames, 10):
    fus.synchronize()
    t0 = time.perf_counter()
    for f in range(c, min(c + 10, frames)):
        d, R, t, poses, masks, rm = inp[f]
        fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, rm)
    fus.synchronize()
    out.append((time.perf_counter() - t0) * 100)
print("SERIES masks=%s " % os.environ.get("MASKS", "every30") + " ".join("%.3f" % v for v in out))
fus.close(); synth.close()
