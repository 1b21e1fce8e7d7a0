"""Diagnostic: how much of the exactly-1.0 free space of the warmed-up bench background lies in
uniform blocks (block size B) whose whole neighbourhood out to Chebyshev distance D is uniform too?
Weighted by voxels inside the current camera frustum (proxy for march samples)."""
import sys, ctypes as C
from pathlib import Path
import numpy as np
from scipy import ndimage
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from emfusion_amd import devmem, ops, pipeline
from emfusion_amd.devmem import DeviceArray, DeviceView

W, H, NOBJ, WARM = 640, 480, 4, 60
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, NOBJ)
fus = pipeline.Fusion(prm)
ids = [fus.add_object(synth.sphere(k, 0)[0], synth.sphere(k, 0)[2]) for k in range(NOBJ)]
keep = []
for f in range(WARM):
    depth, sid = synth.render(f)
    R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32), synth.sphere(i - 1, f)[0]) for i in ids}
    d = DeviceArray.from_numpy(depth); keep.append(d)
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if f % 30 == 0 else {}
    keep.append(masks)
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, f % 30 == 0)
t = fus.volume("tsdf", 0)
one = (t == 1.0)
print("tsdf==1 voxels:", int(one.sum()))
for B in (2, 4, 8):
    n = 512 // B
    blk = one.reshape(n, B, n, B, n, B).all((1, 3, 5))
    print(f"B={B}: voxels in all-one blocks: {blk.sum() * B**3 / one.sum():.3f} of the ==1 voxels")
    cur = blk
    for D in (1, 2, 3, 4, 6, 8):
        er = ndimage.minimum_filter(blk.astype(np.uint8), size=2 * D + 1, mode="constant", cval=1).astype(bool)
        print(f"   D>={D}: {er.sum() * B**3 / one.sum():.3f}")
for i in ids[:2]:
    to = fus.volume("tsdf", i)
    for val, name in ((0.0, "zero"), (1.0, "one")):
        m = (to == val)
        for B in (4, 8):
            n = 128 // B
            blk = m.reshape(n, B, n, B, n, B).all((1, 3, 5))
            er = ndimage.minimum_filter(blk.astype(np.uint8), size=3, mode="constant", cval=1).astype(bool)
            print(f"obj {i} {name}: frac voxels {m.mean():.3f}; B={B} blocks {blk.mean():.3f} D>=1 {er.mean():.3f}")
