"""What the background's tiles (32 x 8 x 8 voxels) hold after N frames of the bench scene: how many are untouched
(all weights 0), saturated free space (all tsdf 1, weight at the cap), or mixed -- what a per-tile state map
could let the integration skip.  python scripts/tile_census.py [frames]"""
import sys, numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from emfusion_amd import ops, pipeline
from emfusion_amd.devmem import DeviceArray
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 70
W, H = 640, 480
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, 4, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
def step(f):
    depth, sid = synth.render(f); R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
    rm = f % prm.mask_frames == 0
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if rm else {}
    d = DeviceArray.from_numpy(depth)
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, rm)
    fus.synchronize()
for f in range(frames):
    step(f)
def tiles(a):  # (512,512,512) z,y,x -> (64,64,16, 8,8,32)
    return a.reshape(64, 8, 64, 8, 16, 32).transpose(0, 2, 4, 1, 3, 5).reshape(64, 64, 16, -1)
w0, t0 = tiles(fus.volume("weights")), tiles(fus.volume("tsdf"))
step(frames)
w1, t1 = tiles(fus.volume("weights")), tiles(fus.volume("tsdf"))
wz = (w0 == 0).all(-1)
sat = ((w0 == 64) & (t0 == 1)).all(-1)
chg_t, chg_w = (t0 != t1).any(-1), (w0 != w1).any(-1)
n = wz.size
print("tiles %d: all-weights-zero %d (%.1f%%), saturated free %d (%.1f%%), other %d" % (n, wz.sum(), 100 * wz.mean(), sat.sum(), 100 * sat.mean(), n - wz.sum() - sat.sum()))
print("changed by frame %d: tsdf tiles %d, weight tiles %d" % (frames, chg_t.sum(), chg_w.sum()))
print("  of the tsdf-changed tiles: all-weights-zero before %d, saturated %d" % ((chg_t & wz).sum(), (chg_t & sat).sum()))
print("  all-weights-zero tiles whose weights changed %d" % (chg_w & wz).sum())
