"""Direction of the tracked camera's drift on the bench scene (wall z = 2.2 + 0.15x - 0.1y, floor y = 1:
sliding along their intersection line ~(1, 0, 0.15) is unobservable for the background tracker)."""
import sys, numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from emfusion_amd import ops, pipeline
from emfusion_amd.devmem import DeviceArray
W, H = 640, 480
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, 4, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
fus.set_tracking(True, False)  # camera tracked, objects supplied
line = np.array([1, 0, 0.15]); line /= np.linalg.norm(line)
for f in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    depth, sid = synth.render(f); R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
    rm = f % prm.mask_frames == 0
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if rm else {}
    d = DeviceArray.from_numpy(depth)
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, rm)
    fus.synchronize()
    if f % 10 == 9:
        Rc, tc = fus.pose(0)
        e = (tc - t).astype(np.float64)
        along = float(e @ line)
        perp = float(np.linalg.norm(e - along * line))
        dR = Rc.astype(np.float64) @ R.reshape(3, 3).astype(np.float64).T
        ang = np.degrees(np.arccos(min(1, max(-1, (np.trace(dR) - 1) / 2))))
        print("frame %3d: error %6.1f mm = %6.1f mm along the wall/floor line + %5.1f mm across; rotation error %.3f deg" % (f, 1e3 * np.linalg.norm(e), 1e3 * along, 1e3 * perp, ang))
