import sys, numpy as np
sys.path.insert(0, ".")
import torch
from emfusion_amd import ops, pipeline
from emfusion_amd.devmem import DeviceArray
W, H = 640, 480
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, 4, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
fus.set_tracking(True, True)
cam, obj, cacc, oacc, cconv = [], [], [], [], []
for f in range(60):
    depth, sid = synth.render(f); R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
    rm = f % prm.mask_frames == 0
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if rm else {}
    d = DeviceArray.from_numpy(depth)
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, rm)
    fus.synchronize()
    if f:
        r0 = fus.track_result(0); ro = [fus.track_result(i) for i in ids]
        cam.append(r0["iterations"]); cacc.append(r0["accepted"]); cconv.append(int(r0["converged"]))
        obj.append(max(r["iterations"] for r in ro)); oacc.append([(r["iterations"], r["accepted"], int(r["converged"])) for r in ro])
print("cam iterations", cam); print("cam accepted  ", cacc); print("cam converged ", cconv)
print("obj max iterations", obj)
print("obj (iterations, accepted, converged) per object, last 10 frames:")
for row in oacc[-10:]: print("  ", row)
print("camera: %d iterations, %d accepted (%.0f %% rejected)" % (sum(cam), sum(cacc), 100.0 * (1 - sum(cacc) / max(1, sum(cam)))))
ti = sum(r[0] for row in oacc for r in row); ta = sum(r[1] for row in oacc for r in row)
print("objects: %d iterations, %d accepted (%.0f %% rejected)" % (ti, ta, 100.0 * (1 - ta / max(1, ti))))
