"""Plain batched integration vs the two-level (box cull + listed tiles) launch on the bench geometry:
bg 512^3 + 4 objects 128^3, 640x480, frames of the synthetic stream.  python scripts/integrate_cull_timing.py"""
import sys, numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from emfusion_amd import ops, pipeline
from emfusion_amd.devmem import DeviceArray, Event, synchronize
from tests.scenes import Pose, rel_OC
W, H = 640, 480
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, 4, seed=0xE3F5)
dummy = DeviceArray.zeros((H, W, 3), np.float32)
dummy8 = DeviceArray.zeros((H, W), np.uint8)
def make(n, vox, pose, mid):
    t, w, a = DeviceArray.zeros((n, n, n)), DeviceArray.zeros((n, n, n)), DeviceArray.full((H, W), 1.0)
    return dict(t=t, w=w, a=a, res=(n, n, n), vox=vox, pose=pose, m=ops.make_model(t, w, a, a, dummy, dummy, dummy8, vox, 10 * vox, 64.0, 0.02, 0.8, 1.0, model_id=mid))
def scene():
    ms = [make(512, 0.01, Pose(t=list(prm.volume_pose_t)), 0)]
    for k in range(4):
        c, r, vs = synth.sphere(k, 0)
        ms.append(make(128, float(np.float32(vs) / np.float32(128)), Pose(t=[float(v) for v in c]), k + 1))
    return ms
A, B = scene(), scene()
tabA, tabB = ops.upload_models([m["m"] for m in A]), ops.upload_models([m["m"] for m in B])
res = [m["res"] for m in A]
il = DeviceArray.zeros((H, W)); ops.compute_inv_lambda(K, il)
surv = DeviceArray.zeros((1,), np.uint32)
scratch = None
def timed(fn, reps=5):
    fn(); synchronize()
    a, b = Event(), Event(); a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_ms(b) / reps
for f in (0, 10, 40, 80):
    depth, _ = synth.render(f); R, t = synth.camera_pose(f)
    cam = Pose(R.reshape(3, 3).astype(np.float64), t.astype(np.float64))
    poses = [(rel_OC(cam, m["pose"]).R32, rel_OC(cam, m["pose"]).t32) for m in A]
    d = DeviceArray.from_numpy(depth)
    tp = timed(lambda: ops.integrate_batched(tabA, poses, res, None, d, K, None, inv_lambda=il))
    scratch = ops.integrate_batched_culled(tabB, poses, res, None, d, K, 0, surv, None, inv_lambda=il, scratch=scratch)
    synchronize(); n = int(surv.numpy()[0])
    tall = timed(lambda: ops.integrate_batched_culled(tabB, poses, res, None, d, K, 0, surv, None, inv_lambda=il, scratch=scratch))
    tex = timed(lambda: ops.integrate_batched_culled(tabB, poses, res, None, d, K, n, surv, None, inv_lambda=il, scratch=scratch))
    test = timed(lambda: ops.integrate_batched_culled(tabB, poses, res, None, d, K, int(n * 1.1) + 8, surv, None, inv_lambda=il, scratch=scratch))
    tsh = timed(lambda: ops.integrate_batched_culled(tabB, poses, res, None, d, K, int(n * 0.9), surv, None, inv_lambda=il, scratch=scratch))
    print(f"frame {f}: survivors {n} boxes of {sum(-(-r[0] // 32) * -(-r[1] // 16) * -(-r[2] // 16) for r in res)}; plain {tp:.3f} ms | culled: grid=all {tall:.3f}, exact {tex:.3f}, +10% {test:.3f}, -10% {tsh:.3f} ms")
# (the two sets see different numbers of launches here; bit-identity of the two launches is
# tests/test_gpu_batched.py::test_integrate_culled_launch_equals_the_plain_batched_one)
