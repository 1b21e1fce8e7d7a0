"""Tracking: the device-resident weighted Levenberg-Marquardt ICP against
the oracle restatement of TSDF.cpp:170-344 (tests/oracle_tracking.py) on the same inputs.

Per-pixel quantities are bit-exact (pose gradients).  The Hessian sums are formed in a different,
fixed order than the oracle's double accumulation and the SE(3) / solver arithmetic is restated on
both sides, so LM states are compared within float-rounding tolerances, and the converged poses
within the north-star 1e-4.
"""
import ctypes as C

import numpy as np
import pytest

from tests.oracle_tracking import OracleTracker, orthonormalise
from tests.parity_util import assert_parity, dev_full, to_dev, to_np
from tests.scenes import Pose, camera_path, intrinsics, rel_CO, rel_OC, render_depth, rot

pytestmark = pytest.mark.gpu

W, H = 160, 120
K = intrinsics(W, H)
SPHERES = [((0.25, 0.05, 1.3), 0.22), ((-0.3, -0.1, 1.6), 0.18)]
BG = dict(n=(64, 64, 64), vox=0.04, pose=Pose(t=[0, 0, 1.28]))
OBJ = dict(n=(32, 32, 32), vox=0.02, pose=Pose(t=SPHERES[0][0]))


@pytest.fixture(scope="module")
def ops(dev):
    from emfusion_amd import ops as _ops
    return _ops


def _integrate(oracle, vol, frames):
    n = vol["n"]
    tsdf, wts = np.zeros((n[2], n[1], n[0]), np.float32), np.zeros((n[2], n[1], n[0]), np.float32)
    for i in frames:
        cam = camera_path(i)
        depth, _ = render_depth(W, H, K, cam, SPHERES, noise=0.002, dropout=0.01, seed=100 + i)
        oc = rel_OC(cam, vol["pose"])
        oracle.update_tsdf(depth, np.ones((H, W), np.float32), tsdf, wts, oc.R32, oc.t32, K, vol["vox"],
                           10 * vol["vox"], 64.0)
    return tsdf, wts


@pytest.fixture(scope="module")
def world(oracle):
    vols = []
    for v in (BG, OBJ):
        tsdf, wts = _integrate(oracle, v, range(4))
        vols.append(dict(v, tsdf=tsdf, wts=wts))
    cam = camera_path(5)
    depth, _ = render_depth(W, H, K, cam, SPHERES, noise=0.002, dropout=0.01, seed=105)
    points = oracle.compute_points(depth, K)
    rng = np.random.default_rng(3)
    assoc = [np.ones((H, W), np.float32), rng.uniform(0.2, 1.0, (H, W)).astype(np.float32)]
    # the tracker starts from a pose that is off by ~1.5 cm and ~0.6 degrees
    guess = cam * Pose(rot([0.2, 1.0, 0.3], 0.6), [0.012, -0.006, 0.008])
    return dict(vols=vols, cam=cam, guess=guess, points=points, assoc=assoc)


def _models(ops, world, which, grad_volumes=None):
    keep, entries = [], []
    for k in which:
        v = world["vols"][k]
        d = dict(tsdf=to_dev(v["tsdf"]), wts=to_dev(v["wts"]), assoc=to_dev(world["assoc"][k]),
                 grads=None if grad_volumes is None else to_dev(grad_volumes[k]),
                 ray=dev_full((H, W), 0.0), vert=dev_full((H, W, 3), 0.0), nrm=dev_full((H, W, 3), 0.0),
                 hit=dev_full((H, W), 0, np.uint8))
        keep.append(d)
        entries.append(ops.make_model(d["tsdf"], d["wts"], d["assoc"], d["ray"], d["vert"], d["nrm"],
                                      d["hit"], float(np.float32(v["vox"])), float(np.float32(10 * v["vox"])),
                                      64.0, 0.02, 0.8, 1.0, model_id=k, grads=d["grads"]))
    return ops.upload_models(entries), keep


def _start_pose(world, k):
    co = rel_CO(world["guess"], world["vols"][k]["pose"])
    return orthonormalise(co.R32.reshape(3, 3)).reshape(-1), co.t32


class DeviceTracker:
    def __init__(self, ops, world, which, grad_volumes=None):
        from emfusion_amd import _lib
        self.ops, self.n = ops, len(which)
        self.table, self.keep = _models(ops, world, which, grad_volumes)
        self.states = dev_full((self.n * C.sizeof(_lib.EmfTrackState),), 0, np.uint8)
        self.per_model = ops.track_scratch_bytes(W, H)
        self.scratch = dev_full((self.n * self.per_model,), 0, np.uint8)
        self.points = to_dev(world["points"])
        self.params = _lib.EmfTrackParams.defaults()
        ops.track_prepare(self.states, [_start_pose(world, k) for k in which])

    def iterate(self, iterations=1):
        self.ops.track_iterate(self.table, self.states, self.n, self.points, self.params, self.scratch,
                               self.per_model, iterations)
        return self.ops.read_track_states(self.states, self.n)


def _oracle_tracker(oracle, world, k):
    v = world["vols"][k]
    t = OracleTracker(oracle, v["tsdf"], v["wts"], v["vox"])
    R, tt = _start_pose(world, k)
    t.prepare(R, tt)
    return t


@pytest.mark.parametrize("use_grad_volume", [False, True])
def test_pose_gradients_bit_exact(oracle, ops, dev, world, use_grad_volume):
    v = world["vols"][0]
    R, t = _start_pose(world, 0)
    grads = oracle.compute_tsdf_grads(v["tsdf"]) if use_grad_volume else None
    want = oracle.compute_pose_gradients(v["tsdf"], grads, world["points"], R, t, v["vox"])
    out = dev_full((H * W, 6), 9.0)
    ops.compute_pose_gradients(to_dev(v["tsdf"]), None if grads is None else to_dev(grads),
                               to_dev(world["points"]), R, t, float(np.float32(v["vox"])), out)
    assert (np.abs(want).sum(1) > 0).sum() > 5000
    assert_parity(to_np(out), want, "pose gradients", exact=True)


@pytest.mark.parametrize("k", [0, 1], ids=["background", "object"])
def test_first_iteration_matches_oracle(oracle, ops, dev, world, k):
    dt = DeviceTracker(ops, world, [k])
    st = dt.iterate(1)[0]
    ot = _oracle_tracker(oracle, world, k)
    ot.iterate(world["points"], world["assoc"][k])
    h = ot.history[0]
    A, b = np.array(st.A, np.float32).reshape(6, 6), np.array(st.b, np.float32)
    assert np.abs(h["A"]).max() > 1.0
    assert np.abs(A - h["A"]).max() <= 2e-5 * np.abs(h["A"]).max(), "Hessian"
    assert np.abs(b - h["b"]).max() <= 2e-5 * max(np.abs(h["b"]).max(), 1e-3), "gradient"
    assert abs(st.err - h["err"]) <= 1e-5 * h["err"], "error at the current pose"
    x = np.array(st.x, np.float32)
    assert np.abs(x - h["x"]).max() <= 1e-4 * np.abs(h["x"]).max(), "LM step"
    assert abs(st.errNew - h["err_new"]) <= 1e-5 * h["err_new"], "error at the trial pose"
    assert (st.rho > 0) == (h["rho"] > 0) and abs(st.rho - h["rho"]) <= 1e-2 * abs(h["rho"]) + 1e-3
    assert st.iterations == 1 and st.accepted == ot.accepted
    assert np.allclose(np.array(st.R, np.float32).reshape(3, 3), ot.R, atol=1e-6)
    assert np.allclose(np.array(st.t, np.float32), ot.t, atol=1e-6)
    assert abs(st.mu - float(ot.mu)) <= 1e-4 * float(ot.mu)


def _pose_error(R, t, world, k):
    """Distance of a rel_pose_CO estimate from the true one (rotation angle in rad, metres)."""
    true = rel_CO(world["cam"], world["vols"][k]["pose"])
    dR = np.asarray(R, np.float64).reshape(3, 3) @ true.R.T
    ang = np.arccos(min(1.0, max(-1.0, (np.trace(dR) - 1) / 2)))
    return ang, float(np.linalg.norm(np.asarray(t, np.float64) - true.t))


def test_tracking_converges_like_the_oracle(oracle, ops, dev, world):
    iters = 100  # params.maxTrackingIter: the damping starts at 1e3 * max diag(A) and shrinks slowly
    dt = DeviceTracker(ops, world, [0])
    st = dt.iterate(iters)[0]
    ot = _oracle_tracker(oracle, world, 0)
    for _ in range(iters):
        ot.iterate(world["points"], world["assoc"][0])
    a0, d0 = _pose_error(*_start_pose(world, 0), world, 0)
    a1, d1 = _pose_error(st.R, st.t, world, 0)
    a2, d2 = _pose_error(ot.R, ot.t, world, 0)
    assert st.accepted >= 5 and ot.accepted >= 5
    assert d1 < 0.5 * d0 and a1 < 0.5 * a0, (a0, d0, a1, d1)       # it tracks
    # both minimise the same objective from the same start: the poses agree to the tolerance
    assert np.abs(np.array(st.R, np.float32).reshape(3, 3) - ot.R).max() < 1e-4, (a1, d1, a2, d2)
    assert np.abs(np.array(st.t, np.float32) - ot.t).max() < 1e-4, (a1, d1, a2, d2)


def test_models_in_lockstep_equal_single_runs(ops, dev, world):
    both = DeviceTracker(ops, world, [0, 1]).iterate(12)
    for k in (0, 1):
        one = DeviceTracker(ops, world, [k]).iterate(12)[0]
        for name in ("R", "t", "A", "b", "x"):
            assert list(getattr(both[k], name)) == list(getattr(one, name)), (k, name)  # deterministic
        assert (both[k].mu, both[k].accepted, both[k].iterations) == (one.mu, one.accepted, one.iterations)


def test_converged_models_do_nothing(ops, dev, world):
    from emfusion_amd import _lib
    dt = DeviceTracker(ops, world, [0])
    dt.params = _lib.EmfTrackParams(0.2, 64.0, 1e3, 1e30, 1e-8, 2.0)  # eps1 huge: converged at once
    st = dt.iterate(3)[0]
    assert st.converged == 1 and st.iterations == 0 and st.accepted == 0
    assert list(st.R) == list(dt.iterate(1)[0].R)


LEGACY = ("R", "t", "Rtrial", "ttrial", "A", "b", "x", "mu", "nu", "rho", "err", "errNew", "maxIwBits",
          "converged", "firstIteration", "evaluateGradient", "haveTrial", "iterations", "accepted", "iwSel")


def _fields(st):
    return {k: (list(v) if hasattr(v, "__len__") else v) for k in LEGACY for v in [getattr(st, k)]}


def test_split_calls_equal_one_call(ops, dev, world):
    """The state carries everything across trackIterate calls (pending sums, buffer parity): 3 + 1 + 6
    iterations leave the state 10 in one call leave, bit for bit."""
    one = DeviceTracker(ops, world, [0, 1]).iterate(10)
    dt = DeviceTracker(ops, world, [0, 1])
    dt.iterate(3)
    dt.iterate(1)
    dt.iterate(0)  # nothing enqueued
    parts = dt.iterate(6)
    for k in (0, 1):
        assert one[k].iterations == 10 and parts[k].iterations == 10
        assert _fields(one[k]) == _fields(parts[k]), k


def _ramped(world):
    """The same world with integration weights that grow along x: their maximum over the image
    changes with the pose, so the speculative Hessian sums of every accepted step are normalised
    with the wrong maximum and have to be re-made (emf_hip_trackIterate's extra launch)."""
    w2 = dict(world)
    vols = []
    for v in world["vols"]:
        nx = v["wts"].shape[2]
        ramp = (1.0 + np.arange(nx, dtype=np.float32) / nx)[None, None, :]
        vols.append(dict(v, wts=(v["wts"] * ramp).astype(np.float32)))
    w2["vols"] = vols
    return w2


@pytest.mark.parametrize("rescale", [1, 0], ids=["sums rescaled", "sums made anew"])
def test_speculation_miss_stays_in_parity(oracle, ops, dev, world, monkeypatch, rescale):
    """Where the weight maximum moves with the pose, the Hessian sums a trial launch makes ahead carry the wrong
    normaliser once the step is accepted.  They are rescaled by the ratio of the two maxima (no launch; EMF_TRACK_RESCALE=0:
    made anew at the accepted pose by one launch more): both stay with the oracle."""
    monkeypatch.setenv("EMF_TRACK_RESCALE", str(rescale))
    w2 = _ramped(world)
    iters = 12
    dt = DeviceTracker(ops, w2, [0])
    st = dt.iterate(iters)[0]
    calls = 1
    if rescale:
        assert st.iterations == iters and st.wFac != 1.0  # (the last accepted step's weights wait with their factor)
    else:  # every accepted step costs a launch more than the call provides for: fewer iterations per call
        assert 0 < st.iterations < iters
    while st.iterations < iters and not st.converged:
        st = dt.iterate(iters - st.iterations)[0]
        calls += 1
        assert calls < 40
    assert st.iterations == iters and st.haveTrial == 0
    ot = _oracle_tracker(oracle, w2, 0)
    for _ in range(iters):
        ot.iterate(w2["points"], w2["assoc"][0])
    assert st.accepted == ot.accepted and ot.accepted >= 3
    assert np.abs(np.array(st.R, np.float32).reshape(3, 3) - ot.R).max() < 1e-5
    assert np.abs(np.array(st.t, np.float32) - ot.t).max() < 1e-5
    assert abs(st.mu - float(ot.mu)) <= 1e-3 * float(ot.mu)
    # and with weights whose maximum does not move (the common case) one call is enough
    assert DeviceTracker(ops, world, [0]).iterate(iters)[0].iterations == iters


def test_large_image_many_blocks(oracle, ops, dev, monkeypatch):
    """1280 x 960: 1011 rows of 1216 pixels -- more than a launch has workgroups (two models: eight rows per workgroup,
    two passes of four; one model: one pass of four) and more rows of partial sums than one batch of the prologue's
    loads (2 batches)."""
    import sys
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "W", 1280)
    monkeypatch.setattr(mod, "H", 960)
    monkeypatch.setattr(mod, "K", intrinsics(1280, 960))
    big = world.__wrapped__(oracle)
    dt = DeviceTracker(ops, big, [0, 1])
    sts = dt.iterate(3)
    for k in (0, 1):
        ot = _oracle_tracker(oracle, big, k)
        for _ in range(3):
            ot.iterate(big["points"], big["assoc"][k])
        st, h = sts[k], ot.history[-1]
        assert st.iterations == 3 and st.accepted == ot.accepted
        A = np.array(st.A, np.float32).reshape(6, 6)
        assert np.abs(A - h["A"]).max() <= 2e-5 * np.abs(h["A"]).max(), "Hessian of the third iteration"
        assert abs(st.errNew - h["err_new"]) <= 2e-5 * h["err_new"]
        assert np.allclose(np.array(st.R, np.float32).reshape(3, 3), ot.R, atol=2e-6)
        assert np.allclose(np.array(st.t, np.float32), ot.t, atol=2e-6)
        one = DeviceTracker(ops, big, [k]).iterate(3)[0]  # another grid (blocks per workgroup), same sums
        assert _fields(one) == _fields(st)


def test_launch_by_launch_with_progress_words(ops, dev, world):
    """emf_hip_trackStep: the same states as emf_hip_trackIterate, and the progress words say what the
    states say (here in device memory and read back; EMFusion::trackModels points them at pinned host
    memory and reads them while the stream runs)."""
    ref = DeviceTracker(ops, world, [0, 1]).iterate(100)
    dt = DeviceTracker(ops, world, [0, 1])
    from emfusion_amd import _lib
    watch = dev_full((3,), 0, np.uint32)
    final = dev_full((2 * C.sizeof(_lib.EmfTrackState),), 0, np.uint8)  # where a done model's state is sent ahead of its word
    launch = 0
    while True:
        for _ in range(8):
            ops.track_step(dt.table, dt.states, dt.n, dt.points, dt.params, dt.scratch, dt.per_model, launch, 100,
                           watch.ptr, launch + 1, final_states=final.ptr)
            launch += 1
        w = to_np(watch)
        assert w[0] == launch
        if w[1] and w[2]:
            break
        assert launch < 220
    sts = ops.read_track_states(dt.states, 2)  # (an even number of launches: the state is in the caller's array)
    sent = ops.read_track_states(final, 2)
    for k in (0, 1):
        assert _fields(sts[k]) == _fields(ref[k])
        assert _fields(sent[k]) == _fields(ref[k])
        assert w[1 + k] == (1 if sts[k].converged else 2)
    assert launch < 100  # both converge long before the iteration budget


def test_gradient_volume_gives_the_same_states(oracle, ops, dev, world):
    """With the reference's materialised gradient volume (TSDF.cu:429-464) in the model table the LM
    loop blends stored differences instead of taking them on the fly: the same values, the same states."""
    grads = {k: oracle.compute_tsdf_grads(world["vols"][k]["tsdf"]) for k in (0, 1)}
    a = DeviceTracker(ops, world, [0, 1]).iterate(15)
    b = DeviceTracker(ops, world, [0, 1], grad_volumes=grads).iterate(15)
    for k in (0, 1):
        assert a[k].iterations == 15 and _fields(a[k]) == _fields(b[k]), k


def test_frame_without_valid_depth_converges_at_once(oracle, ops, dev, world):
    """No valid point: every wave of the first pass is skipped as dead, A = b = 0, max |b| < eps1 --
    converged with the pose untouched, as the oracle (TSDF.cpp:276-278)."""
    empty = dict(world, points=np.zeros_like(world["points"]))
    dt = DeviceTracker(ops, empty, [0, 1])
    sts = dt.iterate(4)
    for k in (0, 1):
        ot = _oracle_tracker(oracle, empty, k)
        ot.iterate(empty["points"], empty["assoc"][k])
        assert ot.converged and ot.iterations == 0
        st = sts[k]
        assert st.converged == 1 and st.iterations == 0 and st.haveTrial == 0
        R, t = _start_pose(world, k)
        assert list(st.R) == [float(x) for x in np.asarray(R, np.float32).reshape(-1)]
        assert list(st.t) == [float(x) for x in np.asarray(t, np.float32)]
        assert list(st.b) == [0.0] * 6 and st.err == 0.0


@pytest.mark.parametrize("k", [0, 1], ids=["background", "object"])
def test_weight_images_of_a_finished_stage(oracle, ops, dev, world, k):
    """emf_hip_trackWeightImages: what TSDF::getHuberWeights / getTrackingWeights download for the debug output
    (reference TSDF.cpp:346-354; the images are made at TSDF.cpp:222-256) -- here evaluated at the pose the stage
    ended with, against the oracle's per-pixel chain at that pose."""
    dt = DeviceTracker(ops, world, [k])
    st = dt.iterate(12)[0]
    assert st.accepted >= 2
    huber, track = dev_full((1, H, W), -1.0), dev_full((1, H, W), -1.0)
    ops.track_weight_images(dt.table, dt.states, 1, dt.points, dt.params, dt.scratch, dt.per_model, huber, track)
    v = world["vols"][k]
    R, t = np.array(st.R, np.float32), np.array(st.t, np.float32)
    vals = oracle.get_volume_vals(v["tsdf"], world["points"], R, t, v["vox"])
    raw = oracle.get_volume_vals(v["wts"], world["points"], R, t, v["vox"])
    tw, comb = oracle.tracking_weights(vals, raw, world["assoc"][k], 0.2, 64.0)
    assert (tw.reshape(-1) > 0).sum() > 1000 and (comb.reshape(-1) > 0).sum() > 1000
    assert_parity(to_np(huber)[0], tw.reshape(H, W), "Huber weights", exact=True)
    assert_parity(to_np(track)[0], comb.reshape(H, W), "combined tracking weights", exact=True)
    # either image alone
    only = dev_full((1, H, W), -1.0)
    ops.track_weight_images(dt.table, dt.states, 1, dt.points, dt.params, dt.scratch, dt.per_model, None, only)
    assert np.array_equal(to_np(only), to_np(track))
