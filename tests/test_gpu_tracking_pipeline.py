"""The whole per-frame schedule with tracking switched on (reference EMFusion::processFrame with
performTracking, EMFusion.cpp:70-129, 672-724): camera tracked against the background, objects
against the camera, E-steps in between -- HIP classes against the frame-level oracle, and both
against the ground-truth poses of the synthetic stream."""
import numpy as np
import pytest

from tests.oracle_pipeline import Affine32, OraclePipeline
from tests.parity_util import to_dev

pytestmark = pytest.mark.gpu

W, H = 160, 120
BG_RES, BG_VOX, OBJ_RES = 64, 0.04, 32
NOBJ, NFRAMES, MASK_EVERY, ITERS = 2, 5, 3, 100


@pytest.fixture(scope="module")
def run(oracle, dev):
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    prm = pipeline.make_params(W, H, BG_RES, BG_VOX, OBJ_RES, visibility_thresh=100, boundary=5,
                               mask_frames=MASK_EVERY)
    assert prm.max_tracking_iter == ITERS
    K = np.array(prm.K, np.float32)
    synth = pipeline.SyntheticStream(W, H, K, NOBJ, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    orc = OraclePipeline(oracle, W, H, K, BG_RES, BG_VOX, list(prm.volume_pose_t), OBJ_RES,
                         visibility_thresh=100, boundary=5)
    ids = []
    for k in range(NOBJ):
        c, r, vs = synth.sphere(k, 0)
        ids.append(fus.add_object(c, vs))
        orc.add_object(c, vs)
    fus.set_tracking(camera=True, objects=True)
    rows = []
    for f in range(NFRAMES):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)  # ground truth: used for frame 0 only
        truth = {i: synth.sphere(i - 1, f)[0] for i in ids}
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), truth[i]) for i in ids}
        run_masks = f % MASK_EVERY == 0
        masks = {i: (sid == i).astype(np.uint8) for i in ids} if run_masks else {}
        d_depth = to_dev(depth)
        d_masks = {i: to_dev(m) for i, m in masks.items()}
        fus.process_frame(image_view(d_depth), R, t, poses, {i: image_view(m) for i, m in d_masks.items()},
                          run_masks)
        fus.synchronize()
        orc.process_frame(depth, Affine32(R.reshape(3, 3), t),
                          {i: Affine32(p[0].reshape(3, 3), p[1]) for i, p in poses.items()}, masks,
                          run_masks, track_camera=True, track_objects=True, track_iters=ITERS)
        rows.append(dict(f=f, cam_true=(R.reshape(3, 3), t), obj_true=truth,
                         cam=fus.pose(0), ocam=(orc.pose.R.copy(), orc.pose.t.copy()),
                         obj={i: fus.pose(i) for i in ids},
                         oobj={v["id"]: (v["pose"].R.copy(), v["pose"].t.copy()) for v in orc.objects},
                         res={i: fus.track_result(i) for i in [0] + ids} if f else {}))
    yield fus, orc, ids, rows
    fus.close()
    synth.close()


def test_frame0_takes_the_supplied_poses(run):
    _, _, ids, rows = run
    r0 = rows[0]
    assert np.allclose(r0["cam"][0], r0["cam_true"][0]) and np.allclose(r0["cam"][1], r0["cam_true"][1])
    assert not r0["res"]


def test_camera_pose_tracks_like_the_oracle_and_the_truth(run):
    _, _, _, rows = run
    for r in rows[1:]:
        assert r["res"][0]["accepted"] >= 3, r["res"][0]
        # same objective, same start, same iteration budget: agreement at the tolerance
        assert np.abs(r["cam"][0] - r["ocam"][0]).max() < 2e-4, ("rotation vs oracle", r["f"])
        assert np.abs(r["cam"][1] - r["ocam"][1]).max() < 2e-4, ("translation vs oracle", r["f"])
        # and it is a tracker: within a fraction of a 4 cm voxel of the true camera position
        assert np.linalg.norm(r["cam"][1] - r["cam_true"][1]) < 0.02, r["f"]


def test_object_poses_track_like_the_oracle(run):
    _, _, ids, rows = run
    for r in rows[1:]:
        for i in ids:
            assert np.abs(r["obj"][i][1] - r["oobj"][i][1]).max() < 5e-4, (r["f"], i)
            # the objects are spheres: their rotation is unobservable and the LM drifts along those
            # directions on rounding noise alone -- not compared; it must stay a rotation, though
            Ro = r["obj"][i][0].astype(np.float64)
            assert np.abs(Ro @ Ro.T - np.eye(3)).max() < 1e-4, (r["f"], i)
            assert np.linalg.norm(r["obj"][i][1] - r["obj_true"][i]) < 0.03, (r["f"], i)
            assert r["res"][i]["accepted"] >= 1


def test_volumes_stay_in_parity_under_tracking(run):
    from tests.parity_util import assert_parity
    fus, orc, _, _ = run
    # poses differ by ~1e-4 between the two sides, so voxels at surface discontinuities may flip:
    # tolerance of the pipeline test with a slightly larger outlier budget
    assert_parity(fus.volume("weights", 0), orc.bg["wts"], "bg weights", rtol=1e-3, atol=1e-4, budget=1e-2)
    assert_parity(fus.volume("tsdf", 0), orc.bg["tsdf"], "bg tsdf", rtol=1e-3, atol=1e-3, budget=2e-2)
