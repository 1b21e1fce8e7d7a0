"""Closed-loop divergence of the tracked pipeline, HIP classes against the frame-level oracle against the truth, as a test
(until round 4 a notebook script, scripts/track_oracle_divergence.sh, on a scene that left one camera
direction unobservable).  12 tracked frames of the observable scene of tests/tum_scene.py at 320 x 240 (background 256^3 at
2 cm, a 64^3 person volume spawned from the mask of frame 0; at 160 x 120 / 4 cm / 32^3 the reference's tracker itself
loses this scene -- 11 cm of camera error after 11 frames on both sides, the person volume deleted at frame 7 -- and a
diverging tracker is no basis for a bound; at 320 x 240 / 2 cm and the full-size test's speed -- up to 17 mm of camera
motion per frame, nearly a voxel -- both sides still drift 3-4 cm in 11 frames and come 7 mm apart: the scene runs at half
speed here, `TIME_SCALE`): camera and object LM-ICP of up to 100 iterations per stage
(reference EMFusion.cpp:672-724, TSDF.cpp:170-344).

Three statements, in the order of how much feedback they contain:
  * per stage, from IDENTICAL state (a scratch oracle restarted from the HIP run's volumes and poses of the frame
    before): the poses agree to 1e-5 -- what differs is the order in which 76 800 residuals are summed;
  * in closed loop the two trajectories separate (an accept / reject verdict of the damped LM hangs on those sums, and the
    frame after starts from the other pose): bounded, and both stay the same distance from the truth;
  * neither is closer to the truth than the other: |HIP - truth| <= 1.1 |oracle - truth| + 0.5 mm in every frame."""
import json
from pathlib import Path

import numpy as np
import pytest

from tests import tum_scene as S

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
W, H, FRAMES = 320, 240, 12
CHECK_FRAMES = (2, 5, 8, 11)
TIME_SCALE = 0.5  # the scene at half speed: <= 9 mm of camera motion per frame against 2 cm voxels


@pytest.fixture(scope="module")
def runs(oracle, dev, tmp_path_factory):
    from emfusion_amd import pipeline
    from tests.tum_runner import run_closed_loop
    S.TIME_SCALE = TIME_SCALE
    try:
        staged = S.stage(tmp_path_factory.mktemp("tum_small"), frames=FRAMES, size=(W, H))
    finally:
        S.TIME_SCALE = 1.0
    prm = pipeline.make_params(W, H, 256, 0.02, 64, visibility_thresh=400, boundary=10, mask_frames=30)
    fx, fy, cx, cy = S.intrinsics(W, H)
    assert np.allclose(np.array(prm.K, np.float32).reshape(3, 3), [[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    fus = pipeline.Fusion(prm, None)
    result = run_closed_loop(fus, oracle, staged, FRAMES, CHECK_FRAMES)
    yield result
    fus.close()


def test_per_stage_poses_agree_from_identical_state(runs):
    assert [r["frame"] for r in runs["stage_cmp"]] == list(CHECK_FRAMES)
    # measured: camera 4e-9 ... 4.5e-6, the person's volume 6e-8 ... 1.3e-5 (its 64^3 voxels are 4.7 cm: a coarser model)
    for r in runs["stage_cmp"]:
        assert r["cam_R"] < 1e-5 and r["cam_t"] < 1e-5, r
        for o in r["objects"].values():
            assert o["t"] < 3e-5 and o["R"] < 3e-5, r


def test_closed_loop_separation_is_bounded_and_neither_side_is_closer_to_the_truth(runs):
    truth = runs["truth"]
    rows = []
    for f in range(1, FRAMES):
        (Rh, th), (Ro, to), (Rt, tt) = runs["hip"][f], runs["oracle"][f], truth[f]
        rows.append(dict(frame=f, sep_t=float(np.linalg.norm(th - to)), sep_R=float(np.abs(Rh - Ro).max()),
                         hip_err=float(np.linalg.norm(th - tt)), oracle_err=float(np.linalg.norm(to - tt)),
                         steps=runs["cam_tracks"][f]["iterations"]))
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "tracking_divergence_report.json").write_text(json.dumps(dict(rows=rows, per_stage=runs["stage_cmp"]), indent=1))
    print(json.dumps(rows))
    for r in rows:
        assert r["sep_t"] < 5e-4 and r["sep_R"] < 2e-4, r          # closed loop (measured: <= 47 um, <= 1.3e-5)
        assert r["hip_err"] < 0.025 and r["oracle_err"] < 0.025, r  # both are trackers: about a voxel from the truth (measured: 9-20 mm)
        assert r["hip_err"] <= 1.1 * r["oracle_err"] + 5e-4, r


def test_the_person_volume_is_tracked_on_both_sides(runs):
    ids = sorted(runs["objects"][FRAMES - 1])
    assert ids == [1]
    for f in range(1, FRAMES):
        assert runs["objects"][f][1]["track"]["iterations"] >= 1
        sep = np.linalg.norm(np.asarray(runs["objects"][f][1]["pose"][1], np.float64) - runs["oracle_objects"][f][1][1])
        assert sep < 5e-3, (f, sep)
