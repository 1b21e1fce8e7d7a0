"""Depth pre-processing (SURVEY f-2): emf_hip_preprocessDepth against the oracle's restatement of
EMFusion::preprocessDepth (cv::cuda::bilateralFilter + NaN / zero patches, EMFusion.cpp:294-305)."""
import numpy as np
import pytest

from tests.parity_util import assert_parity, dev_full, to_dev, to_np
from tests.scenes import camera_path, intrinsics, render_depth

pytestmark = pytest.mark.gpu
SPHERES = [((0.25, 0.05, 1.3), 0.22), ((-0.3, -0.1, 1.6), 0.18)]


@pytest.fixture(scope="module")
def ops(dev):
    from emfusion_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("w,h,pad,ksz", [(160, 120, 0, 7), (161, 77, 3, 7), (96, 72, 0, 5), (40, 9, 0, 15)])
def test_bilateral_filter_and_patches(oracle, ops, dev, w, h, pad, ksz):
    K = intrinsics(w, h)
    depth, _ = render_depth(w, h, K, camera_path(2), SPHERES, noise=0.004, dropout=0.03, seed=7)
    depth[3, 5] = np.nan  # a NaN input poisons its neighbourhood: those outputs are patched to 0
    want = oracle.preprocess_depth(depth, ksz, 0.04, 4.5)
    out = dev_full((h, w), -5.0, pad_cols=pad)
    ops.preprocess_depth(to_dev(depth, dev, pad), out, ksz, 0.04, 4.5)
    got = to_np(out)
    assert (want[depth == 0] == 0).all() and (got[depth == 0] == 0).all()
    assert not np.isnan(got).any() and (got[2:5, 4:7] == 0).all()
    smoothed = np.abs(want - np.nan_to_num(depth))[(depth > 0) & (want > 0)]
    assert smoothed.mean() > 1e-4  # it filters
    assert_parity(got, want, "pre-processed depth", rtol=1e-5, atol=1e-7)  # expf on both sides


def test_argument_checks(ops, dev):
    from emfusion_amd._lib import EmfHipError
    a, b = dev_full((8, 8), 1.0), dev_full((8, 8), 0.0)
    for bad in (dict(ksz=4), dict(ksz=17), dict(sigma_depth=0.0)):
        with pytest.raises(EmfHipError):
            ops.preprocess_depth(a, b, **{**dict(ksz=7, sigma_depth=0.04, sigma_spatial=4.5), **bad})
    with pytest.raises(EmfHipError):
        ops.preprocess_depth(a, a)


def test_pipeline_filters_depth_when_asked(oracle, dev):
    """Two frames through emf::EMFusion with preprocessing on == the oracle pipeline fed with
    oracle-filtered depth."""
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    from tests.oracle_pipeline import Affine32, OraclePipeline
    W, H = 160, 120
    prm = pipeline.make_params(W, H, 64, 0.04, 32, visibility_thresh=100, boundary=5)
    K = np.array(prm.K, np.float32)
    synth = pipeline.SyntheticStream(W, H, K, 0, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    fus.set_preprocess(True)
    orc = OraclePipeline(oracle, W, H, K, 64, 0.04, list(prm.volume_pose_t), 32, visibility_thresh=100,
                         boundary=5)
    for f in range(2):
        depth, _ = synth.render(f)
        R, t = synth.camera_pose(f)
        d = to_dev(depth)
        fus.process_frame(image_view(d), R, t, {}, {}, False)
        fus.synchronize()
        orc.process_frame(oracle.preprocess_depth(depth), Affine32(R.reshape(3, 3), t))
    assert_parity(fus.image("points"), orc.points, "points of the filtered depth", rtol=1e-5, atol=1e-7)
    assert_parity(fus.volume("tsdf", 0), orc.bg["tsdf"], "bg tsdf", rtol=1e-4, atol=1e-6, budget=1e-3)
    fus.close()
    synth.close()
