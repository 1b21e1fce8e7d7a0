"""Phong rendering: emf_hip_renderPhong against the oracle's restatement of
kernel_renderPhong + the colour lookup of renderGPU, and EMFusion::render on a short run."""
import numpy as np
import pytest

from tests.parity_util import dev_full, to_dev

pytestmark = pytest.mark.gpu


def test_render_equals_the_reference_restatement(oracle, dev):
    from emfusion_amd import ops
    rng = np.random.default_rng(12)
    H, W = 61, 83
    pts = rng.uniform(-1, 1, (H, W, 3)).astype(np.float32)
    pts[..., 2] = rng.uniform(0.4, 3.0, (H, W)).astype(np.float32)
    nrm = rng.standard_normal((H, W, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    nrm[..., 2] = -np.abs(nrm[..., 2])            # mostly facing the camera
    nrm[5, 5] = (0, 0, 1)                          # facing away: negative diffuse term
    nrm[6, 6] = 0                                  # degenerate normal: r = -l
    nrm[7, 7] = np.nan                             # NaN normal: every channel becomes 0
    hole = rng.uniform(size=(H, W)) < 0.2
    pts[hole] = 0                                  # no vertex: stays black
    seg = rng.integers(0, 256, (H, W)).astype(np.uint8)
    cmap = rng.integers(0, 256, (256, 3)).astype(np.uint8)
    for light in ((0.0, 0.0, 0.0), (0.5, -0.25, 0.1)):
        want = oracle.render_phong(pts, nrm, seg, cmap, light)
        img = dev_full((H, W, 3), 77, np.uint8)
        ops.render_phong(to_dev(pts), to_dev(nrm), to_dev(seg), cmap, img, light)
        got = img.numpy()
        assert np.array_equal(got, want)
        assert np.all(got[hole] == 0) and got[~hole].max() > 100


def test_fusion_render_shows_the_labelled_models(oracle, dev):
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    W, H = 160, 120
    prm = pipeline.make_params(W, H, 64, 0.04, 32, visibility_thresh=100, boundary=5)
    synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), 1, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    rgb, cmap = fus.render()
    assert not rgb.any()                                       # before the first frame
    assert np.all(cmap[0] == 255) and len(np.unique(cmap, axis=0)) > 200
    c, r, vs = synth.sphere(0, 0)
    oid = fus.add_object(c, vs)
    keep = []
    for f in range(3):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        d, m = to_dev(depth), to_dev((sid == 1).astype(np.uint8))
        keep += [d, m]
        fus.process_frame(image_view(d), R, t, {oid: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(0, f)[0])},
                          {oid: image_view(m)}, f == 0)
        if f == 0:
            first, _ = fus.render()                            # raycasts once for the view
            assert first.any()
    fus.synchronize()
    rgb, cmap = fus.render()
    seg = fus.image("segmentation")
    want = oracle.render_phong(fus.image("vertices"), fus.image("normals"), seg, cmap)
    assert np.array_equal(rgb, want)
    assert (seg == oid).sum() > 100
    obj_px, bg_px = rgb[seg == oid].astype(int), rgb[(seg == 0) & rgb.any(axis=2)].astype(int)
    # the background is white-ish grey (colour 255,255,255), the object carries its label's hue
    assert np.abs(bg_px[:, 0] - bg_px[:, 1]).max() <= 1 and np.abs(bg_px[:, 1] - bg_px[:, 2]).max() <= 1
    assert (obj_px.max(axis=1) - obj_px.min(axis=1)).mean() > 20
    fus.close()
    synth.close()
