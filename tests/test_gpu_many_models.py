"""More models than one batched launch takes (EMF_MAX_BATCH = 32 table slots): the reference loops over any number of
objects (EMFusion.cpp:635-670, 726-795, 865-889), so the batched path serves a longer model list in chunks of the table
-- E-step likelihoods per chunk + one normalisation over all maps, far bounds / raycast / integration per chunk, one
composite over everything -- instead of dropping the frame to the per-volume path as rounds 1-5 did.

Scene: background 256^3 (large enough for relevant-tile lists) + 40 / 70 objects 32^3 at 320 x 240, five frames with a
moving camera, moving objects and a mask frame.  Compared, bit for bit: every volume and every image of the chunked
batched run against the per-volume run (EMF_PER_VOLUME=1: the reference's structure, one launch per volume)."""
import os

import numpy as np
import pytest
import xxhash

from tests.parity_util import to_dev

pytestmark = pytest.mark.gpu
W, H = 320, 240


def _digest(a):
    return xxhash.xxh3_128(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()


def _scene(nobj):
    from emfusion_amd import pipeline
    prm = pipeline.make_params(W, H, 256, 0.02, 32, visibility_thresh=20, boundary=10, mask_frames=3)
    synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), nobj, seed=0xE3F5)
    ids = list(range(1, nobj + 1))
    frames = []
    for f in range(5):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        masks = {i: to_dev((sid == i).astype(np.uint8)) for i in ids} if f % 3 == 0 else {}
        frames.append((to_dev(depth), R, t, masks, {i: synth.sphere(i - 1, f)[0] for i in ids}))
    first = [synth.sphere(k, 0) for k in range(nobj)]
    synth.close()
    return prm, frames, first


def _run(scene, env, track=False):
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    prm, frames, first = scene
    for k, v in env.items():
        os.environ[k] = v
    try:
        fus = pipeline.Fusion(prm, None)
        ids = [fus.add_object(c, vs) for c, _, vs in first]
        for d, R, t, masks, centres in frames:
            poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), centres[i]) for i in ids}
            fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, bool(masks))
        fus.synchronize()
        out = {"vis": tuple(sorted(fus.visible_objects())), "_chunks": fus.batched_chunks()}
        for i in [0] + ids:
            out[f"tsdf {i}"] = _digest(fus.volume("tsdf", i))
            out[f"weights {i}"] = _digest(fus.volume("weights", i))
        for i in ids:
            out[f"fgprobs {i}"] = _digest(fus.volume("fgprobs", i))
            out[f"assoc {i}"] = _digest(fus.image("obj_assoc", i))
            out[f"ray {i}"] = _digest(fus.image("obj_raylengths", i))
        for im in ("raylengths", "segmentation", "assoc_norm", "bg_assoc", "bg_raylengths"):
            out[im] = _digest(fus.image(im))
        seg = fus.image("segmentation")
        out["_labels"] = len(np.unique(seg)) - 1
        fus.close()
        return out
    finally:
        for k in env:
            os.environ.pop(k, None)


@pytest.mark.parametrize("nobj,chunks", [(31, 1), (32, 2), (40, 2), (70, 3)])
def test_chunked_batched_path_equals_the_per_volume_path(dev, nobj, chunks):
    scene = _scene(nobj)
    got = _run(scene, {})
    want = _run(scene, {"EMF_PER_VOLUME": "1"})
    assert got.pop("_chunks") == chunks and want.pop("_chunks") == 0
    assert got["_labels"] >= min(nobj, 12) // 2, "too few objects in the composite for the test to mean anything"
    assert len(got["vis"]) >= 5
    bad = [k for k in want if got[k] != want[k]]
    assert not bad, f"{len(bad)} of {len(want)} outputs differ between the chunked batched and the per-volume path: {bad[:8]}"


def test_the_switches_keep_the_bytes_with_two_chunks(dev):
    """in-place background (no overlap), no far bounds, no box cull: the other launch forms the chunks go through"""
    scene = _scene(40)
    want = _run(scene, {})
    want.pop("_chunks")
    for env in ({"EMF_BG_OVERLAP": "0"}, {"EMF_FAR_BOUNDS": "0"}, {"EMF_INT_CULL": "0"}, {"EMF_BG_OVERLAP": "0", "EMF_INT_CULL": "0"}):
        got = _run(scene, env)
        assert got.pop("_chunks") == 2
        bad = [k for k in want if got[k] != want[k]]
        assert not bad, f"{env}: {bad[:8]}"


def test_tracking_runs_over_more_than_one_chunk(dev):
    """Camera + object tracking with 40 objects: the object stage runs chunk by chunk of the table; every model gets a
    result and the tracked camera stays on the supplied trajectory."""
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    prm, frames, first = _scene(40)
    fus = pipeline.Fusion(prm, None)
    ids = [fus.add_object(c, vs) for c, _, vs in first]
    fus.set_tracking(camera=True, objects=True)
    for f, (d, R, t, masks, centres) in enumerate(frames):
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), centres[i]) for i in ids}
        fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, bool(masks))
    fus.synchronize()
    assert fus.batched_chunks() == 2
    Rt, tt = fus.pose(0)
    _, R, t, _, _ = frames[-1]
    assert np.abs(np.asarray(tt) - np.asarray(t)).max() < 0.02, (tt, t)
    results = [fus.track_result(i) for i in [0] + ids]
    assert all(r is not None for r in results)
    assert sum(1 for r in results if r["iterations"] > 0) >= 10
    fus.close()
