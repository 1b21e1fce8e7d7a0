"""Every run-time switch of emf::EMFusion keeps the results bit-identical on its own (the per-feature tests);
this test sets them TWO AT A TIME -- all pairs -- because a switch that is safe alone can still meet another
one's assumptions (a tile map one path maintains and another reads, a stream one path forks and another joins).

Scene: background 256^3 (8192 tiles: large enough for the relevant-tile lists the far bounds read) + 2 objects
32^3 at 320 x 240, five frames with a moving camera and a mask frame; compared: digests of every volume and of
the last frame's images against the default configuration."""
import itertools
import os

import numpy as np
import pytest
import xxhash

from tests.parity_util import to_dev

pytestmark = pytest.mark.gpu

# every run-time switch of the product build that selects an execution path (round 5: the switches whose A/B is on
# record as lost -- ray footprints, brick flags, the objects' far-bound scan, un-fused points / visibility, late far
# bounds, ... -- are read by -DEMF_DEBUG_SWITCHES builds only, core/types.hpp debugEnv)
SWITCHES = [("EMF_PER_VOLUME", "1"), ("EMF_INT_CULL", "0"), ("EMF_LAMBDA_TABLE", "0"), ("EMF_VOXEL_RCP", "0"),
            ("EMF_BG_OVERLAP", "0"), ("EMF_FAR_BOUNDS", "0"), ("EMF_UNSEEN_TILES", "0"), ("EMF_DEEP_TILES", "0"),
            ("EMF_MARCH_ROWS", "2"), ("EMF_MARCH_ROWS", "4")]
# EMF_FUSION_VARIANT=_dbg (tests/test_gpu_debug_switches.py runs this module against libemf_fusion_dbg.so): the demoted
# switches, which only that build reads -- their paths stay compiled in and must stay exact
if os.environ.get("EMF_FUSION_VARIANT") == "_dbg":
    SWITCHES = [("EMF_FUSE_POINTS", "0"), ("EMF_FUSE_VISIBILITY", "0"), ("EMF_EARLY_FAR_BOUNDS", "0"), ("EMF_OBJ_CULL", "1"),
                ("EMF_FAR_SCAN", "1"), ("EMF_RAY_FOOTPRINTS", "0"), ("EMF_BRICK_FLAGS", "1"), ("EMF_PER_VOLUME", "1"),
                ("EMF_BG_OVERLAP", "0")]
W, H = 320, 240


def _digest(a):
    return xxhash.xxh3_128(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()


@pytest.fixture(scope="module")
def scene(dev):
    from emfusion_amd import pipeline
    prm = pipeline.make_params(W, H, 256, 0.02, 32, visibility_thresh=200, boundary=10, mask_frames=3)
    synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), 2, seed=0xE3F5)
    frames = []
    for f in range(5):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        frames.append((to_dev(depth), R, t, {i: to_dev((sid == i).astype(np.uint8)) for i in (1, 2)} if f % 3 == 0 else {},
                       {i: synth.sphere(i - 1, f)[0] for i in (1, 2)}))
    first = [synth.sphere(k, 0) for k in range(2)]
    synth.close()
    return prm, frames, first


def _run(scene, env):
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
        out = {"vis": tuple(sorted(fus.visible_objects()))}
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
        out["_objects_seen"] = int((seg > 0).sum())
        fus.close()
        return out
    finally:
        for k in env:
            os.environ.pop(k, None)


def test_every_pair_of_switches_keeps_the_bytes(scene):
    base = _run(scene, {})
    assert base["_objects_seen"] > 300 and base["vis"], base
    singles = {sw: _run(scene, dict([sw])) for sw in SWITCHES}
    bad = [(sw, [k for k in base if base[k] != r[k]]) for sw, r in singles.items() if r != base]
    assert not bad, bad
    pairs = [(a, b) for a, b in itertools.combinations(SWITCHES, 2) if a[0] != b[0]]
    same_var = sum(1 for a, b in itertools.combinations(SWITCHES, 2) if a[0] == b[0])
    assert len(pairs) == len(SWITCHES) * (len(SWITCHES) - 1) // 2 - same_var  # (two settings of one variable exclude each other)
    for a, b in pairs:
        r = _run(scene, dict([a, b]))
        diff = [k for k in base if base[k] != r[k]]
        assert not diff, (a, b, diff)
