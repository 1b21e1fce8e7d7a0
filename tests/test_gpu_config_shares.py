"""The one-GPU shares of BASELINE.json's two 8-GPU configurations, at full size, on one MI355X:

  configs[3]  (also north_star's single-GPU target): 640 x 480, background 512^3 @ 1 cm + 8 objects
              128^3 -- the whole schedule through pipeline.Fusion against the frame-level oracle
              (tests/oracle_pipeline.py, reference EMFusion.cpp:70-129), and byte equality of the five
              execution paths of emf::EMFusion;
  configs[4]  1280 x 960, background 1024^3 @ 0.5 cm + 2 objects 256^3 -- two integrations, the
              foreground statistics, a raycast of every volume and one E-step against the oracle, every
              voxel and pixel (kernels: reference TSDF.cu:327-601, ObjTSDF.cu:29-107, TSDF.cpp:125-156,
              EMFusion.cpp:635-670), and the same five-path byte equality through pipeline.Fusion.

The 8-rank jobs themselves are the driver's; what a rank of them computes is what runs here.
"""
import os

import numpy as np
import pytest
import xxhash

from tests.oracle_pipeline import Affine32, OraclePipeline
from tests.parity_util import assert_parity, dev_full, to_dev, to_np
from tests.scenes import Pose, rel_CO, rel_OC

pytestmark = pytest.mark.gpu

SHARES = {
    "config3_share": dict(w=640, h=480, bg=512, vox=0.01, obj=128, nobj=8),
    "config4_share": dict(w=1280, h=960, bg=1024, vox=0.005, obj=256, nobj=2),
}
PATHS = (("second run", {}),
         ("per-volume launches", {"EMF_PER_VOLUME": "1"}),
         ("IEEE divisions, inline 1/lambda", {"EMF_VOXEL_RCP": "0", "EMF_LAMBDA_TABLE": "0"}),
         ("background integrated in place after the raycast", {"EMF_BG_OVERLAP": "0"}),
         ("every ray marched to the end of its range", {"EMF_FAR_BOUNDS": "0"}))


def _host_gib_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


def _digest(a: np.ndarray) -> str:
    return xxhash.xxh3_128(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()


@pytest.fixture(scope="module")
def ops(dev):
    from emfusion_amd import ops as _ops
    return _ops


def _run_share(cfg, frames, env=None, keep=False):
    """`frames` frames of the bench's synthetic stream through emf::EMFusion; digests of every volume and
    of the frame's images (arrays too with keep=True: only for sizes the host can hold twice)."""
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        prm = pipeline.make_params(cfg["w"], cfg["h"], cfg["bg"], cfg["vox"], cfg["obj"])
        Kp = np.array(prm.K, np.float32)
        synth = pipeline.SyntheticStream(cfg["w"], cfg["h"], Kp, cfg["nobj"], seed=0xE3F5)
        fus = pipeline.Fusion(prm, None)
        fus.enable_raycast_stats(True)
        ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(cfg["nobj"])]
        for f in range(frames):
            depth, sid = synth.render(f)
            R, t = synth.camera_pose(f)
            poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
            masks = {i: to_dev((sid == i).astype(np.uint8)) for i in ids} if f == 0 else {}
            d = to_dev(depth)
            fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, f == 0)
            fus.synchronize()
        out = dict(samples=fus.raycast_stats()[0], ids=ids, digest={}, arrays={}, visible=sorted(fus.visible_objects()))

        def take(name, a):
            out["digest"][name] = _digest(a)
            if keep:
                out["arrays"][name] = a
        for which in ("tsdf", "weights"):
            take(f"bg {which}", fus.volume(which, 0))
            for i in ids:
                take(f"obj {i} {which}", fus.volume(which, i))
        for i in ids:
            take(f"obj {i} fgprobs", fus.volume("fgprobs", i))
            take(f"obj {i} assoc", fus.image("obj_assoc", i))
            take(f"obj {i} raylengths", fus.image("obj_raylengths", i))
        for im in ("raylengths", "segmentation", "assoc_norm", "bg_assoc", "bg_raylengths", "vertices"):
            take(im, fus.image(im))
        seg = fus.image("segmentation")
        wts = fus.volume("weights", 0)
        out["seen_voxels"] = int((wts > 0).sum())
        out["object_pixels"] = int((seg > 0).sum())
        del wts
        fus.close()
        synth.close()
        return out
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)


@pytest.mark.parametrize("share", list(SHARES))
def test_share_is_repeatable_and_path_independent(dev, share):
    """Same bytes from the batched launches, the reference-shaped per-volume launches, the IEEE-division
    march, the in-place background integration and the march without far bounds: every volume of the
    share and every image of its last frame."""
    cfg = SHARES[share]
    if cfg["bg"] == 1024 and _host_gib_available() < 24:
        pytest.skip("needs ~16 GiB of host memory to download the 1024^3 volumes")
    frames = 6 if cfg["bg"] < 1024 else 4
    base = _run_share(cfg, frames)
    assert base["seen_voxels"] > 3e6 and base["object_pixels"] > 2000 and base["visible"]
    for what, env in PATHS:
        other = _run_share(cfg, frames, env)
        assert other["ids"] == base["ids"] and other["visible"] == base["visible"], what
        diff = [k for k in base["digest"] if base["digest"][k] != other["digest"][k]]
        assert not diff, (share, what, diff)
        if "EMF_FAR_BOUNDS" in env:  # the far bounds drop march samples, never an output
            assert base["samples"] < other["samples"], (base["samples"], other["samples"])


def test_config3_share_schedule_against_the_frame_level_oracle(oracle, dev):
    """bg 512^3 + 8 x 128^3 at 640 x 480 (north_star's single-GPU target): three frames of the whole
    schedule -- masks, three E-steps, raycast of nine volumes + compositing + visibility, gated
    integration -- HIP host classes vs the oracle's restatement of EMFusion::processFrame."""
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    cfg = SHARES["config3_share"]
    W, H = cfg["w"], cfg["h"]
    oracle.set_threads(oracle.host_threads())
    prm = pipeline.make_params(W, H, cfg["bg"], cfg["vox"], cfg["obj"])
    Kp = np.array(prm.K, np.float32)
    synth = pipeline.SyntheticStream(W, H, Kp, cfg["nobj"], seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    orc = OraclePipeline(oracle, W, H, Kp, cfg["bg"], cfg["vox"], list(prm.volume_pose_t), cfg["obj"])
    ids = []
    for k in range(cfg["nobj"]):
        c, _, vs = synth.sphere(k, 0)
        ids.append(fus.add_object(c, vs))
        assert orc.add_object(c, vs) == ids[-1]
    for f in range(3):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
        masks = {i: (sid == i).astype(np.uint8) for i in ids} if f == 0 else {}
        d_depth, d_masks = to_dev(depth), {i: to_dev(m) for i, m in masks.items()}
        fus.process_frame(image_view(d_depth), R, t, poses, {i: image_view(m) for i, m in d_masks.items()}, f == 0)
        fus.synchronize()
        orc.process_frame(depth, Affine32(R.reshape(3, 3), t),
                          {i: Affine32(p[0].reshape(3, 3), p[1]) for i, p in poses.items()}, masks, f == 0)
        assert sorted(fus.visible_objects()) == sorted(orc.vis), f"visible set, frame {f}"
    assert len(orc.vis) >= 3
    # association weights pass through expf and then through the running average (north_star: 1e-4)
    assert_parity(fus.volume("tsdf", 0), orc.bg["tsdf"], "bg tsdf", rtol=1e-4, atol=1e-6, budget=1e-3)
    assert_parity(fus.volume("weights", 0), orc.bg["wts"], "bg weights", rtol=1e-4, atol=1e-6, budget=1e-3)
    total = fus.image("bg_assoc").astype(np.float64)
    for v in orc.objects:
        i = v["id"]
        assert_parity(fus.volume("tsdf", i), v["tsdf"], f"obj {i} tsdf", rtol=1e-4, atol=1e-6, budget=1e-3)
        assert_parity(fus.volume("weights", i), v["wts"], f"obj {i} weights", rtol=1e-4, atol=1e-6, budget=1e-3)
        assert_parity(fus.volume("fgprobs", i), v["probs"], f"obj {i} fgProbs", rtol=1e-4, atol=1e-6, budget=1e-3)
        assert (fus.volume("fgmask", i) != v["vmask"]).mean() < 1e-3 and (v["vmask"] > 0).sum() > 50
        a = fus.image("obj_assoc", i)
        assert_parity(a, v["assoc"], f"obj {i} association", rtol=1e-4, atol=1e-7, budget=1e-3)
        assert_parity(fus.image("obj_raylengths", i), v["ray"], f"obj {i} raylengths", rtol=1e-4, budget=5e-3)
        total += a
    assert_parity(fus.image("points"), orc.points, "points", exact=True)
    assert_parity(fus.image("assoc_norm"), orc.norm, "associationNorm", rtol=1e-4, budget=1e-3)
    assert_parity(fus.image("bg_assoc"), orc.bg_assoc, "bg association", rtol=1e-4, atol=1e-7, budget=1e-3)
    valid = orc.norm != 0
    assert valid.mean() > 0.9 and np.allclose(total[valid], 1.0, atol=1e-5) and np.all(total[~valid] == 0)
    seg = fus.image("segmentation")
    assert (seg != orc.seg).mean() < 2e-3 and (orc.seg > 0).sum() > 2000
    same = seg == orc.seg
    assert_parity(fus.image("raylengths")[same], orc.ray[same], "composite raylengths", rtol=1e-4, budget=5e-3)
    assert_parity(fus.image("bg_raylengths"), orc.bg_ray, "bg raylengths", rtol=1e-4, budget=5e-3)
    hit = (orc.ray > 0) & same
    assert_parity(fus.image("normals")[hit], orc.nrm[hit], "normals", rtol=1e-3, atol=1e-4, budget=1e-2)
    fus.close()
    synth.close()
    oracle.set_threads(min(8, os.cpu_count() or 1))


def test_config4_share_kernels_against_the_oracle(oracle, ops, dev):
    """1280 x 960, bg 1024^3 @ 0.5 cm + 2 objects 256^3: two association-weighted integrations of every
    volume, fg/bg statistics + foreground probability of the objects, the raycast of every volume (the
    objects' gated by their foreground mask) and one E-step over the three models.  Bit-exact wherever
    the arithmetic is IEEE-exact (integration, foreground, raycast incl. normals and the march-sample
    count), 4e-6 relative behind expf."""
    from emfusion_amd import pipeline
    cfg = SHARES["config4_share"]
    W, H, n, vox, m = cfg["w"], cfg["h"], cfg["bg"], cfg["vox"], cfg["obj"]
    if _host_gib_available() < 64:
        pytest.skip("needs ~40 GiB of host memory for the 1024^3 volumes")
    oracle.set_threads(oracle.host_threads())
    prm = pipeline.make_params(W, H, n, vox, m)
    Kp = np.array(prm.K, np.float32).reshape(3, 3)
    synth = pipeline.SyntheticStream(W, H, Kp.reshape(-1), cfg["nobj"], seed=0xE3F5)
    rng = np.random.default_rng(4)
    il = dev_full((H, W), 0.0)
    ops.compute_inv_lambda(Kp, il)

    models = [dict(name="bg 1024^3", n=n, vox=np.float32(vox), trunc=np.float32(10) * np.float32(vox),
                   pose=lambda f: Pose(t=list(prm.volume_pose_t)), obj=None)]
    for k in range(cfg["nobj"]):
        vs = synth.sphere(k, 0)[2]
        ovox = np.float32(np.float32(vs) / np.float32(m))  # EMFusion::addObject: volSize / float(res)
        models.append(dict(name=f"obj {k + 1} 256^3", n=m, vox=ovox, trunc=np.float32(np.float32(10 * np.float32(vs)) / np.float32(m)),
                           pose=(lambda f, k=k: Pose(t=synth.sphere(k, f)[0].astype(np.float64))), obj=k + 1))
    for mdl in models:
        N = mdl["n"]
        mdl["tsdf"], mdl["wts"] = np.zeros((N, N, N), np.float32), np.zeros((N, N, N), np.float32)
        mdl["d_t"], mdl["d_w"] = to_dev(mdl["tsdf"]), to_dev(mdl["wts"])
        if mdl["obj"]:
            mdl["fgbg"] = np.zeros((N, N, N, 2), np.float32)
            mdl["d_fgbg"] = to_dev(mdl["fgbg"])
    for f in range(2):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        cam = Pose(R.reshape(3, 3).astype(np.float64), t.astype(np.float64))
        d_depth = to_dev(depth)
        for mdl in models:
            oc = rel_OC(cam, mdl["pose"](f))
            assoc = rng.uniform(0.3, 1.0, (H, W)).astype(np.float32)
            oracle.update_tsdf(depth, assoc, mdl["tsdf"], mdl["wts"], oc.R32, oc.t32, Kp, mdl["vox"], mdl["trunc"], 64.0)
            ops.update_tsdf(d_depth, to_dev(assoc), mdl["d_t"], mdl["d_w"], oc.R32, oc.t32, Kp, float(mdl["vox"]),
                            float(mdl["trunc"]), 64.0, inv_lambda=il)
            if mdl["obj"]:
                mask = (sid == mdl["obj"]).astype(np.uint8)
                occ = ((sid != 0) & (sid != mdl["obj"]) & (rng.uniform(size=(H, W)) < 0.5)).astype(np.uint8)
                oracle.update_fgbg_probs(mask, occ, mdl["tsdf"], mdl["wts"], mdl["fgbg"], oc.R32, oc.t32, Kp, mdl["vox"])
                ops.update_fgbg_probs(to_dev(mask), to_dev(occ), mdl["d_t"], mdl["d_w"], mdl["d_fgbg"], oc.R32, oc.t32,
                                      Kp, float(mdl["vox"]))
    for mdl in models:
        assert_parity(to_np(mdl["d_t"]), mdl["tsdf"], f"tsdf {mdl['name']}", exact=True)
        assert_parity(to_np(mdl["d_w"]), mdl["wts"], f"weights {mdl['name']}", exact=True)
        assert (mdl["wts"] > 0).sum() > (3e6 if not mdl["obj"] else 1e5), mdl["name"]
        mdl["probs"], mdl["vmask"], mdl["d_probs"], mdl["d_vmask"] = None, None, None, None
        if mdl["obj"]:
            N = mdl["n"]
            assert_parity(to_np(mdl["d_fgbg"]), mdl["fgbg"], f"fg/bg counts {mdl['name']}", exact=True)
            mdl["probs"], mdl["vmask"] = oracle.compute_fg_probs(mdl["fgbg"])
            mdl["d_probs"], mdl["d_vmask"] = dev_full((N, N, N), 9.0), dev_full((N, N, N), 9, np.uint8)
            ops.compute_fg_probs(mdl["d_fgbg"], mdl["d_probs"], mdl["d_vmask"])
            assert_parity(to_np(mdl["d_probs"]), mdl["probs"], f"fgProbs {mdl['name']}", exact=True)
            assert_parity(to_np(mdl["d_vmask"]), mdl["vmask"], f"fgVolMask {mdl['name']}", exact=True)
            assert (mdl["vmask"] > 0).sum() > 1e4
    # raycast of every volume from the second frame's camera
    hits = 0
    for mdl in models:
        co = rel_CO(cam, mdl["pose"](1))
        want = oracle.raycast_tsdf(mdl["tsdf"], None, mdl["wts"], mdl["vmask"], W, H, co.R32, co.t32, Kp, mdl["vox"],
                                   mdl["trunc"], count_steps=True)
        ray, vert, nrm = dev_full((H, W), 0.0), dev_full((H, W, 3), 0.0), dev_full((H, W, 3), 0.0)
        hit, st = dev_full((H, W), 0, np.uint8), dev_full((4,), 0, np.uint64)
        ops.raycast_tsdf(mdl["d_t"], None, mdl["d_w"], mdl["d_vmask"], ray, vert, nrm, hit, co.R32, co.t32, Kp,
                         float(mdl["vox"]), float(mdl["trunc"]), st, rcp_voxel=ops.voxel_reciprocal(float(mdl["vox"])))
        for got, w_, name in zip((ray, vert, nrm, hit), want, ("ray", "vert", "normal", "mask")):
            assert_parity(to_np(got), w_, f"{name} 1280x960 / {mdl['name']}", exact=True)
        assert int(to_np(st)[0]) == int(want[4].sum()), mdl["name"]
        hits += int(want[3].sum())
        assert want[3].sum() > (1e6 if not mdl["obj"] else 2e3), (mdl["name"], int(want[3].sum()))
    # one E-step over [background, object 1, object 2]
    pts = oracle.compute_points(depth, Kp)
    d_pts = dev_full((H, W, 3), 9.0)
    ops.compute_points(d_depth, Kp, d_pts)
    assert_parity(to_np(d_pts), pts, "points 1280x960", exact=True)
    raw, d_maps = [], []
    for mdl in models:
        co = rel_CO(cam, mdl["pose"](1))
        raw.append(oracle.compute_association(mdl["tsdf"], mdl["probs"], pts, co.R32, co.t32, mdl["vox"], mdl["trunc"],
                                              0.02, 0.8, 1.0))
        out = dev_full((H, W), 9.0)
        ops.compute_association(mdl["d_t"], mdl["d_probs"], d_pts, co.R32, co.t32, float(mdl["vox"]), float(mdl["trunc"]),
                                0.02, 0.8, 1.0, out)
        d_maps.append(out)
        assert_parity(to_np(out), raw[-1], f"un-normalised association {mdl['name']}", rtol=2e-6)
        assert np.array_equal(to_np(out) == 0, raw[-1] == 0), "association mask (exact-zero lookups)"
    want = [r.copy() for r in raw]
    norm = oracle.normalize_association(want)
    d_norm = dev_full((H, W), 0.0)
    ops.normalize_association(d_maps, norm=d_norm)
    assert_parity(to_np(d_norm), norm, "associationNorm", rtol=2e-6)
    for mdl, dm, wv in zip(models, d_maps, want):
        assert_parity(to_np(dm), wv, f"association weights {mdl['name']}", rtol=4e-6)
    assert (norm != 0).mean() > 0.9 and (want[1] > 0.25).sum() > 1000
    synth.close()
    oracle.set_threads(min(8, os.cpu_count() or 1))
