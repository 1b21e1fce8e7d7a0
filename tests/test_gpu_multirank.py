"""Cross-GPU compositing kernels (packHitKeys / compositeFromKeys / visibilityFlagsIndexed) on one
GPU, emulating two ranks: each "rank" packs the keys of the objects it owns, the element-wise
minimum stands in for the RCCL all-reduce(min), and every rank's finished composite must equal
the oracle's list-order composite of ALL objects (vertices / normals only where the winner is
local or the background)."""
import numpy as np
import pytest

from emfusion_amd import sharding
from tests.parity_util import assert_parity, dev_full, to_dev, to_np

pytestmark = pytest.mark.gpu

W, H = 160, 120


@pytest.fixture(scope="module")
def ops(dev):
    from emfusion_amd import ops as _ops
    return _ops


def _scene(nobj, seed):
    rng = np.random.default_rng(seed)
    ids = list(range(1, nobj + 1))
    hit = [(rng.random((H, W)) < 0.3).astype(np.uint8) for _ in ids]
    ray = [(rng.uniform(0.5, 3, (H, W)).astype(np.float32) * s) for s in hit]
    if nobj >= 4:
        ray[3][:20], hit[3][:20] = ray[0][:20], hit[0][:20]  # ties across "ranks"
    vert = [rng.standard_normal((H, W, 3)).astype(np.float32) for _ in ids]
    nrm = [rng.standard_normal((H, W, 3)).astype(np.float32) for _ in ids]
    bg_mask = (rng.random((H, W)) < 0.8).astype(np.uint8)
    bg_ray = rng.uniform(0.5, 3, (H, W)).astype(np.float32) * bg_mask
    bg_vert = rng.standard_normal((H, W, 3)).astype(np.float32)
    bg_norm = rng.standard_normal((H, W, 3)).astype(np.float32)
    diff0 = (rng.uniform(-1, 1, (H, W)) * (rng.random((H, W)) < 0.3)).astype(np.float32)
    return dict(ids=ids, hit=hit, ray=ray, vert=vert, nrm=nrm, bg_mask=bg_mask, bg_ray=bg_ray,
                bg_vert=bg_vert, bg_norm=bg_norm, diff0=diff0)


@pytest.mark.parametrize("nobj,world", [(5, 2), (9, 4), (3, 8)])
def test_two_rank_composite_equals_single_gpu_composite(ops, oracle, dev, nobj, world):
    sc = _scene(nobj, 70 + nobj)
    diff = sc["diff0"].copy()
    want = oracle.composite_raycast(sc["ids"], sc["ray"], sc["vert"], sc["nrm"], sc["hit"],
                                    sc["bg_ray"], sc["bg_vert"], sc["bg_norm"], sc["bg_mask"],
                                    diff, 10)
    w_ray, w_vert, w_nrm, w_seg, w_noobj, w_vis = want
    pos = {i: k for k, i in enumerate(sc["ids"])}
    d = lambda a: to_dev(a)
    # per-rank key packing on the device, checked against the numpy restatement
    rank_keys = []
    for r in range(world):
        mine = sharding.local_objects(sc["ids"], r, world)
        keys = dev_full((H, W), 0, np.uint64)
        ops.pack_hit_keys([pos[i] for i in mine], [d(sc["ray"][pos[i]]) for i in mine],
                          [d(sc["hit"][pos[i]]) for i in mine], keys, W, H)
        got = to_np(keys)
        ref = sharding.pack_hit_keys([sc["ray"][pos[i]] for i in mine],
                                     [sc["hit"][pos[i]] for i in mine], [pos[i] for i in mine]) \
            if mine else np.full((H, W), sharding.NO_HIT, np.uint64)
        assert np.array_equal(got, ref), f"keys of rank {r}"
        rank_keys.append(got)
    merged = rank_keys[0]
    for k in rank_keys[1:]:
        merged = np.minimum(merged, k)  # what ncclAllReduce(min, u64) delivers
    d_merged = d(merged)
    for r in range(world):
        mine = sharding.local_objects(sc["ids"], r, world)
        lp = [pos[i] for i in mine]
        ray, vert, nrm = dev_full((H, W), 5.0), dev_full((H, W, 3), 5.0), dev_full((H, W, 3), 5.0)
        seg, no_obj = dev_full((H, W), 5, np.uint8), dev_full((H, W), 5, np.uint8)
        d_diff = d(sc["diff0"])
        vis = dev_full((nobj,), -1, np.int32)
        ops.composite_from_keys(d_merged, sc["ids"], lp, [d(sc["ray"][p]) for p in lp],
                                [d(sc["vert"][p]) for p in lp], [d(sc["nrm"][p]) for p in lp],
                                d(sc["bg_ray"]), d(sc["bg_vert"]), d(sc["bg_norm"]),
                                d(sc["bg_mask"]), ray, vert, nrm, seg, d_diff, no_obj, 10, vis)
        assert_parity(to_np(seg), w_seg, f"segmentation (rank {r})", exact=True)
        assert_parity(to_np(ray), w_ray, f"raylengths (rank {r})", exact=True)
        assert_parity(to_np(no_obj), w_noobj, f"noObj (rank {r})", exact=True)
        assert_parity(to_np(d_diff), diff, f"diffRaylengths (rank {r})", exact=True)
        assert to_np(vis).tolist() == w_vis.tolist(), f"visibility counts (rank {r})"
        local_or_bg = (w_seg == 0) | np.isin(w_seg, mine)
        assert_parity(to_np(vert)[local_or_bg], w_vert[local_or_bg], "vertices", exact=True)
        assert_parity(to_np(nrm)[local_or_bg], w_nrm[local_or_bg], "normals", exact=True)
        assert np.all(to_np(vert)[~local_or_bg] == 0)  # remote winners: not gathered
        # device-side gate for this rank's model slots
        visible = dev_full((len(mine) + 1,), -1, np.int32)
        ops.visibility_flags_indexed(vis, [0] + lp, 300, visible)
        assert to_np(visible).tolist() == [1] + [int(w_vis[p] > 300) for p in lp]
    assert w_vis.sum() > 0 and (w_seg > 0).any()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_background_raycast_bands_tile_the_image(ops, oracle, dev, world):
    """Multi-GPU background split: every emulated rank marches only its row band of table slot 0
    into the shared images (what the gather of the bands leaves on every rank); the result must be
    the full raycast, bit for bit, and the objects' images must not depend on the band."""
    from tests.scenes import Pose, camera_path, intrinsics, rel_CO, rel_OC, render_depth
    K = intrinsics(W, H)
    spheres = [((0.25, 0.05, 1.3), 0.22)]
    vols = [dict(n=(64, 64, 64), vox=0.04, pose=Pose(t=[0, 0, 1.28])),
            dict(n=(32, 32, 32), vox=0.02, pose=Pose(t=spheres[0][0]))]
    keep, entries = [], []
    for k, v in enumerate(vols):
        n = v["n"]
        tsdf, wts = np.zeros((n[2], n[1], n[0]), np.float32), np.zeros((n[2], n[1], n[0]), np.float32)
        for i in range(3):
            cam = camera_path(i)
            depth, _ = render_depth(W, H, K, cam, spheres, noise=0.002, dropout=0.01, seed=40 + i)
            oc = rel_OC(cam, v["pose"])
            oracle.update_tsdf(depth, np.ones((H, W), np.float32), tsdf, wts, oc.R32, oc.t32, K, v["vox"],
                               10 * v["vox"], 64.0)
        d = dict(tsdf=to_dev(tsdf), wts=to_dev(wts), assoc=dev_full((H, W), 1.0), ray=dev_full((H, W), 7.0),
                 vert=dev_full((H, W, 3), 7.0), nrm=dev_full((H, W, 3), 7.0), hit=dev_full((H, W), 7, np.uint8))
        keep.append(d)
        entries.append(ops.make_model(d["tsdf"], d["wts"], d["assoc"], d["ray"], d["vert"], d["nrm"], d["hit"],
                                      float(np.float32(v["vox"])), float(np.float32(10 * v["vox"])), 64.0,
                                      0.02, 0.8, 1.0, model_id=k, rcp_voxel=ops.voxel_reciprocal(v["vox"])))
    table = ops.upload_models(entries)
    cam = camera_path(3)
    poses = [(rel_CO(cam, v["pose"]).R32, rel_CO(cam, v["pose"]).t32) for v in vols]
    res = [v["n"] for v in vols]
    ops.raycast_batched(table, poses, res, W, H, K)  # single GPU: all rows
    full = {k: [to_np(keep[m][k]) for m in (0, 1)] for k in ("ray", "vert", "nrm", "hit")}
    assert full["hit"][0].sum() > 5000
    for d in keep:  # poison, then let the emulated ranks fill their bands
        d["ray"].copy_from(np.full((H, W), 9, np.float32))
        d["hit"].copy_from(np.full((H, W), 9, np.uint8))
    covered = np.zeros(H, bool)
    for rank in range(world):
        r0, rows = sharding.bg_band(rank, world, H)
        band_rows = ((((H + 15) // 16) + world - 1) // world) * 16
        ops.raycast_batched(table, poses, res, W, H, K, bg_band=(r0, band_rows))
        covered[r0:r0 + rows] = True
    assert covered.all()
    for k in ("ray", "hit"):
        assert to_np(keep[0][k]).tobytes() == full[k][0].tobytes(), k   # background: union of bands
        assert to_np(keep[1][k]).tobytes() == full[k][1].tobytes(), k   # object: untouched by bands
    with pytest.raises(Exception):
        ops.raycast_batched(table, poses, res, W, H, K, bg_band=(8, 16))  # not tile aligned
