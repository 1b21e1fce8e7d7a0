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
