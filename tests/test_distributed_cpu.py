"""World-size-2 gloo tests (CPU) of the object-sharded multi-GPU design:
the two exchanges -- all-reduce(sum) of the object association partials and all-reduce(min) of the
packed nearest-hit keys -- reproduce the single-process results, with the oracle standing in for
the device kernels (test infrastructure) and torch.distributed used exactly as bench.py uses it
(gloo rendezvous on 127.0.0.1, barrier, max-reduce of the elapsed time)."""
import os
import socket

import numpy as np
import pytest

from emfusion_amd import sharding

W, H = 96, 72


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    """Synthetic per-object images with overlaps and ties; same on every rank."""
    rng = np.random.default_rng(1234)
    nobj = 5
    ids = [1, 2, 3, 4, 5]
    hit = [(rng.random((H, W)) < 0.35).astype(np.uint8) for _ in ids]
    ray = [(rng.uniform(0.5, 3.0, (H, W)).astype(np.float32) * h) for h in hit]
    ray[3][:10] = ray[1][:10]  # ties between objects owned by different ranks
    hit[3][:10] = hit[1][:10]
    vert = [rng.standard_normal((H, W, 3)).astype(np.float32) for _ in ids]
    nrm = [rng.standard_normal((H, W, 3)).astype(np.float32) for _ in ids]
    bg_mask = (rng.random((H, W)) < 0.8).astype(np.uint8)
    bg_ray = rng.uniform(0.5, 3.0, (H, W)).astype(np.float32) * bg_mask
    bg_vert = rng.standard_normal((H, W, 3)).astype(np.float32)
    bg_norm = rng.standard_normal((H, W, 3)).astype(np.float32)
    assoc = [rng.uniform(0, 2, (H, W)).astype(np.float32) for _ in range(nobj + 1)]
    for a in assoc:
        a[:4] = 0
    return dict(ids=ids, hit=hit, ray=ray, vert=vert, nrm=nrm, bg_mask=bg_mask, bg_ray=bg_ray,
                bg_vert=bg_vert, bg_norm=bg_norm, assoc=assoc)


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = _scene()
        ids = sc["ids"]
        mine = sharding.local_objects(ids, rank, world)
        pos = {i: k for k, i in enumerate(ids)}

        # exchange 1: normaliser.  local partial of the object maps -> all-reduce(sum)
        partial = np.zeros((H, W), np.float32)
        for i in mine:
            partial = partial + sc["assoc"][pos[i] + 1]
        t = torch.from_numpy(partial.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        norm = sc["assoc"][0] + t.numpy()  # nsum = 1 (background) + extraSum
        with np.errstate(invalid="ignore", divide="ignore"):
            maps = {i: np.where(norm != 0, sc["assoc"][pos[i] + 1] / norm, 0).astype(np.float32)
                    for i in mine}
            bg_map = np.where(norm != 0, sc["assoc"][0] / norm, 0).astype(np.float32)

        # exchange 2: nearest-hit keys -> all-reduce(min) (int64 view: keys of hits are < 2^63)
        keys = sharding.pack_hit_keys([sc["ray"][pos[i]] for i in mine],
                                      [sc["hit"][pos[i]] for i in mine], [pos[i] for i in mine])
        if not mine:
            keys = np.full((H, W), sharding.NO_HIT, np.uint64)
        as_i64 = np.where(keys == sharding.NO_HIT, np.iinfo(np.int64).max, keys).astype(np.int64)
        kt = torch.from_numpy(as_i64)
        dist.all_reduce(kt, op=dist.ReduceOp.MIN)
        merged = kt.numpy()
        merged_u = np.where(merged == np.iinfo(np.int64).max, sharding.NO_HIT,
                            merged.astype(np.uint64))
        ray, win = sharding.unpack_hit_keys(merged_u)

        # the timing reduction bench.py performs
        el = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        out[rank] = dict(mine=mine, norm=norm, bg_map=bg_map, maps=maps, ray=ray, win=win,
                         elapsed=float(el.item()))
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_ranks():
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    return dict(out)


def test_ownership_is_a_partition():
    ids = list(range(1, 65))
    for world in (1, 2, 4, 8):
        parts = [sharding.local_objects(ids, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == ids
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert [sharding.owner_of(i, 8) for i in (1, 8, 9, 64)] == [0, 7, 0, 7]


def test_key_packing_orders_like_the_reference_rule():
    ray = [np.array([[2.0, 1.0, 1.0, 0.0]], np.float32), np.array([[1.0, 1.0, 3.0, 0.0]], np.float32)]
    hit = [np.array([[1, 1, 1, 0]], np.uint8), np.array([[1, 1, 0, 0]], np.uint8)]
    r, p = sharding.unpack_hit_keys(sharding.pack_hit_keys(ray, hit, [0, 1]))
    assert p.tolist() == [[1, 0, 0, -1]]       # nearer wins; ties keep the earlier object
    assert r.tolist() == [[1.0, 1.0, 1.0, 0.0]]


def test_both_ranks_finished_and_time_is_the_max(two_ranks):
    assert set(two_ranks) == {0, 1}
    assert two_ranks[0]["mine"] == [1, 3, 5] and two_ranks[1]["mine"] == [2, 4]
    assert two_ranks[0]["elapsed"] == two_ranks[1]["elapsed"] == pytest.approx(0.2)


def test_sharded_normaliser_matches_sequential_sum(two_ranks, oracle):
    sc = _scene()
    full = [a.copy() for a in sc["assoc"]]
    norm = oracle.normalize_association(full)  # reference order: ((bg + o1) + o2) + ...
    for r in (0, 1):
        got = two_ranks[r]
        # order of the float additions differs between the layouts: a few ulp at most
        assert np.allclose(got["norm"], norm, rtol=3e-7, atol=0)
        assert np.array_equal(got["norm"] == 0, norm == 0)
        assert np.allclose(got["bg_map"], full[0], rtol=1e-6, atol=1e-9)
        for i, m in got["maps"].items():
            assert np.allclose(m, full[i], rtol=1e-6, atol=1e-9)
    assert np.array_equal(two_ranks[0]["norm"], two_ranks[1]["norm"])  # every rank sees one norm


def test_sharded_composite_matches_list_order_rule(two_ranks, oracle):
    sc = _scene()
    diff = np.zeros((H, W), np.float32)
    ray, vert, nrm, seg, no_obj, vis = oracle.composite_raycast(
        sc["ids"], sc["ray"], sc["vert"], sc["nrm"], sc["hit"], sc["bg_ray"], sc["bg_vert"],
        sc["bg_norm"], sc["bg_mask"], diff, 5)
    # what the merged keys say before the background override
    pre_ray, pre_seg = np.zeros((H, W), np.float32), np.zeros((H, W), np.int32)
    for k, i in enumerate(sc["ids"]):
        take = (sc["hit"][k] != 0) & ((pre_ray <= 0) | (sc["ray"][k] < pre_ray))
        pre_ray = np.where(take, sc["ray"][k], pre_ray)
        pre_seg = np.where(take, i, pre_seg)
    for r in (0, 1):
        got = two_ranks[r]
        win_id = np.where(got["win"] >= 0, np.array(sc["ids"])[np.maximum(got["win"], 0)], 0)
        assert np.array_equal(win_id, pre_seg)
        assert np.array_equal(got["ray"], pre_ray)
        assert np.array_equal(got["ray"], ray)  # composite raylength is never replaced by bg
    assert (pre_seg[:10] == 2).any()  # tie rows: the earlier object (id 2) beats id 4


def test_background_bands_partition_the_rows():
    from emfusion_amd import sharding
    for height in (480, 120, 960, 16, 17):
        for world in (1, 2, 3, 4, 8):
            rows = []
            for rank in range(world):
                r0, n = sharding.bg_band(rank, world, height)
                assert r0 % 16 == 0 and n >= 0
                rows += list(range(r0, r0 + n))
            assert rows == list(range(height)), (height, world)
