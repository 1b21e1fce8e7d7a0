"""Host logic of bench_extras.py on CPU: what the N > 1 bench line reports about the transport is gathered over a real
world-size-2 gloo group (as bench.py does it), with stand-in communicators whose describe() says what a node's RCCL would
(two distinct devices) and what the one-GPU rehearsal's does (one device twice); the strong-scaling partition; the staged
TUM-layout stream bench.py --entry / --track read; the summary arithmetic of the tracked lines."""
import json
import os
import pickle
import socket

import numpy as np
import pytest

import bench_extras as X


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeComm:
    def __init__(self, rank, world, bus, transport="rccl", version=22707):
        self.d = dict(transport=transport, ranks=world, rank=rank, device=rank if bus != "same" else 0,
                      pci_bus_id="0000:%02x:00.0" % (0x23 + (rank if bus != "same" else 0)), version=version)

    def describe(self):
        return dict(self.d)


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        node = X.transport_report(_FakeComm(rank, world, "distinct"), dist, world, "rccl")
        shared = X.transport_report(_FakeComm(rank, world, "same", transport="peer-write"), dist, world, "peer")
        liar = _FakeComm(rank, world, "distinct")
        if rank == 1:
            liar.d["ranks"] = 1  # a transport that came up with fewer ranks than the launcher started
        broken = X.transport_report(liar, dist, world, "rccl")
        out[rank] = json.dumps(dict(node=node, shared=shared, broken=broken))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def reports():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    return {r: json.loads(v) for r, v in dict(out).items()}


def test_transport_report_is_the_same_on_every_rank_and_names_the_devices(reports):
    assert set(reports) == {0, 1} and reports[0] == reports[1]
    node = reports[0]["node"]
    assert node["ranks"] == 2 and node["ranks_agree"] and node["one_device_per_rank"] and node["distinct_devices"] == 2
    assert [d["rank"] for d in node["devices"]] == [0, 1] and [d["device"] for d in node["devices"]] == [0, 1]
    assert node["devices"][0]["pci_bus_id"] != node["devices"][1]["pci_bus_id"] and node["version"] == 22707
    assert len({d["pid"] for d in node["devices"]}) == 2  # one process per rank


def test_ranks_sharing_a_device_and_a_short_communicator_are_visible_in_the_line(reports):
    shared = reports[0]["shared"]
    assert shared["ranks_agree"] and not shared["one_device_per_rank"] and shared["distinct_devices"] == 1
    assert shared["transport"] == "peer-write" and shared["requested"] == "peer"
    assert not reports[0]["broken"]["ranks_agree"]


def test_strong_scaling_partition_is_round_robin_and_balanced():
    from emfusion_amd import sharding
    ids = list(range(1, 65))
    for world in (1, 2, 4, 8):
        per = [len(sharding.local_objects(ids, r, world)) for r in range(world)]
        assert sum(per) == 64 and max(per) - min(per) <= 1
        assert max(per) + 1 <= 33 or world == 1  # background + objects of a rank: at most two chunks of the table from N = 2 on


def test_tracked_line_summaries():
    track = [(20, 15, 30, 1), (100, 60, 10, 1), (10, 9, 100, 2), (5, 5, 5, 2)]
    s = X._tracking_steps(track, 100)
    assert s["camera_per_frame"] == 33.8 and s["camera_accepted_per_frame"] == 22.2 and s["objects_longest_per_frame"] == 36.2
    assert s["frames_in_which_a_stage_used_the_whole_budget"] == 2 and s["live_objects"] == [1, 2]
    a = np.ones((4, 4), np.float32)
    b = a.copy()
    b[0, 0] = 1.001
    assert X._outside(a, b) == pytest.approx(1 / 16) and X._outside(a, a) == 0.0 and X._outside(a, b, rtol=1e-2) == 0.0


def test_the_staged_stream_is_what_the_readers_take():
    from emfusion_amd import readers
    st = X.TumStream(2)
    try:
        assert len(st.depth) == 2 and st.depth[0].shape == (480, 640) and st.depth[0].dtype == np.float32
        q = st.depth[0] * 5000.0
        assert np.abs(q - np.round(q)).max() < 1e-2 and (st.depth[0] == 0).mean() > 0.005  # PNG quantisation, drop-outs
        assert sorted(os.listdir(st.masks_dir)) == ["Mask0000.plk"]  # masks every 30 frames
        with open(st.masks_dir / "Mask0000.plk", "rb") as fh:
            boxes, masks, scores = pickle.load(fh)
        assert len(masks) == 1 and masks[0].shape == (480, 640) and masks[0].sum() > 2000
        assert int(np.argmax(scores[0])) == st.S.PERSON_CLASS and np.array_equal(masks[0].astype(np.uint8), st.masks[0])
        pm = readers.load_preprocessed_masks(st.masks_dir / "Mask0000.plk") if hasattr(readers, "load_preprocessed_masks") else None
        assert pm is None or len(pm[1]) == 1
        assert X.TUM_CFG.exists()
    finally:
        st.close()
