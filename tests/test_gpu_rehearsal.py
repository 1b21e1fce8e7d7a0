"""The multi-GPU code path with world > 1 on ONE GPU: N emf::EMFusion instances on N host threads joined
by the in-process rehearsal communicator (RCCL refuses two ranks per device).  Every rank runs exactly
what it runs in an N-GPU job -- object ownership, its band of the background raycast, the per-frame
sequence of collectives (a mismatch would time out) -- and the joint result must equal the single-GPU
run: bit for bit where no float sum is re-ordered, to rounding in the association weights."""
import threading

import numpy as np
import pytest

from tests.parity_util import to_dev

pytestmark = pytest.mark.gpu

W, H = 160, 120
NFRAMES, MASK_EVERY = 6, 3


def run_job(world, nobj, depth_broadcast, track=False):
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    prm = pipeline.make_params(W, H, 64, 0.04, 32, visibility_thresh=100, boundary=5, mask_frames=MASK_EVERY)
    K = np.array(prm.K, np.float32)
    synth = pipeline.SyntheticStream(W, H, K, nobj, seed=0xE3F5)
    frames = []
    for f in range(NFRAMES):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        frames.append((depth, sid, R, t))
    spheres = [[synth.sphere(k, f)[0] for f in range(NFRAMES)] for k in range(nobj)]
    first = [synth.sphere(k, 0) for k in range(nobj)]
    synth.close()
    comms = pipeline.Communicator.local_group(world) if world > 1 else [None]
    out, errors = [None] * world, []
    ready = threading.Barrier(world)  # the ranks enter their first frame together (a bounded wait sits in the peer transport)

    def rank_main(r):
        try:
            fus = pipeline.Fusion(prm, comms[r])
            if depth_broadcast and world > 1:
                fus.set_depth_broadcast(0)
            ids = [fus.add_object(c, vs) for c, _, vs in first]
            mine = [i for i in ids if fus.owns_object(i)]
            keep = []
            ready.wait(timeout=120)
            for f, (depth, sid, R, t) in enumerate(frames):
                # with the broadcast on, only rank 0 holds the real depth: the others start from zeros
                d = to_dev(depth if (r == 0 or not depth_broadcast) else np.zeros_like(depth))
                poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), spheres[i - 1][f]) for i in mine}
                rm = f % MASK_EVERY == 0
                masks = {i: to_dev((sid == i).astype(np.uint8)) for i in mine} if rm else {}
                keep += [d, masks]
                if track and f == 1:
                    fus.set_tracking(camera=True, objects=True)  # frame 0 defines the world frame
                fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, rm)
            fus.synchronize()
            res = dict(mine=mine, seg=fus.image("segmentation"), ray=fus.image("raylengths"),
                       bg_ray=fus.image("bg_raylengths"), bg_assoc=fus.image("bg_assoc"),
                       bg_tsdf=fus.volume("tsdf", 0), bg_w=fus.volume("weights", 0),
                       vis=sorted(fus.visible_objects()), cam=fus.pose(0), poses={i: fus.pose(i) for i in mine},
                       obj={i: (fus.volume("tsdf", i), fus.volume("weights", i), fus.image("obj_assoc", i)) for i in mine})
            out[r] = res
            fus.close()
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "a rank hangs"
    assert not errors, errors
    for c in comms:
        if c is not None:
            c.close()
    return out


@pytest.mark.parametrize("world,depth_broadcast", [(2, False), (2, True), (4, True), (3, False)])
def test_sharded_job_on_threads_equals_single_gpu(dev, world, depth_broadcast):
    nobj = 4
    single = run_job(1, nobj, False)[0]
    ranks = run_job(world, nobj, depth_broadcast)
    assert sorted(i for r in ranks for i in r["mine"]) == list(range(1, nobj + 1))
    assert all(len(r["mine"]) >= 1 for r in ranks[:min(world, nobj)])
    def close(a, b, what):  # the sharded normaliser sums in another order: values move by ulps, and a
        # march decision may flip on a handful of pixels
        ok = np.isclose(a, b, rtol=1e-4, atol=1e-6)
        assert ok.mean() > 0.995, (what, ok.mean())

    r0 = ranks[0]
    for r in ranks:
        # the replicated background and the joint images are IDENTICAL on every rank (the all-reduce
        # hands every rank the same bits), so the replicas cannot drift apart
        for k in ("seg", "ray", "bg_ray", "bg_assoc", "bg_tsdf", "bg_w"):
            assert np.array_equal(r[k], r0[k]), k
        assert r["vis"] == r0["vis"] == single["vis"]
        # ... and equal to the single-GPU run up to the re-ordered float sum
        assert (r["seg"] == single["seg"]).mean() > 0.995
        close(r["bg_ray"], single["bg_ray"], "bg raylengths (bands gathered from all ranks)")
        close(r["ray"], single["ray"], "composited raylengths")
        close(r["bg_assoc"], single["bg_assoc"], "bg association")
        close(r["bg_tsdf"], single["bg_tsdf"], "bg tsdf")
        assert ((r["bg_w"] > 0) == (single["bg_w"] > 0)).mean() > 0.9999
        for i in r["mine"]:
            t, w, a = r["obj"][i]
            ts, ws, as_ = single["obj"][i]
            close(a, as_, f"association of object {i}")
            close(t, ts, f"tsdf of object {i}")
            assert ((w > 0) == (ws > 0)).mean() > 0.999
    assert (single["seg"] > 0).sum() > 200 and len(single["vis"]) >= 2


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_tracking_keeps_the_ranks_in_step(dev, world):
    """With tracking on, every rank tracks the camera against ITS replica of the background and the
    objects it owns: the camera poses must come out bit-identical on all ranks (same inputs, same
    deterministic kernels) -- otherwise the replicas would drift -- and agree with the single-GPU run."""
    nobj = 4
    single = run_job(1, nobj, False, track=True)[0]
    ranks = run_job(world, nobj, True, track=True)
    R0, t0 = ranks[0]["cam"]
    for r in ranks:
        R, t = r["cam"]
        assert np.array_equal(R, R0) and np.array_equal(t, t0)
        assert np.array_equal(r["bg_tsdf"], ranks[0]["bg_tsdf"]) and np.array_equal(r["seg"], ranks[0]["seg"])
        for i in r["mine"]:
            Ro, to = r["poses"][i]
            Rs, ts = single["poses"][i]
            assert np.allclose(to, ts, atol=2e-3) and np.allclose(Ro, Rs, atol=2e-3), i
    Rs, ts = single["cam"]
    assert np.allclose(t0, ts, atol=1e-3) and np.allclose(R0, Rs, atol=1e-3)
    assert np.linalg.norm(ts) > 1e-3  # the camera did move
