"""EMFusion::processFrame(const RGBD&) hands the host depth map to the device through two pinned staging buffers, a copy
stream and two device images (round 6, `EMFusion::stageDepth`; reference EMFusion.cpp:72 + the reader thread of
RGBDReader.cpp:72-117): a slot is reused by the frame two calls later, the frame's stream waits for the copy's event, the
copy stream for the end of the frame that last read the slot.  A hazard in that hand-over would show as a frame fusing
another frame's depth -- so the same sequence runs with the synchronous pageable upload of rounds 1-5 (EMF_ASYNC_UPLOAD=0)
and must give the same bytes; the caller's buffer is overwritten right after every call (it may be reused at once)."""
import os

import numpy as np
import pytest
import xxhash

pytestmark = pytest.mark.gpu
W, H, FRAMES = 320, 240, 14


def _digest(a):
    return xxhash.xxh3_128(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()


def _run(env, sync_every):
    from emfusion_amd import pipeline
    for k, v in env.items():
        os.environ[k] = v
    try:
        prm = pipeline.make_params(W, H, 128, 0.04, 32, visibility_thresh=100, boundary=5, mask_frames=100)
        synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), 2, seed=0xE3F5)
        fus = pipeline.Fusion(prm, None)
        fus.set_tracking(camera=True, objects=False)
        buf = np.empty((H, W), np.float32)  # ONE caller buffer, refilled for every frame and poisoned behind the call
        poses = []
        for f in range(FRAMES):
            buf[:] = synth.render(f)[0]
            fus.process_rgbd(buf)
            buf[:] = 123.0
            if sync_every and f % sync_every == 0:
                fus.synchronize()
            poses.append(np.concatenate([np.asarray(p, np.float64).reshape(-1) for p in fus.pose(0)]))
        fus.synchronize()
        out = dict(tsdf=_digest(fus.volume("tsdf", 0)), weights=_digest(fus.volume("weights", 0)),
                   ray=_digest(fus.image("bg_raylengths")), assoc=_digest(fus.image("bg_assoc")), poses=np.array(poses),
                   upload=fus.upload_host_time(), seen=int((fus.volume("weights", 0) > 0).sum()))
        fus.close()
        synth.close()
        return out
    finally:
        for k in env:
            os.environ.pop(k, None)


@pytest.mark.parametrize("sync_every", [0, 1, 3], ids=["never_synchronised", "every_frame", "every_third_frame"])
def test_double_buffered_upload_gives_the_bytes_of_the_synchronous_one(dev, sync_every):
    a = _run({}, sync_every)
    b = _run({"EMF_ASYNC_UPLOAD": "0"}, sync_every)
    assert a["seen"] > 10000 and a["upload"][1] == FRAMES and b["upload"][1] == FRAMES
    for k in ("tsdf", "weights", "ray", "assoc"):
        assert a[k] == b[k], k
    assert np.array_equal(a["poses"], b["poses"])
    assert np.abs(a["poses"][-1][9:]).max() < 0.2 and np.abs(np.diff(a["poses"][:, 9:], axis=0)).max() > 0  # the camera moved and was followed
