"""The reference's per-frame debug images (SURVEY.md 8 f-4 remainder): with setupOutput() on, every frame keeps
the association weights before and after tracking (storeAssocs, reference EMFusion.cpp:79-91, 307-320), the Huber
and combined tracking weights of the stages that ran (EMFusion.cpp:110-118; TSDF.cpp:346-354), the objects'
foreground-probability look-ups (ObjTSDF.cpp:237-240) and the renderings (EMFusion.cpp:158-160), each as
cv::Mat::convertTo(CV_8U, 255), and writeResults() writes them as <dir>/{output, assoc_weights/<bg|id>/<pre|post>Track,
huber_weights/<bg|id>, track_weights/<bg|id>, fg_probs/<id>}/%04d.png (EMFusion.cpp:1009-1145, 1256-1261)."""
import numpy as np
import pytest

from tests.parity_util import dev_full, to_dev, to_np

pytestmark = pytest.mark.gpu
W, H = 320, 240


@pytest.fixture(scope="module")
def ops(dev):
    from emfusion_amd import ops as _ops
    return _ops


def _u8(img):
    """convertTo(CV_8U, 255): round half to even, saturate"""
    v = img.astype(np.float32) * np.float32(255)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def _run(tmp_path, track, frames=4):
    from emfusion_amd import pipeline, readers
    from emfusion_amd.ops import image_view
    prm = pipeline.make_params(W, H, 128, 0.04, 32, visibility_thresh=400, boundary=10)
    synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), 2, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    fus.setup_output(False, False)
    if track:
        fus.set_tracking(True, True)
    centers, keep, per_frame = {}, [], {}
    for f in range(frames):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        d = to_dev(depth)
        masks = {i: to_dev((sid == i).astype(np.uint8)) for i in centers}
        keep += [d, masks]
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), c) for i, c in centers.items()}
        if f == 0:
            new = [to_dev((sid == k).astype(np.uint8)) for k in (1, 2)]
            keep.append(new)
            fus.queue_new_object_masks([image_view(m) for m in new])
        fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, f == 0)
        fus.synchronize()
        if f == 0:
            centers = {k: fus.pose(k)[1] for k in (1, 2)}
        else:
            per_frame[f] = dict(bg=fus.image("bg_assoc"), obj={k: fus.image("obj_assoc", k) for k in (1, 2)},
                                points=fus.image("points"), fg={k: fus.volume("fgprobs", k) for k in (1, 2)},
                                pose={k: fus.pose(k) for k in (0, 1, 2)}, render=fus.render()[0])
    fus.write_results(str(tmp_path), volumes=True)
    fus.close(); synth.close()
    # voxel size of the mask-made object volumes: header of the reference's volume dump (int32 x 3, u64, f32)
    vox = {k: float(np.frombuffer((tmp_path / "tsdfs" / f"tsdf_{k}.bin").read_bytes()[20:24], np.float32)[0]) for k in (1, 2)}
    return per_frame, readers, vox


def test_image_log_with_supplied_poses(dev, ops, tmp_path):
    per_frame, readers, VOX = _run(tmp_path, track=False)
    assert sorted(per_frame) == [1, 2, 3]
    for f, rec in per_frame.items():
        name = "%04d.png" % f
        # poses supplied: the frame's last E-step runs at the frame's poses, so postTrack == the frame's maps; the
        # first one still sees the previous frame's poses (the schedule of EMFusion.cpp:79-91)
        got = readers.read_png_gray(tmp_path / "assoc_weights" / "bg" / "postTrack" / name)
        assert got.dtype == np.uint8 and np.array_equal(got, _u8(rec["bg"])), f
        pre = readers.read_png_gray(tmp_path / "assoc_weights" / "bg" / "preTrack" / name)
        assert pre.shape == got.shape and np.abs(pre.astype(int) - got.astype(int)).mean() < 8
        for k in (1, 2):
            got = readers.read_png_gray(tmp_path / "assoc_weights" / str(k) / "postTrack" / name)
            assert np.array_equal(got, _u8(rec["obj"][k])), (f, k)
            pre = readers.read_png_gray(tmp_path / "assoc_weights" / str(k) / "preTrack" / name)
            assert pre.shape == got.shape and (pre > 0).any()
        assert 0 < (_u8(rec["obj"][1]) > 0).sum() < W * H
        # fg_probs/<id>: getVolumeVals(fgProbs, points, rel_pose_CO) of the frame's E-step (no mask frame after 0: the
        # volume read back after the frame is the one the E-step saw)
        Rc, tc = rec["pose"][0]
        for k in (1, 2):
            Ro, to = rec["pose"][k]
            Rco = (Ro.reshape(3, 3).T @ Rc.reshape(3, 3)).astype(np.float32)
            tco = (Ro.reshape(3, 3).T @ (tc - to)).astype(np.float32)
            vals = dev_full((H, W), 0.0)
            ops.get_volume_vals(to_dev(rec["fg"][k]), to_dev(rec["points"]), Rco.reshape(-1), tco, VOX[k], vals)
            got = readers.read_png_gray(tmp_path / "fg_probs" / str(k) / name)
            assert got.shape == (H, W) and got.dtype == np.uint8
            want = _u8(to_np(vals))
            assert (got != want).mean() < 2e-3, (f, k, int((got != want).sum()))  # (pose product re-formed on the host here)
            seen = got[rec["obj"][k] > 0.5]
            assert seen.size > 50 and (seen > 0).mean() > 0.5, (f, k)  # where the object explains the pixel it is foreground
        # no tracking stage ran: nothing in huber_weights / track_weights, but the directories exist (the reference
        # creates them unconditionally)
        assert (tmp_path / "huber_weights" / "bg").is_dir() and not list((tmp_path / "huber_weights" / "bg").iterdir())
        assert (tmp_path / "track_weights" / "bg").is_dir()
    # renderings: render() of frame f is kept under frameCount - 1 = f
    for f, rec in per_frame.items():
        raw = (tmp_path / "output" / ("%04d.png" % f)).read_bytes()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n" and raw[25] == 2  # colour type: truecolour
        got = _png_rgb(raw)
        assert np.array_equal(got, rec["render"])


def _png_rgb(raw):
    import struct
    import zlib
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(raw):
        n, kind = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert zlib.crc32(kind + body) == struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0]
        if kind == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
            assert body[8:] == bytes([8, 2, 0, 0, 0])
        elif kind == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not rows[:, 0].any()  # filter type None
    return rows[:, 1:].reshape(h, w, 3)


def test_image_log_with_tracking(dev, tmp_path):
    per_frame, readers, _ = _run(tmp_path, track=True)
    for f in per_frame:
        name = "%04d.png" % f
        for who in ("bg", "1", "2"):
            hub = readers.read_png_gray(tmp_path / "huber_weights" / who / name)
            trk = readers.read_png_gray(tmp_path / "track_weights" / who / name)
            assert hub.shape == trk.shape == (H, W) and hub.dtype == np.uint8
            # combined = Huber x normalised integration weight (<= 1) x association weight (<= 1)
            assert (trk.astype(int) <= hub.astype(int) + 1).all(), (f, who)
            assert (hub > 0).sum() > 200, (f, who)
        hub = readers.read_png_gray(tmp_path / "huber_weights" / "bg" / name)
        assert (hub == 255).mean() > 0.5  # residuals under the threshold almost everywhere for a tracked background
        pre = readers.read_png_gray(tmp_path / "assoc_weights" / "1" / "preTrack" / name)
        post = readers.read_png_gray(tmp_path / "assoc_weights" / "1" / "postTrack" / name)
        assert pre.shape == post.shape == (H, W) and (post > 0).any()
