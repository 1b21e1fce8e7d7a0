"""Object creation / matching primitives (SURVEY f-3): masked radix-select point statistics against
sorting (reference filterPoints + transformPoints + computePercentiles, EMFusion.cu:63-98), and the
all-objects overlap counts against the compare / and / or / countNonZero chain of
EMFusion::matchSegmentation (EMFusion.cpp:797-825)."""
import numpy as np
import pytest

from tests.parity_util import dev_full, to_dev
from tests.scenes import Pose, camera_path, intrinsics, render_depth, rot

pytestmark = pytest.mark.gpu
W, H = 160, 120
K = intrinsics(W, H)
SPHERES = [((0.25, 0.05, 1.3), 0.22), ((-0.3, -0.1, 1.6), 0.18)]


@pytest.fixture(scope="module")
def ops(dev):
    from emfusion_amd import ops as _ops
    return _ops


def sorted_stats(points, mask, R, t):
    """The reference's way: compact, transform, sort each channel, take two columns."""
    f32 = np.float32
    valid = (mask != 0) & np.any(points != 0, axis=2)
    p = points[valid].astype(f32)
    n = len(p)
    if n == 0:
        return 0, np.zeros(3, f32), np.zeros(3, f32)
    R = np.asarray(R, f32).reshape(3, 3)
    q = np.empty_like(p)
    for i in range(3):  # (r0 x + r1 y) + r2 z, then + t: the product's operation order
        q[:, i] = f32(f32(f32(R[i, 0] * p[:, 0]) + f32(R[i, 1] * p[:, 1])) + f32(R[i, 2] * p[:, 2])) + f32(t[i])
    s = np.sort(q, axis=0)
    return n, s[int(f32(n) * f32(.1))], s[int(f32(n) * f32(.9))]


@pytest.mark.parametrize("case", ["sphere", "rotated", "empty", "single", "negative", "padded"])
def test_masked_point_stats_equal_sorting(oracle, ops, dev, case):
    cam = camera_path(3)
    depth, ids = render_depth(W, H, K, cam, SPHERES, noise=0.003, dropout=0.02, seed=11)
    points = oracle.compute_points(depth, K)
    mask = (ids == 1).astype(np.uint8)
    R, t = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    pad = 0
    if case == "rotated":
        R, t = rot([0.3, 1, 0.1], 25).astype(np.float32), np.array([0.4, -2.0, 0.3], np.float32)
    elif case == "empty":
        mask[:] = 0
    elif case == "single":
        mask[:] = 0
        mask[60, 80] = 7
    elif case == "negative":
        mask = ((ids > 0) * 255).astype(np.uint8)
        t = np.array([-0.1, 0.0, -1.45], np.float32)  # coordinates of both signs, zeros nearby
    elif case == "padded":
        pad = 5
        mask = np.ones((H, W), np.uint8)  # everything valid: the dropout pixels must be skipped
    n, p10, p90 = sorted_stats(points, mask, R, t)
    got_n, g10, g90 = ops.masked_point_stats(to_dev(points, dev, pad), to_dev(mask, dev, pad), R, t)
    assert got_n == n
    if case in ("sphere", "rotated", "negative", "padded"):
        assert n > 500 and np.all(p90 > p10)
    assert g10.tobytes() == p10.astype(np.float32).tobytes(), (g10, p10)
    assert g90.tobytes() == p90.astype(np.float32).tobytes(), (g90, p90)


def test_mask_overlap_counts(ops, dev):
    rng = np.random.default_rng(5)
    model = np.zeros((H, W), np.uint8)
    model[20:70, 30:90] = 1
    model[50:110, 70:150] = 2
    model[0:5, 0:5] = 255
    seg = np.zeros((H, W), np.uint8)
    seg[40:100, 60:120] = 1
    seg[rng.uniform(size=(H, W)) < 0.02] = 200
    n, inter, area = ops.mask_overlap(to_dev(seg, dev, 3), to_dev(model, dev))
    assert n == int((seg != 0).sum())
    for i in (1, 2, 255, 9):
        assert inter[i] == int(((seg != 0) & (model == i)).sum()), i
        assert area[i] == int((model == i).sum()), i
    iou1 = inter[1] / (n + area[1] - inter[1])
    want = ((seg != 0) & (model == 1)).sum() / ((seg != 0) | (model == 1)).sum()
    assert abs(iou1 - want) < 1e-12


# ---- host logic on top: EMFusion::initNewObjVolume / volumeIOU / matchSegmentation ----------------

def volume_iou(low, high, prev_size, p10, p90, vol_pad=2.0):
    """EMFusion::volumeIOU (EMFusion.cpp:559-611) in float32."""
    f32 = np.float32
    center = (p10 + p90) / f32(2)
    vs = f32(vol_pad) * (p90 - p10).max()
    low_n, high_n = center - vs / f32(2), center + vs / f32(2)
    d = np.minimum(high, high_n) - np.maximum(low, low_n)
    if (d < 0).any():
        return 0.0
    vi = f32(np.prod(d.astype(f32)))
    return float(vi / (f32(vs) ** 3 + f32(np.prod(prev_size)) - vi))


def test_objects_are_created_and_matched_from_masks(oracle, dev):
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    Wf, Hf = 320, 240
    prm = pipeline.make_params(Wf, Hf, 128, 0.04, 32, visibility_thresh=400, boundary=10)
    Kf = np.array(prm.K, np.float32)
    synth = pipeline.SyntheticStream(Wf, Hf, Kf, 2, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    keep, centers = [], {}
    for f in range(4):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        d = to_dev(depth)
        masks = {i: to_dev((sid == i).astype(np.uint8)) for i in centers}
        keep += [d, masks]
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), c) for i, c in centers.items()}
        if f == 0:  # both instance masks are "unmatched": created inside the frame, like the reference
            new = [to_dev((sid == k).astype(np.uint8)) for k in (1, 2)]
            tiny = np.zeros((Hf, Wf), np.uint8)
            tiny[100:110, 100:110] = 1
            new.append(to_dev(tiny))  # fewer than visibilityThresh valid points
            new.append(new[0])        # the first mask again: the volume would coincide (volume IoU)
            keep.append(new)
            fus.queue_new_object_masks([image_view(m) for m in new])
        fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, True)
        fus.synchronize()
        if f == 0:
            assert fus.last_created() == [1, 2, -1, -1]
            pts = oracle.compute_points(depth, Kf)
            for k in (1, 2):
                n, p10, p90 = sorted_stats(pts, (sid == k).astype(np.uint8), R.reshape(3, 3), t)
                Ro, to = fus.pose(k)
                assert n >= 400 and np.allclose(Ro, np.eye(3))
                assert np.allclose(to, (p10 + p90) / np.float32(2), atol=1e-6)
                centers[k] = to
            assert fus.create_object_from_mask(image_view(new[1])) == -1  # also between frames
    # after a few frames the raycast segmentation shows the objects: masks match them
    seg = fus.image("segmentation")
    assert set(np.unique(seg)) >= {0, 1, 2}
    depth, sid = synth.render(3)
    for k in (1, 2):
        m = (sid == k).astype(np.uint8)
        got_id, iou = fus.match_mask(image_view(to_dev(m)))
        want = ((m != 0) & (seg == k)).sum() / ((m != 0) | (seg == k)).sum()
        assert got_id == k and abs(iou - want) < 1e-6 and iou > 0.2
    empty = np.zeros((Hf, Wf), np.uint8)
    empty[0:3, 0:3] = 1
    assert fus.match_mask(image_view(to_dev(empty)))[0] == -1
    assert volume_iou(np.full(3, -1, np.float32), np.full(3, 1, np.float32), np.full(3, 2, np.float32),
                      np.full(3, -.5, np.float32), np.full(3, .5, np.float32)) == pytest.approx(1.0)
    fus.close()
    synth.close()


def test_mask_association_mass(ops, dev):
    rng = np.random.default_rng(8)
    seg = (rng.uniform(size=(H, W)) < 0.2).astype(np.uint8)
    match = (rng.uniform(size=(H, W)) < 0.1).astype(np.uint8) * 255
    assoc = rng.uniform(0, 1, (H, W)).astype(np.float32)
    for m in (None, match):
        inside = (seg != 0) if m is None else ((seg != 0) | (m != 0))
        n, total = ops.mask_association_mass(to_dev(seg, dev, 2), None if m is None else to_dev(m, dev),
                                             to_dev(assoc, dev))
        assert n == int(inside.sum())
        assert abs(total - float(assoc[inside].astype(np.float64).sum())) < 1e-9 * n


@pytest.mark.parametrize("exp_vols", [False, True])
def test_invisible_objects_are_cleaned_up(oracle, dev, tmp_path, exp_vols):
    """cleanUpObjs: an object that the raycast no longer sees is deleted (EMFusion.cpp:951-976).  With
    the log on its last mesh is kept, and with setupOutput's exp_vols its volumes too (EMFusion.cpp:
    962-973); writeResults writes the meshes in either case and tsdfs/ only with exp_vols
    (EMFusion.cpp:273-291)."""
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    Wf, Hf = 320, 240
    prm = pipeline.make_params(Wf, Hf, 128, 0.04, 32, visibility_thresh=400, boundary=10)
    synth = pipeline.SyntheticStream(Wf, Hf, np.array(prm.K, np.float32), 2, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    fus.set_cleanup(True)
    fus.setup_output(False, exp_vols)
    last_tsdf2 = None
    centers, keep = {}, []
    for f in range(5):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        d = to_dev(depth)
        masks = {i: to_dev((sid == i).astype(np.uint8)) for i in centers}
        keep += [d, masks]
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), c) for i, c in centers.items()}
        if f == 3:  # object 2 is reported far behind the camera: the raycast cannot see it any more
            poses[2] = (poses[2][0], np.array([0, 0, -30], np.float32))
        if f == 0:
            new = [to_dev((sid == k).astype(np.uint8)) for k in (1, 2)]
            keep.append(new)
            fus.queue_new_object_masks([image_view(m) for m in new])
        if f == 3:
            last_tsdf2 = fus.volume("tsdf", 2)
        fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, True)
        fus.synchronize()
        if f == 0:
            assert fus.last_created() == [1, 2] and fus.last_deleted() == []
            centers = {k: fus.pose(k)[1] for k in (1, 2)}
        elif f < 3:
            assert fus.last_deleted() == [], f
        elif f == 3:
            assert fus.last_deleted() == [2]
            del centers[2]
        else:
            assert fus.last_deleted() == [] and sorted(fus.visible_objects()) == [1]
    fus.write_results(tmp_path, volumes=False)
    for name in ("poses-cam.txt", "poses-1.txt", "poses-2.txt", "mesh_bg.ply", "mesh_1.ply", "mesh_2.ply"):
        assert (tmp_path / name).stat().st_size > 0, name
    if exp_vols:
        import struct
        names = sorted(p.name for p in (tmp_path / "tsdfs").iterdir())
        assert names == sorted(["bg_tsdf.bin"] + [f"{k}_{i}.bin" for k in ("tsdf", "weights", "fgProbs") for i in (1, 2)])
        raw = (tmp_path / "tsdfs" / "tsdf_2.bin").read_bytes()
        assert struct.unpack_from("<3i", raw, 0) == (32, 32, 32)
        # the deleted object's volume as it was when cleanUpObjs dropped it: frame 3 integrated nothing
        # into it (invisible), so it is the volume read back before that frame
        assert np.array_equal(np.frombuffer(raw, np.float32, offset=24).reshape(32, 32, 32), last_tsdf2)
    else:
        assert not (tmp_path / "tsdfs").exists()
    fus.close()
    synth.close()


def test_model_table_slots_are_reused_after_clean_up(oracle, dev):
    """EMF_MAX_MODELS bounds the LIVE objects, not the number ever created: a long run that keeps
    spawning spurious objects and cleaning them up (the normal flow on TUM walking sequences) goes
    past 255 creations without trouble."""
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    Wf, Hf = 80, 60
    prm = pipeline.make_params(Wf, Hf, 32, 0.08, 8, visibility_thresh=50, boundary=2)
    synth = pipeline.SyntheticStream(Wf, Hf, np.array(prm.K, np.float32), 1, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    fus.set_cleanup(True)
    depth, _ = synth.render(0)
    R, t = synth.camera_pose(0)
    d = to_dev(depth)
    eye = np.eye(3, dtype=np.float32).reshape(-1)
    behind = np.array([0, 0, -30], np.float32)
    created = 0
    for f in range(270):
        oid = fus.add_object(behind, 0.5)  # never visible: cleaned up at the end of the frame
        created += 1
        fus.process_frame(image_view(d), R, t, {oid: (eye, behind)}, {}, False)
        fus.synchronize()
        # frame 0 has no raycast (EMFusion.cpp:78-95), so its object lives one frame longer
        assert fus.last_deleted() == ([] if f == 0 else [1, 2] if f == 1 else [oid]), f
    assert created == 270 and fus.object_ids() == []
    fus.close()
    synth.close()


def test_carve_mask(ops, dev):
    rng = np.random.default_rng(2)
    seg = (rng.uniform(size=(H, W)) < 0.5).astype(np.uint8) * 3
    model = rng.integers(0, 4, (H, W)).astype(np.uint8)
    match = (rng.uniform(size=(H, W)) < 0.2).astype(np.uint8)
    for m in (None, match):
        d_seg = to_dev(seg, dev, 1)
        taken = (model == 2) if m is None else ((model == 2) | (m != 0))
        pre, post = ops.carve_mask(d_seg, to_dev(model, dev), 2, None if m is None else to_dev(m, dev))
        want = np.where(taken, 0, seg)
        assert pre == int((seg != 0).sum()) and post == int((want != 0).sum())
        assert np.array_equal(d_seg.numpy(), want)


def test_instance_masks_run_the_reference_control_flow(oracle, dev):
    """initOrMatchObjs inside the frame: spawn on frame 0, match afterwards, a second mask on the
    same model goes through the unmatched path, is carved to (almost) nothing and spawns nothing.
    Frame 3 pins quirk Q20 (EMFusion.cpp:424-437, 462-478): a second mask that matches BETTER replaces
    the first in matches[] but still continues as unmatched; matches[id] being a shallow copy of it, it
    is carved against itself, so the model ends up matched to an all-zero mask."""
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    Wf, Hf = 320, 240
    prm = pipeline.make_params(Wf, Hf, 128, 0.04, 32, visibility_thresh=400, boundary=10)
    synth = pipeline.SyntheticStream(Wf, Hf, np.array(prm.K, np.float32), 2, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    fus.set_cleanup(True)
    centers, keep = {}, []
    for f in range(4):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        d = to_dev(depth)
        inst = [to_dev((sid == k).astype(np.uint8)) for k in (1, 2)]
        if f == 2:  # Mask R-CNN reports sphere 1 twice: a slightly eroded duplicate
            dup = (sid == 1).astype(np.uint8)
            dup[:, : Wf // 2 - 10] = 0
            inst.append(to_dev(dup))
        if f == 3:  # ... and once more with the worse mask first: the full one replaces it
            full = (sid == 1).astype(np.uint8)
            rows = np.flatnonzero(full.any(axis=1))
            worse = full.copy()
            worse[: rows[0] + len(rows) // 4] = 0  # top quarter missing: IoU ~ 0.8 < IoU of the full mask
            inst = [to_dev(worse), inst[1], to_dev(full)]
        keep += [d, inst]
        fus.queue_instance_masks([image_view(m) for m in inst])
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), c) for i, c in centers.items()}
        fus.process_frame(image_view(d), R, t, poses, {}, False)
        fus.synchronize()
        a = fus.last_mask_assignment()
        if f == 0:
            assert a == [1, 2] and fus.last_created() == [1, 2]
            centers = {k: fus.pose(k)[1] for k in (1, 2)}
        elif f == 2:
            assert a[:2] == [1, 2] and a[2] == -1 and fus.last_created() == [-1]
            assert inst[2].numpy().sum() < 0.5 * ((sid == 1).sum())  # carved in place
        elif f == 3:
            assert a == [-1, 2, 1] and fus.last_created() == [-1], (a, fus.last_created())
            assert inst[2].numpy().sum() == 0           # carved against its own alias
            assert inst[0].numpy().sum() == worse.sum()  # the replaced mask is simply dropped
        else:
            assert a == [1, 2] and fus.last_created() == []
        assert fus.last_deleted() == []
    assert sorted(fus.visible_objects()) == [1, 2]
    assert (fus.volume("fgmask", 1) > 0).sum() > 50  # the matched masks were integrated
    fus.close()
    synth.close()


# ---- updateObj / resize: iso-surface vertex cloud + points, copyValues ------------------------------

def mesh_cloud(tsdf, weights, fg, voxel):
    """Vertex cloud of the reference's marching cubes (TSDF.cu:855-1152, ObjTSDF.cpp:247-268) in numpy:
    one vertex per sign-changing edge of every cube whose 8 voxels pass the mask, vertexInterp."""
    f32 = np.float32
    nz, ny, nx = tsdf.shape
    ok = weights > 0 if fg is None else (weights > 0) & (fg != 0)
    corner = [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)]  # dx, dy, dz
    sub = lambda a, c: a[c[2]:nz - 1 + c[2], c[1]:ny - 1 + c[1], c[0]:nx - 1 + c[0]]
    valid = np.ones((nz - 1, ny - 1, nx - 1), bool)
    for c in corner:
        valid &= sub(ok, c)
    zz, yy, xx = np.meshgrid(np.arange(nz - 1), np.arange(ny - 1), np.arange(nx - 1), indexing="ij")
    half = [f32(n - 1) / f32(2) for n in (nx, ny, nz)]
    pos = lambda c: np.stack([(f32(1) * (xx + c[0]).astype(f32) - half[0]) * f32(voxel),
                              ((yy + c[1]).astype(f32) - half[1]) * f32(voxel),
                              ((zz + c[2]).astype(f32) - half[2]) * f32(voxel)], -1).astype(f32)
    out = []
    for a, b in [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]:
        v1, v2 = sub(tsdf, corner[a]), sub(tsdf, corner[b])
        sel = valid & ((v1 < 0) != (v2 < 0))
        p1, p2, v1, v2 = pos(corner[a])[sel], pos(corner[b])[sel], v1[sel], v2[sel]
        mu = (-v1 / (v2 - v1)).astype(f32)
        p = (p1 + (mu[:, None] * (p2 - p1)).astype(f32)).astype(f32)
        use1 = (np.abs(v1).astype(np.float64) < 1e-5)
        use2 = ~use1 & (np.abs(v2).astype(np.float64) < 1e-5)
        use3 = ~use1 & ~use2 & (np.abs(v1 - v2).astype(np.float64) < 1e-5)
        p[use1 | use3] = p1[use1 | use3]
        p[use2] = p2[use2]
        out.append(p)
    return np.concatenate(out) if out else np.zeros((0, 3), f32)


def test_object_extent_stats_equal_mesh_cloud_plus_points(oracle, ops, dev):
    from tests.scenes import rel_CO, rel_OC
    f32 = np.float32
    n, vox, pose = (32, 32, 32), 0.02, Pose(t=SPHERES[0][0])
    tsdf, wts = np.zeros((32, 32, 32), f32), np.zeros((32, 32, 32), f32)
    for i in range(3):
        cam = camera_path(i)
        depth, _ = render_depth(W, H, K, cam, SPHERES, noise=0.002, dropout=0.01, seed=70 + i)
        oc = rel_OC(cam, pose)
        oracle.update_tsdf(depth, np.ones((H, W), f32), tsdf, wts, oc.R32, oc.t32, K, vox, 10 * vox, 64.0)
    rng = np.random.default_rng(4)
    fg = (rng.uniform(size=tsdf.shape) < 0.9).astype(np.uint8) * 255
    cam = camera_path(3)
    depth, ids = render_depth(W, H, K, cam, SPHERES, noise=0.003, dropout=0.02, seed=74)
    points = oracle.compute_points(depth, K)
    mask = (ids == 1).astype(np.uint8)
    co = rel_CO(cam, pose)
    for fgm in (None, fg):
        cloud = mesh_cloud(tsdf, wts, fgm, vox)
        assert len(cloud) > 300
        valid = (mask != 0) & np.any(points != 0, axis=2)
        R = co.R32.reshape(3, 3)
        p = points[valid]
        q = np.stack([f32(f32(f32(R[i, 0] * p[:, 0]) + f32(R[i, 1] * p[:, 1])) + f32(R[i, 2] * p[:, 2])) + co.t32[i]
                      for i in range(3)], -1).astype(f32)
        allp = np.concatenate([q, cloud])
        s = np.sort(allp, axis=0)
        cnt = len(allp)
        want10, want90 = s[int(f32(cnt) * f32(.1))], s[int(f32(cnt) * f32(.9))]
        got_n, g10, g90 = ops.object_extent_stats(to_dev(points), to_dev(mask), co.R32, co.t32, to_dev(tsdf),
                                                  to_dev(wts), None if fgm is None else to_dev(fgm), vox)
        assert got_n == cnt
        assert g10.tobytes() == want10.tobytes() and g90.tobytes() == want90.tobytes()


@pytest.mark.parametrize("channels", [1, 2])
def test_copy_values(ops, dev, channels):
    rng = np.random.default_rng(6)
    src = rng.standard_normal((6, 5, 8) if channels == 1 else (6, 5, 8, channels)).astype(np.float32)  # (Nz,Ny,Nx)
    for off, dres in (((-2, 1, 0), (12, 6, 6)), ((3, -1, 2), (4, 8, 4)), ((0, 0, 0), (8, 5, 6))):
        dshape = (dres[2], dres[1], dres[0]) + (() if channels == 1 else (channels,))
        dst = dev_full(dshape, 7.0)
        ops.copy_values(to_dev(src), dst, off)
        want = np.zeros(dshape, np.float32)
        for z in range(6):
            for y in range(5):
                for x in range(8):  # kernel_copyValues: x_new = x - offset
                    xn, yn, zn = x - off[0], y - off[1], z - off[2]
                    if 0 <= xn < dres[0] and 0 <= yn < dres[1] and 0 <= zn < dres[2]:
                        want[zn, yn, xn] = src[z, y, x]
        assert np.array_equal(dst.numpy(), want), (off, dres)


def f32_transform(R, t, p):
    f32 = np.float32
    return np.stack([f32(f32(f32(R[i, 0] * p[:, 0]) + f32(R[i, 1] * p[:, 1])) + f32(R[i, 2] * p[:, 2])) + t[i]
                     for i in range(3)], -1).astype(f32)


def expected_resize(res, vox, p10, p90, vol_pad=2.0):
    """ObjTSDF::resize (ObjTSDF.cpp:80-165) in float32 numpy: (new centre, new res, voxel offset) or None."""
    f32 = np.float32
    half = (np.array(res, f32) - f32(1)) * f32(.5) * f32(vox)
    if not (np.any(p10 < -half) or np.any(p90 > half)):
        return None
    center = (p10 + p90) * f32(.5)
    pix = np.rint(center * (f32(1) / f32(vox))).astype(np.int32)  # cvRound: half to even
    center = pix.astype(f32) * f32(vox)
    size = f32(f32(vol_pad) * np.max(p90 - p10)) / f32(vox)
    n = (int(np.ceil(size)) + 1) // 2 * 2
    return center, n, pix - (n - np.array(res, np.int32)) // 2


def shifted(src, off, n):
    """copyValues: dst(v) = src(v + off) inside the source, 0 elsewhere; arrays are (z, y, x[, c])."""
    dst = np.zeros((n, n, n) + src.shape[3:], src.dtype)
    sz, sy, sx = src.shape[:3]
    lo = [max(0, -o) for o in off]                          # first dst index with a source
    hi = [min(n, s - o) for o, s in zip(off, (sx, sy, sz))]  # one past the last
    if all(h > l for l, h in zip(lo, hi)):
        dst[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]] = \
            src[lo[2] + off[2]:hi[2] + off[2], lo[1] + off[1]:hi[1] + off[1], lo[0] + off[0]:hi[0] + off[0]]
    return dst


def test_matched_object_outgrows_its_volume_and_is_resized(oracle, dev, tmp_path):
    """updateObj + ObjTSDF::resize against a numpy restatement, then one more frame on the resized
    (Nx % 4 != 0 allowed) volume against the oracle's integration."""
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    f32 = np.float32
    Wf, Hf = 320, 240
    prm = pipeline.make_params(Wf, Hf, 128, 0.04, 32, visibility_thresh=100, boundary=10)
    Kf = np.array(prm.K, np.float32).reshape(3, 3)
    synth = pipeline.SyntheticStream(Wf, Hf, Kf, 2, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    fus.enable_pose_log(True)
    keep = []
    depth, sid = synth.render(0)
    ys, xs = np.nonzero(sid == 1)
    cy, cx = int(ys.mean()), int(xs.mean())
    patch = np.zeros((Hf, Wf), np.uint8)
    patch[cy - 6:cy + 6, cx - 6:cx + 6] = 1
    patch &= (sid == 1).astype(np.uint8)
    centre = None
    for f in range(3):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        d, full = to_dev(depth), to_dev((sid == 1).astype(np.uint8))
        keep += [d, full]
        if f == 0:
            pm = to_dev(patch)
            keep.append(pm)
            fus.queue_new_object_masks([image_view(pm)])
            fus.process_frame(image_view(d), R, t, {}, {}, True)
            assert fus.last_created() == [1]
            centre = fus.pose(1)[1]
            continue
        fus.process_frame(image_view(d), R, t, {1: (np.eye(3, dtype=f32).reshape(-1), centre)},
                          {1: image_view(full)}, True)
    fus.synchronize()
    assert 1 in fus.visible_objects()
    before = {k: fus.volume(k, 1) for k in ("tsdf", "weights", "fgprobs", "fgmask")}
    res = before["tsdf"].shape[::-1]
    assert res == (32, 32, 32)
    pts = oracle.compute_points(depth, Kf.reshape(-1))
    depth0, _ = synth.render(0)
    R0, t0 = synth.camera_pose(0)
    n0, w10, w90 = sorted_stats(oracle.compute_points(depth0, Kf.reshape(-1)), patch, R0.reshape(3, 3), t0)
    vox = f32(f32(2) * np.max(w90 - w10)) / f32(32)  # EMFusion.cpp:535-547
    # expected percentiles: surface vertices of the volume + masked points, object frame
    Ro, to = fus.pose(1)
    Rc, tc = R.reshape(3, 3).astype(f32), t.astype(f32)
    assert np.array_equal(Ro, np.eye(3, dtype=f32))
    rel_t = (tc + (-to)).astype(f32)
    mask = (sid == 1).astype(np.uint8)
    valid = (mask != 0) & np.any(pts != 0, axis=2)
    allp = np.concatenate([f32_transform(Rc, rel_t, pts[valid]),
                           mesh_cloud(before["tsdf"], before["weights"], before["fgmask"], vox)])
    s = np.sort(allp, axis=0)
    p10, p90 = s[int(f32(len(allp)) * f32(.1))], s[int(f32(len(allp)) * f32(.9))]
    want = expected_resize(res, vox, p10, p90)
    assert want is not None, ("the scenario must outgrow the volume", p10, p90, vox, len(allp), int(valid.sum()))
    centre_shift, n, off = want
    got_shift = fus.update_object(1, image_view(full))
    assert got_shift.tobytes() == centre_shift.astype(f32).tobytes()
    after = {k: fus.volume(k, 1) for k in ("tsdf", "weights", "fgprobs", "fgmask")}
    assert after["tsdf"].shape == (n, n, n) and n > 32
    for k in after:
        assert np.array_equal(after[k], shifted(before[k], off, n)), k
    assert (after["weights"] > 0).sum() == (before["weights"] > 0).sum() > 0  # nothing was cropped here
    Rn, tn = fus.pose(1)
    assert np.array_equal(Rn, Ro) and tn.tobytes() == (to + centre_shift).astype(f32).tobytes()
    assert not np.any(fus.update_object(1, image_view(full)))  # contained now: no second resize
    # the next frame runs on the resized volume (batched path, any Nx) and integrates like the oracle
    depth, sid = synth.render(3)
    R, t = synth.camera_pose(3)
    d, full = to_dev(depth), to_dev((sid == 1).astype(np.uint8))
    fus.process_frame(image_view(d), R, t, {1: (np.eye(3, dtype=f32).reshape(-1), tn)}, {1: image_view(full)}, True)
    fus.synchronize()
    assert 1 in fus.visible_objects()
    tsdf, wts = after["tsdf"].copy(), after["weights"].copy()
    assoc = fus.image("obj_assoc", 1)
    Rc, tc = R.reshape(3, 3).astype(f32), t.astype(f32)
    R_oc = Rc.T.copy()                       # camera^-1 * object pose, object rotation = identity
    t_oc = (f32_transform(R_oc, np.zeros(3, f32), tn[None])[0] + (-f32_transform(R_oc, np.zeros(3, f32), tc[None])[0])).astype(f32)
    oracle.update_tsdf(depth, assoc, tsdf, wts, R_oc.reshape(-1), t_oc, Kf.reshape(-1), float(vox),
                       float(f32(f32(prm.obj_rel_truncdist) * f32(f32(2) * np.max(w90 - w10))) / f32(32)),
                       float(prm.max_tsdf_weight))
    assert np.array_equal(fus.volume("weights", 1), wts)
    assert np.array_equal(fus.volume("tsdf", 1), tsdf)
    # pose files: raw trajectory jumps with the centre, the corrected one does not
    fus.write_results(str(tmp_path), volumes=False)
    raw = np.loadtxt(tmp_path / "poses-1.txt")
    cor = np.loadtxt(tmp_path / "poses-1-corrected.txt")
    assert raw.shape == cor.shape and raw.shape[0] >= 3
    assert np.allclose(cor[:, 1:4], raw[0, 1:4], atol=1e-5)
    assert np.abs(raw[-1, 1:4] - raw[0, 1:4]).max() > 1e-3
    fus.close()
    synth.close()


def test_class_scores_accumulate_and_ignore_person_hides_the_object(oracle, dev, tmp_path):
    """Class probabilities (ObjTSDF::updateClassProbs / getClassID) and Params.ignore_person: the object
    whose accumulated scores say "person" is fused and tracked like the other one, but the rendering
    shows the background in its place and writeResults leaves its mesh out."""
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    Wf, Hf = 320, 240
    prm = pipeline.make_params(Wf, Hf, 128, 0.04, 32, visibility_thresh=400, boundary=10)
    synth = pipeline.SyntheticStream(Wf, Hf, np.array(prm.K, np.float32), 2, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    fus.enable_pose_log(True)
    rng = np.random.default_rng(3)
    person = np.zeros(81); person[1] = 0.9; person[57] = 0.1      # COCO: 1 = person, 57 = chair
    chair = np.zeros(81); chair[57] = 0.6; chair[1] = 0.3
    centers, keep = {}, []
    for f in range(4):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        d = to_dev(depth)
        inst = [to_dev((sid == k).astype(np.uint8)) for k in (1, 2)]
        keep += [d, inst]
        fus.queue_instance_masks([image_view(m) for m in inst])
        # object 2 looks like a person on two frames of three; a noisy "chair" vote in between
        fus.queue_instance_scores([chair + 0.01 * rng.random(81), person if f != 2 else chair])
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), c) for i, c in centers.items()}
        fus.process_frame(image_view(d), R, t, poses, {}, False)
        fus.synchronize()
        if f == 0:
            assert fus.last_mask_assignment() == [1, 2]
            centers = {k: fus.pose(k)[1] for k in (1, 2)}
    assert fus.object_class(1) == 57 and fus.object_class(2) == 1
    seg = fus.image("segmentation")
    assert (seg == 2).sum() > 100
    shown, _ = fus.render()
    fus.set_ignore_person(True)
    hidden, cmap = fus.render()
    seg_after = fus.image("segmentation")
    assert not (seg_after == 2).any() and (seg_after == 1).sum() == (seg == 1).sum()
    was = seg == 2
    want = oracle.render_phong(fus.image("vertices"), fus.image("normals"), seg_after, cmap)
    assert np.array_equal(hidden, want)
    assert np.array_equal(hidden[~was], shown[~was]) and not np.array_equal(hidden[was], shown[was])
    fus.write_results(str(tmp_path), volumes=True)
    assert (tmp_path / "mesh_1.ply").exists() and not (tmp_path / "mesh_2.ply").exists()
    assert (tmp_path / "tsdfs" / "tsdf_1.bin").exists() and not (tmp_path / "tsdfs" / "tsdf_2.bin").exists()
    fus.close()
    synth.close()
