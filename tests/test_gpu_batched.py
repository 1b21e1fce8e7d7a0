"""Parity of the level-3 (batched, model-table) launches and of the brick uniformity flags.

Bars: brick flags are exactly the uniformity class of the final voxel values; a raycast that
uses them is bit-identical to one that does not (and to the oracle); each batched launch is
bit-identical to running the per-volume level-1/2 entry points model by model."""
import numpy as np
import pytest

from tests.parity_util import assert_parity, dev_full, to_dev, to_np
from tests.scenes import Pose, camera_path, intrinsics, rel_CO, rel_OC, render_depth, rot

pytestmark = pytest.mark.gpu

W, H = 160, 120
K = intrinsics(W, H)
SPHERES = [((0.25, 0.05, 1.3), 0.22), ((-0.3, -0.1, 1.6), 0.18)]
SIGMA, ALPHA, PRIOR, MAXW = 0.02, 0.8, 1.0, 64.0


@pytest.fixture(scope="module")
def ops(dev):
    from emfusion_amd import ops as _ops
    return _ops


def frame(i):
    cam = camera_path(i)
    depth, ids = render_depth(W, H, K, cam, SPHERES, noise=0.002, dropout=0.01, seed=100 + i)
    return cam, depth, ids


BRICK = 4  # EMF_BRICK


def brick_classes(tsdf):
    """Reference flags from voxel values: 1/2/4 where a whole 4^3 brick is exactly 0/+1/-1."""
    nz, ny, nx = tsdf.shape
    bz, by, bx = [(n + BRICK - 1) // BRICK for n in (nz, ny, nx)]
    out = np.zeros((bz, by, bx), np.uint8)
    for code, val in ((1, 0.0), (2, 1.0), (4, -1.0)):
        eq = np.ones((bz * BRICK, by * BRICK, bx * BRICK), bool)  # outside the volume: neutral
        eq[:nz, :ny, :nx] = tsdf == val
        u = eq.reshape(bz, BRICK, by, BRICK, bx, BRICK).all((1, 3, 5))
        out[u] = code
    return out


def dilate(raw, max_depth=3):
    """Dilated flags: class | D << 3, D in 1..3 the largest depth such that every in-volume brick
    within Chebyshev distance D shares the (non-zero) class; 0 where D would be 0."""
    bz, by, bx = raw.shape
    depth = np.zeros(raw.shape, np.uint8)
    ok = raw != 0
    for D in range(1, max_depth + 1):
        pad = np.full((bz + 2 * D, by + 2 * D, bx + 2 * D), 255, np.uint8)  # 255 = outside: ignored
        pad[D:-D, D:-D, D:-D] = raw
        for dz in range(2 * D + 1):
            for dy in range(2 * D + 1):
                for dx in range(2 * D + 1):
                    nb = pad[dz:dz + bz, dy:dy + by, dx:dx + bx]
                    ok &= (nb == raw) | (nb == 255)
        depth[ok & ((raw == 2) | (D == 1))] = D  # depth > 1 is only searched for free space (+1)
    return np.where(depth > 0, raw | (depth << 3), 0).astype(np.uint8)


def check_flags(got, tsdf, what):
    want = brick_classes(tsdf)
    assert_parity(got[0], want, f"raw brick flags {what}", exact=True)
    assert_parity(got[1], dilate(want), f"dilated brick flags {what}", exact=True)
    return want


class Model:
    """One volume with its device buffers, integrated identically on oracle and device."""

    def __init__(self, ops, oracle, res, vox, pose, is_obj, mid):
        self.ops, self.oracle = ops, oracle
        self.res, self.vox, self.pose, self.id = res, np.float32(vox), pose, mid
        nz, ny, nx = res[2], res[1], res[0]
        self.tsdf = np.zeros((nz, ny, nx), np.float32)
        self.wts = np.zeros((nz, ny, nx), np.float32)
        self.d_tsdf, self.d_wts = to_dev(self.tsdf), to_dev(self.wts)
        self.d_flags = dev_full(ops.brick_shape(self.tsdf.shape), 0, np.uint8)
        ops.reset_brick_flags(self.d_tsdf, self.d_flags)
        self.is_obj = is_obj
        if is_obj:
            self.fgbg = np.zeros((nz, ny, nx, 2), np.float32)
            self.probs = np.zeros((nz, ny, nx), np.float32)
            self.vmask = np.zeros((nz, ny, nx), np.uint8)
        self.d_assoc = dev_full((H, W), 1.0)
        self.d_ray = dev_full((H, W), 0.0)
        self.d_vert = dev_full((H, W, 3), 0.0)
        self.d_nrm = dev_full((H, W, 3), 0.0)
        self.d_hit = dev_full((H, W), 0, np.uint8)
        self.d_sign = None  # sign maps (far bounds of the raycast), set by the tests that use them
        self.d_rel = None   # relevant-tile list built from them

    @property
    def trunc(self):
        return np.float32(10) * self.vox

    def integrate(self, cam, depth, assoc, with_flags=True):
        oc = rel_OC(cam, self.pose)
        self.oracle.update_tsdf(depth, assoc, self.tsdf, self.wts, oc.R32, oc.t32, K, self.vox,
                                self.trunc, MAXW)
        self.ops.update_tsdf(to_dev(depth), to_dev(assoc), self.d_tsdf, self.d_wts, oc.R32, oc.t32,
                             K, self.vox, self.trunc, MAXW,
                             brick_flags=self.d_flags if with_flags else None)

    def finish_fg(self, cam, ids):
        oc = rel_OC(cam, self.pose)
        self.oracle.update_fgbg_probs((ids == self.id).astype(np.uint8), np.zeros((H, W), np.uint8),
                                      self.tsdf, self.wts, self.fgbg, oc.R32, oc.t32, K, self.vox)
        self.probs, self.vmask = self.oracle.compute_fg_probs(self.fgbg)
        self.d_probs, self.d_vmask = to_dev(self.probs), to_dev(self.vmask)

    def table_entry(self):
        return self.ops.make_model(
            self.d_tsdf, self.d_wts, self.d_assoc, self.d_ray, self.d_vert, self.d_nrm, self.d_hit,
            float(self.vox), float(self.trunc), MAXW, SIGMA, ALPHA, PRIOR, model_id=self.id,
            fg_probs=self.d_probs if self.is_obj else None,
            fg_mask=self.d_vmask if self.is_obj else None, brick_flags=self.d_flags,
            rcp_voxel=self.ops.voxel_reciprocal(self.vox), sign_maps=self.d_sign, relevant_tiles=self.d_rel,
            unseen_tiles=getattr(self, "d_unseen", None))


@pytest.fixture(scope="module")
def scene(ops, oracle, dev):
    models = [Model(ops, oracle, (64, 64, 64), 0.04, Pose(t=[0, 0, 1.28]), False, 0),
              Model(ops, oracle, (32, 32, 32), 0.025, Pose(t=SPHERES[0][0]), True, 1),
              Model(ops, oracle, (40, 32, 24), 0.025, Pose(rot([0, 1, 0], 7), SPHERES[1][0]), True, 2)]
    for i in range(4):
        cam, depth, ids = frame(i)
        for m in models:
            m.integrate(cam, depth, np.ones((H, W), np.float32))
            if m.is_obj:
                m.finish_fg(cam, ids)
    dev.synchronize()
    return models


# ---- brick flags ---------------------------------------------------------------------------------

def test_brick_flags_are_exact_uniformity_classes(scene):
    for m in scene:
        assert_parity(to_np(m.d_tsdf), m.tsdf, f"tsdf model {m.id}", exact=True)
        check_flags(to_np(m.d_flags), m.tsdf, f"model {m.id}")
    bg = brick_classes(scene[0].tsdf)
    assert (bg == 1).any() and (bg == 2).any() and (bg == 0).any()


def test_brick_flags_linear_kernel_is_conservative(ops, oracle, dev):
    # Nx % 4 != 0 takes the one-lane-per-voxel kernel: flags may only degrade to MIXED
    res = (30, 22, 18)
    m = Model(ops, oracle, res, 2.56 / 30, Pose(t=[0, 0, 1.28]), False, 0)
    for i in range(2):
        cam, depth, _ = frame(i)
        m.integrate(cam, depth, np.ones((H, W), np.float32))
    assert_parity(to_np(m.d_tsdf), m.tsdf, "tsdf", exact=True)
    want, got = brick_classes(m.tsdf), to_np(m.d_flags)
    assert np.all((got[0] == want) | (got[0] == 0))
    assert_parity(got[1], dilate(got[0]), "dilated flags follow the raw flags", exact=True)


def test_culled_tiles_leave_volume_and_flags_untouched(ops, oracle, dev):
    # camera looking away: every tile projects outside the image or behind the camera
    m = Model(ops, oracle, (64, 64, 64), 0.04, Pose(t=[3.0, 0, 1.28]), False, 0)
    cam, depth, _ = frame(0)
    m.integrate(cam, depth, np.ones((H, W), np.float32))
    assert_parity(to_np(m.d_tsdf), m.tsdf, "tsdf", exact=True)
    check_flags(to_np(m.d_flags), m.tsdf, "culled volume")


@pytest.mark.parametrize("cam_name", ["tracked", "rotated", "inside"])
def test_raycast_with_brick_flags_is_bit_identical(ops, oracle, scene, cam_name, dev):
    cams = {"tracked": camera_path(4), "rotated": Pose(rot([0.3, 1, 0.2], 14), [0.2, -0.1, 0.15]),
            "inside": Pose(rot([0, 1, 0], -8), [0.0, 0.0, 0.5])}
    for m in scene:
        co = rel_CO(cams[cam_name], m.pose)
        want = oracle.raycast_tsdf(m.tsdf, None, m.wts, m.vmask if m.is_obj else None, W, H, co.R32,
                                   co.t32, K, m.vox, m.trunc, count_steps=True)
        outs = {}
        for use in (False, True):
            ray, vert, nrm, hit = (dev_full((H, W), 0.0), dev_full((H, W, 3), 0.0),
                                   dev_full((H, W, 3), 0.0), dev_full((H, W), 0, np.uint8))
            st = dev_full((4,), 0, np.uint64)
            ops.raycast_tsdf(m.d_tsdf, None, m.d_wts, m.d_vmask if m.is_obj else None, ray, vert,
                             nrm, hit, co.R32, co.t32, K, m.vox, m.trunc, st,
                             brick_flags=m.d_flags if use else None)
            outs[use] = [to_np(ray), to_np(vert), to_np(nrm), to_np(hit), to_np(st)]
        for k, name in enumerate(["ray", "vert", "normal", "mask"]):
            assert_parity(outs[True][k], want[k], f"{name} with flags vs oracle (model {m.id})",
                          exact=True)
            assert_parity(outs[True][k], outs[False][k], f"{name} flags vs no flags", exact=True)
        assert int(outs[True][4][0]) == int(want[4].sum()) == int(outs[False][4][0])


# ---- batched launches ----------------------------------------------------------------------------

def test_estep_batched_matches_per_model_calls(ops, oracle, scene, dev):
    cam, depth, _ = frame(4)
    pts = oracle.compute_points(depth, K)
    d_pts = to_dev(pts)
    table = ops.upload_models([m.table_entry() for m in scene])
    poses = [(rel_CO(cam, m.pose).R32, rel_CO(cam, m.pose).t32) for m in scene]
    # per-model path on the device (bit reference) and the oracle (tolerance reference)
    per, orc = [], []
    for m, (R, t) in zip(scene, poses):
        out = dev_full((H, W), 9.0)
        ops.compute_association(m.d_tsdf, m.d_probs if m.is_obj else None, d_pts, R, t, m.vox,
                                m.trunc, SIGMA, ALPHA, PRIOR, out)
        per.append(out)
        orc.append(oracle.compute_association(m.tsdf, m.probs if m.is_obj else None, pts, R, t,
                                              m.vox, m.trunc, SIGMA, ALPHA, PRIOR))
    raw = [to_np(p) for p in per]
    # un-normalised + object partial sum (multi-GPU form)
    d_sum = dev_full((H, W), 7.0)
    ops.estep_batched(table, poses, d_pts, normalize=False, obj_sum=d_sum)
    for m, r in zip(scene, raw):
        assert_parity(to_np(m.d_assoc), r, f"un-normalised map {m.id}", exact=True)
    assert_parity(to_np(d_sum), raw[1] + raw[2], "object partial sum", exact=True)
    # fused normalisation (single-GPU form)
    d_norm = dev_full((H, W), 7.0)
    ops.estep_batched(table, poses, d_pts, normalize=True, norm=d_norm)
    d_norm2 = dev_full((H, W), 0.0)
    ops.normalize_association(per, norm=d_norm2)
    assert_parity(to_np(d_norm), to_np(d_norm2), "norm", exact=True)
    for m, p in zip(scene, per):
        assert_parity(to_np(m.d_assoc), to_np(p), f"normalised map {m.id}", exact=True)
    want = [o.copy() for o in orc]
    oracle.normalize_association(want)
    for m, wv in zip(scene, want):
        assert_parity(to_np(m.d_assoc), wv, f"map {m.id} vs oracle", rtol=4e-6)


@pytest.mark.parametrize("normalize", [True, False], ids=["normalised", "partial_sums"])
def test_estep_from_depth_equals_points_then_estep(ops, oracle, scene, dev, normalize):
    """The frame's first E-step forms the points from the depth itself: the same points image (bit
    for bit the oracle's) and the same maps as compute_points followed by the E-step."""
    cam, depth, _ = frame(4)
    table = ops.upload_models([m.table_entry() for m in scene])
    poses = [(rel_CO(cam, m.pose).R32, rel_CO(cam, m.pose).t32) for m in scene]
    d_depth = to_dev(depth)
    d_pts = ops.compute_points(d_depth, K, dev_full((H, W, 3), 5.0))
    d_a, d_b = dev_full((H, W), 7.0), dev_full((H, W), 7.0)
    kw = dict(norm=d_a) if normalize else dict(obj_sum=d_a)
    ops.estep_batched(table, poses, d_pts, normalize=normalize, **kw)
    want = [to_np(m.d_assoc).copy() for m in scene]
    for m in scene:
        m.d_assoc.copy_from(np.full((H, W), 3.0, np.float32))
    d_pts2 = dev_full((H, W, 3), 5.0)
    kw = dict(norm=d_b) if normalize else dict(obj_sum=d_b)
    ops.estep_batched_from_depth(table, poses, d_depth, K, d_pts2, normalize=normalize, **kw)
    assert_parity(to_np(d_pts2), oracle.compute_points(depth, K), "points", exact=True)
    assert_parity(to_np(d_pts2), to_np(d_pts), "points vs compute_points", exact=True)
    assert_parity(to_np(d_b), to_np(d_a), "normaliser / partial sum", exact=True)
    for m, w in zip(scene, want):
        assert_parity(to_np(m.d_assoc), w, f"map {m.id}", exact=True)


@pytest.mark.parametrize("use_flags,footprints,rows", [(False, False, 1), (True, False, 1), (False, True, 1), (False, True, 2),
                                                       (False, False, 4)],
                         ids=["plain", "brick_flags", "object_footprints", "two_lanes_per_ray", "four_lanes_per_ray"])
def test_raycast_batched_matches_per_model_calls_and_zero_fills(ops, oracle, scene, dev, monkeypatch, use_flags, footprints, rows):
    """footprints: with the voxel sizes on the host, objects get marching workgroups only where their box
    projects to, and zero-fill workgroups elsewhere -- every pixel of every image is still written.
    rows: the background's rays with 2 / 4 lanes each (march_quad, EMF_MARCH_ROWS): same pixels, same sample count."""
    cam = camera_path(4)
    table = ops.upload_models([m.table_entry() for m in scene])
    poses = [(rel_CO(cam, m.pose).R32, rel_CO(cam, m.pose).t32) for m in scene]
    for m in scene:  # poison: the batched kernel must overwrite everything
        m.d_ray.copy_from(np.full((H, W), 5, np.float32))
        m.d_vert.copy_from(np.full((H, W, 3), 5, np.float32))
        m.d_nrm.copy_from(np.full((H, W, 3), 5, np.float32))
        m.d_hit.copy_from(np.full((H, W), 5, np.uint8))
    st = dev_full((4,), 0, np.uint64)
    ops.raycast_batched(table, poses, [m.res for m in scene], W, H, K, stats=st,
                        use_brick_flags=use_flags, voxel_sizes=[m.vox for m in scene] if footprints else None, lanes=rows)
    total = 0
    for m, (R, t) in zip(scene, poses):
        want = oracle.raycast_tsdf(m.tsdf, None, m.wts, m.vmask if m.is_obj else None, W, H, R, t,
                                   K, m.vox, m.trunc, count_steps=True)
        total += int(want[4].sum())
        assert want[3].sum() > 100
        for got, w_, name in zip([m.d_ray, m.d_vert, m.d_nrm, m.d_hit], want,
                                 ["ray", "vert", "normal", "mask"]):
            assert_parity(to_np(got), w_, f"{name} model {m.id}", exact=True)
    assert int(to_np(st)[0]) == total


@pytest.mark.parametrize("nmaps", [1, 2, 9, 17, 33, 70])
def test_table_normalisation_equals_the_map_list_form(ops, dev, nmaps):
    """emf_hip_normalizeAssociationTable (one launch over the `assoc` pointers of a device model table: what finishes the
    chunked E-step of more than EMF_MAX_BATCH models) against emf_hip_normalizeAssociation over the same maps and against
    the sequential float32 chain in numpy: bit-identical maps and normaliser, x / 0 := 0 included."""
    rng = np.random.default_rng(100 + nmaps)
    maps = [rng.uniform(0, 2, (H, W)).astype(np.float32) for _ in range(nmaps)]
    for m in maps:
        m[:3] = 0  # rows in which every likelihood is 0
    maps[0][5, :7] = 1e-30
    s = maps[0].copy()
    for m in maps[1:]:
        s = s + m
    with np.errstate(divide="ignore", invalid="ignore"):
        want = [np.where(s != 0, m / s, np.float32(0)).astype(np.float32) for m in maps]
    dummy = dev_full((4, 4, 4), 0.0)
    img, img3, hit = dev_full((H, W), 0.0), dev_full((H, W, 3), 0.0), dev_full((H, W), 0, np.uint8)
    d_maps = [to_dev(m, dev) for m in maps]
    table = ops.upload_models([ops.make_model(dummy, dummy, dm, img, img3, img3, hit, 0.01, 0.1, MAXW, SIGMA, ALPHA, PRIOR, model_id=k)
                               for k, dm in enumerate(d_maps)])
    d_norm = dev_full((H, W), -1.0)
    ops.normalize_association_table(table, nmaps, W, H, norm=d_norm)
    d_list = [to_dev(m, dev) for m in maps]
    d_norm2 = dev_full((H, W), -1.0)
    ops.normalize_association(d_list, norm=d_norm2)
    assert_parity(to_np(d_norm), s, "normaliser vs the numpy chain", exact=True)
    assert_parity(to_np(d_norm), to_np(d_norm2), "normaliser vs the map-list form", exact=True)
    for k in range(nmaps):
        assert_parity(to_np(d_maps[k]), want[k], f"map {k} vs numpy", exact=True)
        assert_parity(to_np(d_maps[k]), to_np(d_list[k]), f"map {k} vs the map-list form", exact=True)


def test_raycast_of_an_objects_only_table_chunk(ops, oracle, scene, dev):
    """emf_hip_raycastBatchedObjects: a chunk of the model table without a background in slot 0 (what a model list longer
    than EMF_MAX_BATCH is served with) -- every slot marched over its footprint, zero-filled elsewhere, every pixel
    written, same bits as the per-model march."""
    objs = [m for m in scene if m.is_obj]
    assert len(objs) >= 2
    cam = camera_path(4)
    table = ops.upload_models([m.table_entry() for m in objs])
    poses = [(rel_CO(cam, m.pose).R32, rel_CO(cam, m.pose).t32) for m in objs]
    for footprints in (False, True):
        for m in objs:
            m.d_ray.copy_from(np.full((H, W), 5, np.float32))
            m.d_vert.copy_from(np.full((H, W, 3), 5, np.float32))
            m.d_nrm.copy_from(np.full((H, W, 3), 5, np.float32))
            m.d_hit.copy_from(np.full((H, W), 5, np.uint8))
        st = dev_full((4,), 0, np.uint64)
        ops.raycast_batched(table, poses, [m.res for m in objs], W, H, K, stats=st, objects_only=True,
                            voxel_sizes=[m.vox for m in objs] if footprints else None)
        total = 0
        for m, (R, t) in zip(objs, poses):
            want = oracle.raycast_tsdf(m.tsdf, None, m.wts, m.vmask, W, H, R, t, K, m.vox, m.trunc, count_steps=True)
            total += int(want[4].sum())
            for got, w_, name in zip([m.d_ray, m.d_vert, m.d_nrm, m.d_hit], want, ["ray", "vert", "normal", "mask"]):
                assert_parity(to_np(got), w_, f"{name} model {m.id} (footprints {footprints})", exact=True)
        assert int(to_np(st)[0]) == total


def tile_sign_maps(tsdf):
    """numpy restatement of the sign maps: per 32 x 8 x 8 tile, any tsdf > 0 / any tsdf < 0."""
    nz, ny, nx = tsdf.shape
    tz, ty, tx = -(-nz // 8), -(-ny // 8), -(-nx // 32)
    pad = np.zeros((tz * 8, ty * 8, tx * 32), np.float32)
    pad[:nz, :ny, :nx] = tsdf
    t = pad.reshape(tz, 8, ty, 8, tx, 32)
    return np.concatenate([(t > 0).any(axis=(1, 3, 5)).reshape(-1), (t < 0).any(axis=(1, 3, 5)).reshape(-1)]).astype(np.uint8)


@pytest.mark.parametrize("view", ["path", "turned", "inside", "far_side", "grazing"])
def test_raycast_far_bounds_cut_marches_without_changing_a_pixel(ops, oracle, scene, dev, view):
    """emf_hip_raycastFarBounds + the cut in the march: every output equals the oracle's full march bit
    for bit from all kinds of viewpoints, with fewer samples taken; the rebuilt sign maps equal their
    numpy restatement."""
    cam = {"path": camera_path(4),
           "turned": Pose(rot([0, 1, 0], 38) @ rot([1, 0, 0], -11), [0.35, -0.1, 0.2]),
           "inside": Pose(rot([0.3, 1, 0], 160), [0.1, 0.05, 1.1]),       # inside the background volume, looking back
           "far_side": Pose(rot([0, 1, 0], 180), [0.0, 0.0, 3.2]),         # behind everything, looking at the back sides
           "grazing": Pose(rot([0, 1, 0], 89.5), [-1.2, 0.0, 1.0])}[view]  # along the volume's faces
    for m in scene:
        m.d_sign = dev_full((ops.sign_map_bytes(m.res),), 7, np.uint8)
        ops.rebuild_sign_maps(m.d_tsdf, m.d_sign)
        assert np.array_equal(to_np(m.d_sign), tile_sign_maps(m.tsdf)), m.id
    try:
        table = ops.upload_models([m.table_entry() for m in scene])
        poses = [(rel_CO(cam, m.pose).R32, rel_CO(cam, m.pose).t32) for m in scene]
        res = [m.res for m in scene]
        bounds = ops.raycast_far_bounds(table, poses, res, W, H, K)
        st_cut, st_full = dev_full((4,), 0, np.uint64), dev_full((4,), 0, np.uint64)
        ops.raycast_batched(table, poses, res, W, H, K, stats=st_full)
        full = [[to_np(a).copy() for a in (m.d_ray, m.d_vert, m.d_nrm, m.d_hit)] for m in scene]
        ops.raycast_batched(table, poses, res, W, H, K, stats=st_cut, far_bounds=bounds,
                            voxel_sizes=[m.vox for m in scene])  # + object footprints from all these viewpoints
        hits = 0
        for m, (R, t), f in zip(scene, poses, full):
            want = oracle.raycast_tsdf(m.tsdf, None, m.wts, m.vmask if m.is_obj else None, W, H, R, t, K, m.vox, m.trunc)
            hits += int(want[3].sum())
            for got, w_, f_, name in zip([m.d_ray, m.d_vert, m.d_nrm, m.d_hit], want, f, ["ray", "vert", "normal", "mask"]):
                assert_parity(to_np(got), w_, f"{view}: {name} model {m.id}", exact=True)
                assert to_np(got).tobytes() == f_.tobytes()
        b = to_np(bounds)
        assert b.shape == (len(scene), 2 * ((H + 15) // 16), 2 * ((W + 15) // 16)) and np.isfinite(b).all() and (b >= 0).all()
        # the same bounds from relevant-tile lists (what the host classes keep up to date after every integration)
        for m in scene:
            m.d_rel = dev_full((ops.relevant_tile_words(m.res),), 0xdead, np.uint32)
        table = ops.upload_models([m.table_entry() for m in scene])
        ops.update_relevant_tiles(table, res)
        assert np.array_equal(to_np(ops.raycast_far_bounds(table, poses, res, W, H, K, scan_mask=0)), b)
        assert np.array_equal(to_np(ops.raycast_far_bounds(table, poses, res, W, H, K, scan_mask=0b010)), b)  # mixed
        counts = [int(to_np(m.d_rel)[0]) for m in scene]
        assert all(0 < c <= ops.sign_map_bytes(m.res) // 2 for c, m in zip(counts, scene)), counts
        for m in scene:
            m.d_rel = None
        cut, whole = int(to_np(st_cut)[0]), int(to_np(st_full)[0])
        assert cut <= whole
        if view in ("path", "turned"):
            assert hits > 1000  # (volumes this small lie within reach of their surfaces almost everywhere:
            #                      the saving is checked at full size, tests/test_gpu_fullsize.py)
        # a model without sign maps keeps its whole range
        scene[1].d_sign = None
        table = ops.upload_models([m.table_entry() for m in scene])
        b2 = to_np(ops.raycast_far_bounds(table, poses, res, W, H, K))
        assert np.isinf(b2[1]).all() and np.array_equal(b2[0], b[0]) and np.array_equal(b2[2], b[2])
        # ... and so does one with maps that is neither scanned nor listed
        b3 = to_np(ops.raycast_far_bounds(table, poses, res, W, H, K, scan_mask=0b001))
        assert np.isinf(b3[1]).all() and np.isinf(b3[2]).all() and np.array_equal(b3[0], b[0])
    finally:
        for m in scene:
            m.d_sign = m.d_rel = None


@pytest.mark.parametrize("rows", [1, 2, 4], ids=["one_lane", "two_lanes", "four_lanes"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_far_bounds_are_conservative_on_adversarial_volumes(ops, oracle, dev, monkeypatch, seed, rows):
    """Volumes no integration would produce -- isolated positive and negative blobs in unseen space, thin
    sheets, sign noise, a surface hugging the volume's outer shell -- seen from random poses (outside,
    inside, grazing): the batched raycast with far bounds, relevant-tile lists and object footprints equals
    the oracle's full march bit for bit (a bound that is too tight anywhere would lose a hit).  rows: the same with
    two / four lanes per ray (march_quad) -- sign noise, thin sheets and shell surfaces are where its transparency
    test, its dropped speculation and its `continue` / `break` cases get exercised."""
    rng = np.random.default_rng(1000 + seed)
    res, vox = (96, 64, 80), 0.02
    nz, ny, nx = res[2], res[1], res[0]
    zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    t = np.zeros((nz, ny, nx), np.float32)
    w = np.zeros((nz, ny, nx), np.float32)
    for _ in range(6):  # blobs: positive outside, negative inside, observed
        c = rng.uniform([8, 8, 8], [nz - 8, ny - 8, nx - 8])
        r = rng.uniform(3, 14)
        d = (np.sqrt((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) - r) / 6.0
        near = np.abs(d) < 1.5
        t[near] = np.clip(d[near], -1, 1)
        w[near] = rng.integers(1, 64)
    if seed % 2:  # a thin sheet and sign noise
        t[nz // 2] = np.where(xx[nz // 2] % 7 < 3, -0.4, 0.6)
        w[nz // 2] = 5
        noisy = rng.uniform(size=t.shape) < 0.002
        t[noisy] = rng.choice([-1.0, 1.0, -0.3, 0.9], noisy.sum()).astype(np.float32)
        w[noisy] = 1
    else:  # a surface in the outermost cells of the volume
        t[:, :, :3] = np.array([0.8, 0.1, -0.7], np.float32)
        w[:, :, :3] = 9
        t[-3:] = np.array([-0.9, 0.2, 0.9], np.float32)[:, None, None]
        w[-3:] = 3
    m = Model(ops, oracle, res, vox, Pose(rot([0.1, 1, 0.2], 11 * seed), [0.1, -0.05, 1.4]), False, 0)
    m.tsdf[:], m.wts[:] = t, w
    m.d_tsdf.copy_from(t)
    m.d_wts.copy_from(w)
    m.d_probs = m.d_vmask = dev_full((1,), 0, np.uint8)
    m.d_sign = dev_full((ops.sign_map_bytes(res),), 9, np.uint8)
    ops.rebuild_sign_maps(m.d_tsdf, m.d_sign)
    m.d_rel = dev_full((ops.relevant_tile_words(res),), 0, np.uint32)
    table = ops.upload_models([m.table_entry()])
    ops.update_relevant_tiles(table, [res])
    hits = 0
    for k in range(6):
        axis, ang = rng.normal(size=3), rng.uniform(0, 360)
        cam = Pose(rot(axis, ang), m.pose.t + rng.uniform(-1.6, 1.6, 3) * (0.35 if k % 3 == 0 else 1.0))
        co = rel_CO(cam, m.pose)
        poses = [(co.R32, co.t32)]
        want = oracle.raycast_tsdf(m.tsdf, None, m.wts, None, W, H, co.R32, co.t32, K, m.vox, m.trunc)
        hits += int(want[3].sum())
        for mask in (0, 1):  # list walked / maps scanned
            bounds = ops.raycast_far_bounds(table, poses, [res], W, H, K, scan_mask=mask)
            ops.raycast_batched(table, poses, [res], W, H, K, far_bounds=bounds, voxel_sizes=[m.vox], lanes=rows)
            for got, w_, name in zip([m.d_ray, m.d_vert, m.d_nrm, m.d_hit], want, ["ray", "vert", "normal", "mask"]):
                assert_parity(to_np(got), w_, f"seed {seed} pose {k} scan {mask}: {name}", exact=True)
    assert hits > 2000, hits


def test_sign_maps_kept_by_the_integration_cover_the_exact_ones(ops, oracle, dev):
    """The tile integration launches (in place and out of place) leave sticky sign maps behind that are
    never short of a sign that is in the volume."""
    shapes = [((96, 64, 72), 0.035, Pose(t=[0, 0, 1.28]), False), ((32, 32, 32), 0.025, Pose(t=SPHERES[0][0]), True)]
    a = [Model(ops, oracle, r, v, p, o, i) for i, (r, v, p, o) in enumerate(shapes)]
    b = [Model(ops, oracle, r, v, p, o, i) for i, (r, v, p, o) in enumerate(shapes)]
    b2 = [Model(ops, oracle, r, v, p, o, i) for i, (r, v, p, o) in enumerate(shapes)]
    for m in a + b + b2:
        m.d_probs = m.d_vmask = dev_full((1,), 0, np.uint8)
    for m in a + b:
        m.d_sign = dev_full((ops.sign_map_bytes(m.res),), 0, np.uint8)
    for x, y in zip(b, b2):
        y.d_sign = x.d_sign  # the two copies of a volume share one pair of maps
    maps = [[dev_full((ops.integrate_dirty_map_bytes(r),), 0, np.uint8) for _ in range(2)] for r, _, _, _ in shapes]
    visible = dev_full((2,), 1, np.int32)
    front, back = b, b2
    for i in range(5):
        base, depth, _ = frame(i)
        cam = Pose(base.R @ rot([0, 1, 0], [0, 25, -20, 0, 8][i]), base.t)
        poses = []
        for k, m in enumerate(a):
            oc = rel_OC(cam, m.pose)
            poses.append((oc.R32, oc.t32))
        d_depth = to_dev(depth)
        res = [m.res for m in a]
        ops.integrate_batched_culled(ops.upload_models([m.table_entry() for m in a]), poses, res, visible, d_depth, K)
        outs = [(bk.d_tsdf, bk.d_wts, mp[i % 2], mp[1 - i % 2]) for bk, mp in zip(back, maps)]
        ops.integrate_batched_culled_out(ops.upload_models([m.table_entry() for m in front]), poses, res, visible, d_depth, K, outs)
        dev.synchronize()
        front, back = back, front
        for m, f in zip(a, front):
            exact = tile_sign_maps(to_np(m.d_tsdf))
            assert exact.sum() > 0
            for got, name in ((to_np(m.d_sign), "in place"), (to_np(f.d_sign), "out of place")):
                assert ((got != 0) | (exact == 0)).all(), (i, m.id, name)  # got covers exact
                assert got.sum() <= exact.sum() + 0.2 * exact.size       # ... without being everything


def tile_all(a, pred):
    """per 32 x 8 x 8 tile (index (tz * nty + ty) * ntx + tx): pred holds for every voxel"""
    nz, ny, nx = a.shape
    ntx, nty, ntz = -(-nx // 32), -(-ny // 8), -(-nz // 8)
    out = np.zeros(ntx * nty * ntz, np.uint8)
    for tz in range(ntz):
        for ty in range(nty):
            for tx in range(ntx):
                out[(tz * nty + ty) * ntx + tx] = pred(a[tz * 8:tz * 8 + 8, ty * 8:ty * 8 + 8, tx * 32:tx * 32 + 32]).all()
    return out


def test_unseen_tiles_are_integrated_without_being_read_and_to_the_same_bits(ops, oracle, dev):
    """With emf_model_t.unseenTiles a tile whose weights are all 0 is integrated from the depth alone (drop-outs
    flip unseen voxels between -1 and 0 from frame to frame, pixels outside the image and association weight
    0 keep what is there): volumes bit-identical to the launches without the map -- in place and out of
    place, under a swinging camera -- and the map stays exactly "every weight of the tile is 0"."""
    shapes = [((96, 64, 72), 0.035, Pose(t=[0, 0, 1.28]), False), ((32, 32, 32), 0.025, Pose(t=SPHERES[0][0]), True)]
    mk = lambda: [Model(ops, oracle, r, v, p, o, i) for i, (r, v, p, o) in enumerate(shapes)]
    ref, a, b, b2 = mk(), mk(), mk(), mk()
    for m in ref + a + b + b2:
        m.d_probs = m.d_vmask = dev_full((1,), 0, np.uint8)
    for m in a + b:
        m.d_unseen = dev_full((ops.unseen_tile_bytes(m.res),), 1, np.uint8)  # cleared volumes: every tile unseen
    for x, y in zip(b, b2):
        y.d_unseen = x.d_unseen  # the two copies of a volume share the map
    maps = [[dev_full((ops.integrate_dirty_map_bytes(r),), 0, np.uint8) for _ in range(2)] for r, _, _, _ in shapes]
    visible = dev_full((2,), 1, np.int32)
    rng = np.random.default_rng(11)
    front, back = b, b2
    for i in range(6):
        base, depth, _ = frame(i)
        cam = Pose(base.R @ rot([0, 1, 0], [0, 25, -20, 0, 8, 30][i]), base.t)
        assoc = rng.uniform(0.0, 1.0, (H, W)).astype(np.float32)
        assoc[rng.uniform(size=(H, W)) < 0.2] = 0.0  # association weight 0: the voxel keeps its value
        poses = [(rel_OC(cam, m.pose).R32, rel_OC(cam, m.pose).t32) for m in ref]
        d_depth, res = to_dev(depth), [m.res for m in ref]
        for group in (ref, a, front):
            for m in group:
                m.d_assoc.copy_from(assoc)
        ops.integrate_batched_culled(ops.upload_models([m.table_entry() for m in ref]), poses, res, visible, d_depth, K)
        ops.integrate_batched_culled(ops.upload_models([m.table_entry() for m in a]), poses, res, visible, d_depth, K)
        outs = [(bk.d_tsdf, bk.d_wts, mp[i % 2], mp[1 - i % 2]) for bk, mp in zip(back, maps)]
        ops.integrate_batched_culled_out(ops.upload_models([m.table_entry() for m in front]), poses, res, visible,
                                         d_depth, K, outs)
        dev.synchronize()
        front, back = back, front
        for r, m, f in zip(ref, a, front):
            want_t, want_w = to_np(r.d_tsdf), to_np(r.d_wts)
            assert (want_w > 0).any() and (want_t == -1).any()
            for got, name in ((m, "in place"), (f, "out of place")):
                assert_parity(to_np(got.d_tsdf), want_t, f"frame {i} model {r.id} tsdf {name}", exact=True)
                assert_parity(to_np(got.d_wts), want_w, f"frame {i} model {r.id} weights {name}", exact=True)
                exact = tile_all(want_w, lambda w: w == 0)
                assert np.array_equal(to_np(got.d_unseen) != 0, exact != 0), (i, r.id, name)
        assert 0 < (to_np(a[0].d_unseen) != 0).sum() < a[0].d_unseen.shape[0]
    # the map from the values
    m = a[0]
    again = dev_full((ops.unseen_tile_bytes(m.res),), 7, np.uint8)
    ops.rebuild_unseen_tiles(m.d_tsdf, m.d_wts, again)
    assert np.array_equal(to_np(again), tile_all(to_np(m.d_wts), lambda w: w == 0))


@pytest.mark.parametrize("table", [False, True], ids=["inline", "table"])
def test_integrate_batched_matches_per_model_calls_and_honours_gate(ops, oracle, dev, table):
    il = None
    if table:
        il = dev_full((H, W), -3.0)
        ops.compute_inv_lambda(K, il)
    models = [Model(ops, oracle, (64, 64, 64), 0.04, Pose(t=[0, 0, 1.28]), False, 0),
              Model(ops, oracle, (32, 32, 32), 0.025, Pose(t=SPHERES[0][0]), True, 1),
              Model(ops, oracle, (32, 32, 32), 0.025, Pose(t=SPHERES[1][0]), True, 2)]
    for m in models:
        m.d_probs = m.d_vmask = dev_full((1,), 0, np.uint8)  # unused by integrate
    rng = np.random.default_rng(9)
    visible = dev_full((3,), 1, np.int32)
    stats = dev_full((1,), 0, np.uint64)
    expect_vox = 0
    for i in range(3):
        cam, depth, ids = frame(i)
        gate = [1, 1, 0 if i == 1 else 1]  # object 2 is "not visible" on frame 1
        visible.copy_from(np.array(gate, np.int32))
        mtab = ops.upload_models([m.table_entry() for m in models])
        d_depth = to_dev(depth)
        poses = []
        for m, g in zip(models, gate):
            assoc = rng.uniform(0, 1, (H, W)).astype(np.float32)
            m.d_assoc.copy_from(assoc)
            oc = rel_OC(cam, m.pose)
            poses.append((oc.R32, oc.t32))
            if g:
                oracle.update_tsdf(depth, assoc, m.tsdf, m.wts, oc.R32, oc.t32, K, m.vox, m.trunc,
                                   MAXW)
                expect_vox += m.tsdf.size
        ops.integrate_batched(mtab, poses, [m.res for m in models], visible, d_depth, K, stats,
                              inv_lambda=il)
        dev.synchronize()
    for m in models:
        assert_parity(to_np(m.d_tsdf), m.tsdf, f"tsdf model {m.id}", exact=True)
        assert_parity(to_np(m.d_wts), m.wts, f"weights model {m.id}", exact=True)
        check_flags(to_np(m.d_flags), m.tsdf, f"model {m.id}")
    assert int(to_np(stats)[0]) == expect_vox


def test_integrate_batched_mixes_tiled_and_untiled_models(ops, oracle, dev):
    # an object resized by ObjTSDF::resize only has an even Nx: it rides in the same call on the
    # one-voxel-per-lane launch, wherever it sits in the table
    models = [Model(ops, oracle, (34, 34, 34), 0.025, Pose(t=SPHERES[1][0]), True, 1),
              Model(ops, oracle, (64, 64, 64), 0.04, Pose(t=[0, 0, 1.28]), False, 0),
              Model(ops, oracle, (30, 22, 18), 0.03, Pose(rot([0, 1, 0], 5), SPHERES[0][0]), True, 2),
              Model(ops, oracle, (32, 32, 32), 0.025, Pose(t=SPHERES[0][0]), True, 3),
              Model(ops, oracle, (33, 31, 35), 0.025, Pose(t=SPHERES[1][0]), True, 4)]
    for m in models:
        m.d_probs = m.d_vmask = dev_full((1,), 0, np.uint8)
    rng = np.random.default_rng(21)
    visible = dev_full((5,), 1, np.int32)
    stats = dev_full((1,), 0, np.uint64)
    expect_vox = 0
    for i in range(3):
        cam, depth, ids = frame(i)
        gate = [1, 1, 0 if i == 1 else 1, 1, 0 if i == 2 else 1]
        visible.copy_from(np.array(gate, np.int32))
        mtab = ops.upload_models([m.table_entry() for m in models])
        poses = []
        for m, g in zip(models, gate):
            assoc = rng.uniform(0, 1, (H, W)).astype(np.float32)
            m.d_assoc.copy_from(assoc)
            oc = rel_OC(cam, m.pose)
            poses.append((oc.R32, oc.t32))
            if g:
                oracle.update_tsdf(depth, assoc, m.tsdf, m.wts, oc.R32, oc.t32, K, m.vox, m.trunc,
                                   MAXW)
                expect_vox += m.tsdf.size
        ops.integrate_batched(mtab, poses, [m.res for m in models], visible, to_dev(depth), K, stats)
        dev.synchronize()
    for m in models:
        assert_parity(to_np(m.d_tsdf), m.tsdf, f"tsdf model {m.id}", exact=True)
        assert_parity(to_np(m.d_wts), m.wts, f"weights model {m.id}", exact=True)
        want, got = brick_classes(m.tsdf), to_np(m.d_flags)
        if m.res[0] % 4 == 0:
            check_flags(got, m.tsdf, f"model {m.id}")
        else:
            assert np.all((got[0] == want) | (got[0] == 0))
    assert int(to_np(stats)[0]) == expect_vox
    # a table of untiled models only: the tiled launch is skipped
    only = [models[0], models[2]]
    cam, depth, _ = frame(3)
    mtab = ops.upload_models([m.table_entry() for m in only])
    poses = []
    for m in only:
        assoc = np.ones((H, W), np.float32)
        m.d_assoc.copy_from(assoc)
        oc = rel_OC(cam, m.pose)
        poses.append((oc.R32, oc.t32))
        oracle.update_tsdf(depth, assoc, m.tsdf, m.wts, oc.R32, oc.t32, K, m.vox, m.trunc, MAXW)
    ops.integrate_batched(mtab, poses, [m.res for m in only], None, to_dev(depth), K, None)
    dev.synchronize()
    for m in only:
        assert_parity(to_np(m.d_tsdf), m.tsdf, f"tsdf model {m.id}", exact=True)


@pytest.mark.parametrize("launch", ["all", "exact", "short", "one"])
def test_integrate_culled_launch_equals_the_plain_batched_one(ops, oracle, dev, launch):
    """Two-level launch (cull 2x2x2-tile boxes, then only the survivors' tiles): same volumes bit for bit,
    whatever the grid estimate -- all boxes, the exact survivor count, too few (the strided rest kernel
    picks up the remainder), a single box."""
    models = [Model(ops, oracle, (128, 96, 80), 0.03, Pose(t=[0, 0, 1.28]), False, 0),
              Model(ops, oracle, (32, 32, 32), 0.025, Pose(t=SPHERES[0][0]), True, 1),
              Model(ops, oracle, (40, 24, 36), 0.025, Pose(rot([0, 1, 0], 9), SPHERES[1][0]), True, 2)]
    twins = [Model(ops, oracle, m.res, m.vox, m.pose, m.is_obj, m.id) for m in models]
    for m in models + twins:
        m.d_probs = m.d_vmask = dev_full((1,), 0, np.uint8)
    rng = np.random.default_rng(31)
    visible = dev_full((3,), 1, np.int32)
    survivors = dev_full((1,), 0, np.uint32)
    stats_a, stats_b = dev_full((1,), 0, np.uint64), dev_full((1,), 0, np.uint64)
    nboxes = sum(-(-r[0] // 32) * -(-r[1] // 16) * -(-r[2] // 16) for r in (m.res for m in models))
    counts = []
    for i in range(3):
        cam, depth, ids = frame(i)
        gate = [1, 1, 0 if i == 1 else 1]
        visible.copy_from(np.array(gate, np.int32))
        poses = []
        for m, tw in zip(models, twins):
            assoc = rng.uniform(0, 1, (H, W)).astype(np.float32)
            m.d_assoc.copy_from(assoc)
            tw.d_assoc.copy_from(assoc)
            oc = rel_OC(cam, m.pose)
            poses.append((oc.R32, oc.t32))
        d_depth = to_dev(depth)
        ops.integrate_batched(ops.upload_models([m.table_entry() for m in twins]), poses, [m.res for m in twins],
                              visible, d_depth, K, stats_b)
        tab = ops.upload_models([m.table_entry() for m in models])
        if launch == "all":
            lb = 0
        elif launch == "one":
            lb = 1
        else:  # learn the count from a dry run on scratch copies? no: survivors only depend on geometry + gate
            probe = [Model(ops, oracle, m.res, m.vox, m.pose, m.is_obj, m.id) for m in models]
            for p_ in probe:
                p_.d_probs = p_.d_vmask = dev_full((1,), 0, np.uint8)
            ops.integrate_batched_culled(ops.upload_models([p_.table_entry() for p_ in probe]), poses,
                                         [m.res for m in models], visible, d_depth, K, 0, survivors)
            dev.synchronize()
            n = int(to_np(survivors)[0])
            lb = n if launch == "exact" else max(1, n // 3)
        ops.integrate_batched_culled(tab, poses, [m.res for m in models], visible, d_depth, K, lb, survivors, stats_a)
        dev.synchronize()
        counts.append(int(to_np(survivors)[0]))
    assert all(0 < c < nboxes for c in counts), (counts, nboxes)  # some boxes culled, some kept
    for m, tw in zip(models, twins):
        assert_parity(to_np(m.d_tsdf), to_np(tw.d_tsdf), f"tsdf model {m.id}", exact=True)
        assert_parity(to_np(m.d_wts), to_np(tw.d_wts), f"weights model {m.id}", exact=True)
    assert int(to_np(stats_a)[0]) == int(to_np(stats_b)[0]) > 0


@pytest.mark.parametrize("prepared", [False, True], ids=["self_clearing", "prepared_ahead"])
def test_integrate_out_of_place_equals_in_place_over_a_swinging_camera(ops, oracle, dev, prepared):
    """emf_hip_integrateBatchedCulledOut on double-buffered volumes: after every frame the copy that was
    written equals the in-place result bit for bit -- also where the camera has swung away from what it
    integrated a frame earlier (boxes outside the view cone whose tiles are still dirty get copied), with
    frames in which nothing at all is seen, and with the visibility gate closing a model for a frame."""
    shapes = [((128, 96, 80), 0.03, Pose(t=[0, 0, 1.28]), False), ((32, 32, 32), 0.025, Pose(t=SPHERES[0][0]), True),
              ((40, 24, 36), 0.025, Pose(rot([0, 1, 0], 9), SPHERES[1][0]), True)]
    twins = [Model(ops, oracle, r, v, p, o, i) for i, (r, v, p, o) in enumerate(shapes)]
    front = [Model(ops, oracle, r, v, p, o, i) for i, (r, v, p, o) in enumerate(shapes)]
    back = [Model(ops, oracle, r, v, p, o, i) for i, (r, v, p, o) in enumerate(shapes)]
    for m in twins + front + back:
        m.d_probs = m.d_vmask = dev_full((1,), 0, np.uint8)
    maps = [[dev_full((ops.integrate_dirty_map_bytes(r),), 0, np.uint8) for _ in range(2)] for r, _, _, _ in shapes]
    rng = np.random.default_rng(77)
    visible = dev_full((3,), 1, np.int32)
    # yaw in degrees per frame: look ahead, swing far right, back, far left + up, ahead, away (sees nothing), ahead
    swings = [(0, 0), (35, 0), (0, 0), (-40, 12), (5, -3), (170, 0), (0, 0), (2, 1)]
    copied_only = 0
    scratch = None
    for i, (yaw, pitch) in enumerate(swings):
        base, depth, _ = frame(i)
        cam = Pose(base.R @ rot([0, 1, 0], yaw) @ rot([1, 0, 0], pitch), base.t)
        if i >= 3:  # the depth image need not match the view: only determinism matters here
            depth = np.roll(depth, 17 * i, axis=1)
        visible.copy_from(np.array([1, 1, 0 if i == 2 else 1], np.int32))
        poses = []
        for a, b, c in zip(twins, front, back):
            assoc = rng.uniform(0, 1, (H, W)).astype(np.float32)
            for m in (a, b, c):
                m.d_assoc.copy_from(assoc)
            oc = rel_OC(cam, a.pose)
            poses.append((oc.R32, oc.t32))
        d_depth = to_dev(depth)
        res = [m.res for m in twins]
        ops.integrate_batched_culled(ops.upload_models([m.table_entry() for m in twins]), poses, res, visible, d_depth, K)
        outs = [(b.d_tsdf, b.d_wts, mp[i % 2], mp[1 - i % 2]) for b, mp in zip(back, maps)]
        before = [to_np(b.d_tsdf).copy() for b in back]
        # prepared: the counter and the next-dirty maps were cleared behind the previous call
        # (emf_hip_integratePrepareOut), as the host classes do to keep the fills off the frame's path
        scratch = ops.integrate_batched_culled_out(ops.upload_models([m.table_entry() for m in front]), poses, res,
                                                   visible, d_depth, K, outs, scratch=scratch, prepared=prepared and i > 0)
        if prepared:  # the copies swap roles: the next call's dirtyNext is this call's dirtyPrev
            ops.integrate_prepare_out([(f.d_tsdf, f.d_wts, mp[1 - i % 2], mp[i % 2]) for f, mp in zip(front, maps)],
                                      res, scratch)
        dev.synchronize()
        for a, f, b, old in zip(twins, front, back, before):
            assert_parity(to_np(b.d_tsdf), to_np(a.d_tsdf), f"frame {i} tsdf model {a.id}", exact=True)
            assert_parity(to_np(b.d_wts), to_np(a.d_wts), f"frame {i} weights model {a.id}", exact=True)
            # tiles that were only brought up to date (the front copy already held these values)
            copied_only += int(((to_np(b.d_tsdf) != old) & (to_np(b.d_tsdf) == to_np(f.d_tsdf))).sum())
        front, back = back, front  # flip
    assert copied_only > 1000, copied_only
    # a gated model is left alone in BOTH copies: its next-dirty map stays clean and nothing is lost
    assert int(to_np(maps[0][0]).sum()) + int(to_np(maps[0][1]).sum()) > 0


def test_visibility_flags(ops, dev):
    counts = to_dev(np.array([1601, 1600, 0, 99999], np.int32))
    vis = dev_full((5,), -1, np.int32)
    ops.visibility_flags(counts, 5, 1600, vis)
    assert to_np(vis).tolist() == [1, 1, 0, 0, 1]


def test_batched_argument_checks(ops, dev):
    from emfusion_amd._lib import EmfHipError
    pts = dev_full((H, W, 3), 0.0)
    table = dev_full((64,), 0, np.uint8)
    with pytest.raises(EmfHipError) as e:
        ops.estep_batched(table, [(np.eye(3), np.zeros(3))] * 33, pts)
    assert e.value.code == -5  # EMF_E_LIMIT
    with pytest.raises(EmfHipError) as e:  # (normalize=False without objSum is legal since ABI 8: a chunk of a longer list)
        ops.raycast_batched(table, [(np.eye(3), np.zeros(3))], [(32, 32, 32)], W, H, K, lanes=3)
    assert e.value.code == -4  # 1, 2 or 4 lanes per background ray


def test_argument_checks_of_the_newer_entry_points(ops, dev):
    """Rejected calls enqueue nothing and name the reason (EMF_E_*): culled integration, meshes, rendering,
    resize helpers."""
    from emfusion_amd._lib import EmfHipError
    table = dev_full((288,), 0, np.uint8)
    depth = dev_full((H, W), 1.0)
    pose = [(np.eye(3), np.zeros(3))]
    with pytest.raises(EmfHipError) as e:  # untiled model: the two-level launch is for float4 tiles only
        ops.integrate_batched_culled(table, pose, [(30, 22, 18)], None, depth, K)
    assert e.value.code == -2
    with pytest.raises(EmfHipError) as e:
        ops.integrate_batched_culled(table, pose * 33, [(32, 32, 32)] * 33, None, depth, K)
    assert e.value.code == -5
    vol = dev_full((4, 4, 4), 0.0)
    with pytest.raises(EmfHipError) as e:  # more than 3 channels
        ops.copy_values(dev_full((4, 4, 4, 4), 0.0), dev_full((4, 4, 4, 4), 0.0), (0, 0, 0))
    assert e.value.code == -4
    img3 = dev_full((H, W, 3), 0.0)
    seg = dev_full((H, W), 0, np.uint8)
    with pytest.raises(EmfHipError) as e:  # image of another size
        ops.render_phong(img3, img3, seg, np.zeros((256, 3), np.uint8), dev_full((H, W + 1, 3), 0, np.uint8))
    assert e.value.code == -2
    v, n, t = ops.extract_mesh(vol, vol, 0.01)  # smallest useful volume: nothing observed, empty mesh
    assert len(v) == 0 and len(t) == 0


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_unseen_and_deep_tiles_under_random_cameras(ops, oracle, dev, seed):
    """Random intrinsics (short and unequal focal lengths: the margin of the deep-tile test depends on
    them), random camera poses around and inside the volume, depth maps that are near, far, full of holes
    or empty: the two-level launch with the unseen-tile map (and its deep-tile shortcut) leaves the bytes
    of the one-level launch without any map, frame after frame."""
    rng = np.random.default_rng(1000 + seed)
    res, vox = (96, 64, 56), 0.03
    w, h = 96, 72
    fx, fy = rng.uniform(40, 160), rng.uniform(40, 160)
    Kr = np.array([fx, 0, w / 2 - 0.5 + rng.uniform(-5, 5), 0, fy, h / 2 - 0.5 + rng.uniform(-5, 5), 0, 0, 1], np.float32)

    def vol():
        m = Model(ops, oracle, res, vox, Pose(t=[0, 0, 1.5]), False, 0)
        m.d_probs = m.d_vmask = dev_full((1,), 0, np.uint8)
        m.d_assoc = dev_full((h, w), 1.0)
        return m
    plain, fast = vol(), vol()
    fast.d_unseen = dev_full((ops.unseen_tile_bytes(res),), 1, np.uint8)
    visible = dev_full((1,), 1, np.int32)
    for i in range(7):
        kind = rng.integers(0, 5)
        if kind == 0:
            depth = rng.uniform(0.3, 1.2, (h, w)).astype(np.float32)
        elif kind == 1:
            depth = rng.uniform(1.0, 4.0, (h, w)).astype(np.float32)
        elif kind == 2:
            depth = np.full((h, w), rng.uniform(0.4, 2.0), np.float32)
        elif kind == 3:
            depth = (rng.uniform(0.5, 2.5, (h, w)) * (rng.uniform(size=(h, w)) < 0.5)).astype(np.float32)
        else:
            depth = np.zeros((h, w), np.float32)
        if kind != 4:
            depth[rng.uniform(size=(h, w)) < 0.03] = 0.0
        ang = rng.uniform(-40, 40, 3)
        cam = Pose(rot([1, 0, 0], ang[0]) @ rot([0, 1, 0], ang[1]) @ rot([0, 0, 1], ang[2]),
                   rng.uniform(-0.6, 0.6, 3) + np.array([0, 0, rng.uniform(-0.5, 1.2)]))
        assoc = rng.uniform(0.0, 1.0, (h, w)).astype(np.float32)
        assoc[rng.uniform(size=(h, w)) < 0.1] = 0.0
        for m in (plain, fast):
            m.d_assoc.copy_from(assoc)
        oc = rel_OC(cam, plain.pose)
        d_depth = to_dev(depth)
        ops.integrate_batched(ops.upload_models([plain.table_entry()]), [(oc.R32, oc.t32)], [res], visible, d_depth, Kr)
        ops.integrate_batched_culled(ops.upload_models([fast.table_entry()]), [(oc.R32, oc.t32)], [res], visible,
                                     d_depth, Kr)
        dev.synchronize()
        assert_parity(to_np(fast.d_tsdf), to_np(plain.d_tsdf), f"seed {seed} frame {i} (depth kind {kind}) tsdf", exact=True)
        assert_parity(to_np(fast.d_wts), to_np(plain.d_wts), f"seed {seed} frame {i} weights", exact=True)
        assert np.array_equal(to_np(fast.d_unseen) != 0, tile_all(to_np(plain.d_wts), lambda x: x == 0) != 0)
