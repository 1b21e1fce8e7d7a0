"""Parity of the HIP path (through the emf_hip_* C ABI) against the CPU oracle on identical
seeded inputs, one test group per SURVEY.md section-8 row.  Integer / mask outputs must match
bit for bit; float outputs must be within the north-star tolerance (1e-4 relative) -- and for
the kernels whose arithmetic is IEEE-exact on both sides (no expf) the tests demand bit-identical
floats, which is stronger and keeps real bugs from hiding inside an outlier budget.
"""
import numpy as np
import pytest

from tests.parity_util import assert_parity, dev_full, to_dev, to_np
from tests.scenes import Pose, camera_path, intrinsics, rel_CO, rel_OC, render_depth, rot

pytestmark = pytest.mark.gpu

W, H = 160, 120
K = intrinsics(W, H)
SPHERES = [((0.25, 0.05, 1.3), 0.22), ((-0.3, -0.1, 1.6), 0.18)]
BG = dict(n=(64, 64, 64), vox=0.04, pose=Pose(t=[0, 0, 1.28]))
SIGMA, ALPHA, PRIOR, MAXW = 0.02, 0.8, 1.0, 64.0


@pytest.fixture(scope="module")
def ops(dev):
    from emfusion_amd import ops as _ops
    return _ops


def frame(i, noise=0.002, dropout=0.01):
    cam = camera_path(i)
    depth, ids = render_depth(W, H, K, cam, SPHERES, noise=noise, dropout=dropout, seed=100 + i)
    return cam, depth, ids


def alloc_vol(res_xyz, ch=1, dtype=np.float32):
    nx, ny, nz = res_xyz
    return np.zeros((nz, ny, nx) if ch == 1 else (nz, ny, nx, ch), dtype)


def inv_lambda_table(ops, pad=0):
    """The per-pixel 1 / lambda table of emf_hip_computeInvLambda for the test intrinsics."""
    tab = dev_full((H, W), -3.0, pad_cols=pad)
    ops.compute_inv_lambda(K, tab)
    return tab


def integrate_both(oracle, ops, dev, res, vox, vol_pose, frames, assoc_fn=None, max_w=MAXW,
                   pad=0, table=False):
    """Run the same integration sequence on the oracle (numpy) and the HIP path (device)."""
    tsdf, wts = alloc_vol(res), alloc_vol(res)
    d_tsdf, d_wts = to_dev(tsdf, dev), to_dev(wts, dev)
    il = inv_lambda_table(ops, pad) if table else None
    for i in frames:
        cam, depth, ids = frame(i)
        assoc = np.ones((H, W), np.float32) if assoc_fn is None else assoc_fn(i, ids)
        oc = rel_OC(cam, vol_pose)
        oracle.update_tsdf(depth, assoc, tsdf, wts, oc.R32, oc.t32, K, vox, 10 * vox, max_w)
        ops.update_tsdf(to_dev(depth, dev, pad), to_dev(assoc, dev, pad), d_tsdf, d_wts, oc.R32,
                        oc.t32, K, vox, 10 * vox, max_w, inv_lambda=il)
    dev.synchronize()
    return (tsdf, wts), (d_tsdf, d_wts)


# ---- a1 ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("pad", [0, 5])
def test_compute_points(oracle, ops, dev, pad):
    _, depth, _ = frame(0)
    want = oracle.compute_points(depth, K)
    pts = dev_full((H, W, 3), -7.0, pad_cols=pad)
    ops.compute_points(to_dev(depth, dev, pad), K, pts)
    assert_parity(to_np(pts), want, "points", exact=True)


# ---- a7 ------------------------------------------------------------------------------------------

def test_inv_lambda_table_is_the_inline_expression(ops, dev):
    """1 / |((x - cx) / fx, (y - cy) / fy, 1)| in float32, operation by operation (TSDF.cu:374-380)."""
    f32 = np.float32
    k = np.asarray(K, f32).reshape(-1)
    xs = (np.arange(W, dtype=f32) - k[2]) / k[0]
    ys = (np.arange(H, dtype=f32) - k[5]) / k[4]
    xx, yy = xs[None, :] * xs[None, :], ys[:, None] * ys[:, None]
    want = f32(1) / np.sqrt((xx + yy) + f32(1), dtype=f32)
    got = to_np(inv_lambda_table(ops, pad=3))
    assert_parity(got, want.astype(f32), "invLambda", exact=True)


@pytest.mark.parametrize("table", [False, True], ids=["inline", "table"])
@pytest.mark.parametrize("res,pad", [((64, 64, 64), 0), ((64, 64, 64), 3), ((30, 22, 18), 0),
                                     ((36, 20, 28), 0)])
def test_integrate_sequence(oracle, ops, dev, res, pad, table):
    vox = 2.56 / max(res)
    (tsdf, wts), (d_t, d_w) = integrate_both(oracle, ops, dev, res, vox, BG["pose"], range(3),
                                             pad=pad, table=table)
    assert (wts > 0).sum() > 1000 and (tsdf == -1).sum() > 10
    assert_parity(to_np(d_t), tsdf, f"tsdf {res}", exact=True)
    assert_parity(to_np(d_w), wts, f"weights {res}", exact=True)


def test_integrate_association_weights_cap_and_zero_sum(oracle, ops, dev):
    rng = np.random.default_rng(11)

    def assoc(i, ids):
        a = rng.uniform(0, 1, (H, W)).astype(np.float32)
        a[ids == 1] = 0.0  # w + a == 0 on first touch: voxel must stay untouched
        return a

    (tsdf, wts), (d_t, d_w) = integrate_both(oracle, ops, dev, (64, 64, 64), 0.04, BG["pose"],
                                             range(5), assoc_fn=assoc, max_w=2.5)
    assert wts.max() == 2.5
    assert_parity(to_np(d_t), tsdf, "tsdf", exact=True)
    assert_parity(to_np(d_w), wts, "weights", exact=True)


def test_integrate_camera_behind_and_rotated_volume(oracle, ops, dev):
    # volume partly behind the camera (pos_cam.z <= 0 branch) and rotated against the camera
    pose = Pose(rot([1, 2, 0.5], 25), [0.1, -0.05, 0.6])
    (tsdf, wts), (d_t, d_w) = integrate_both(oracle, ops, dev, (48, 40, 56), 0.04, pose, range(2))
    assert_parity(to_np(d_t), tsdf, "tsdf", exact=True)
    assert_parity(to_np(d_w), wts, "weights", exact=True)


# ---- a8 ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("res", [(64, 64, 64), (30, 22, 18)])
def test_tsdf_grads(oracle, ops, dev, res):
    rng = np.random.default_rng(4)
    tsdf = rng.uniform(-1, 1, (res[2], res[1], res[0])).astype(np.float32)
    want = oracle.compute_tsdf_grads(tsdf)
    g = dev_full(tsdf.shape + (3,), 5.0)  # last planes must be overwritten with 0
    ops.compute_tsdf_grads(to_dev(tsdf, dev), g)
    assert_parity(to_np(g), want, "grads", exact=True)


# ---- a2 ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("ch", [1, 2, 3])
def test_get_volume_vals(oracle, ops, dev, ch):
    rng = np.random.default_rng(20 + ch)
    n = (40, 32, 36)
    vol = rng.standard_normal((n[2], n[1], n[0]) + ((ch,) if ch > 1 else ())).astype(np.float32)
    cam, depth, _ = frame(1)
    pts = oracle.compute_points(depth, K)
    co = rel_CO(cam, Pose(rot([0, 1, 0], 12), [0.1, 0, 1.4]))
    want = oracle.get_volume_vals(vol, pts, co.R32, co.t32, 0.03)
    assert (want != 0).mean() > 0.05 and (want == 0).mean() > 0.01
    vals = dev_full(want.shape, 9.0)  # callee zero-fills
    ops.get_volume_vals(to_dev(vol, dev), to_dev(pts, dev), co.R32, co.t32, 0.03, vals)
    assert_parity(to_np(vals), want, f"vals ch={ch}", exact=True)


# ---- a10 / a11 -----------------------------------------------------------------------------------

_RCP = {}


def checked_reciprocal(ops, vox):
    """emf_hip_voxelReciprocal, once per voxel size."""
    key = float(np.float32(vox))
    if key not in _RCP:
        _RCP[key] = ops.voxel_reciprocal(key)
    return _RCP[key]


def test_voxel_reciprocal_is_checked_and_exact(ops, dev):
    for vox in (0.01, 0.02, 0.04, 2 * 0.3 / 128, 0.0123456):
        r = checked_reciprocal(ops, vox)
        assert r == float(np.float32(1) / np.float32(vox)), vox  # accepted: the plain reciprocal
    from emfusion_amd._lib import EmfHipError
    with pytest.raises(EmfHipError):
        ops.voxel_reciprocal(0.0)


def test_short_reciprocal_check_gives_the_exhaustive_verdict(ops, dev):
    """The product's check sweeps 3 x 2^23 inputs (abi_common.hip, k_check_reciprocal); its verdict must be the one of
    the sweep over all 2^32 bit patterns -- for the sizes the configurations use, the objects' 2 * extent / N, the
    edges of the accepted range, a thousand random sizes, log-uniform over it, and the divisors whose mantissa is all ones
    (the known exception of the Markstein correction this form is: its only candidates for a rejection)."""
    rng = np.random.default_rng(0x5EC1)
    sizes = [0.01, 0.02, 0.04, 0.005, 1e-6, 1e3, 1.0, 0.5, 3.0, 0.0123456]
    sizes += [2 * e / n for e in (0.15, 0.3, 0.45, 0.6, 1.2) for n in (64, 128, 256)]
    sizes += list(2 * rng.uniform(0.05, 1.5, 200) / 128)          # spawned / resized objects
    sizes += list(np.exp(rng.uniform(np.log(1e-6), np.log(1e3), 1000)))
    ones = np.array([(e << 23) | m for e in range(108, 136) for m in (0x7fffff, 0x7ffffe, 0x7ffffd)], np.uint32).view(np.float32)
    sizes += [float(v) for v in ones if 1e-6 <= v <= 1e3]
    accepted = rejected = 0
    for v in sizes:
        v = float(np.float32(v))
        r = ops.voxel_reciprocal(v)
        bad = ops.voxel_reciprocal_exhaustive(v)
        assert (r != 0.0) == (bad == 0), (v, r, bad)
        if r != 0.0:
            assert r == float(np.float32(1) / np.float32(v))
            accepted += 1
        else:
            rejected += 1
    assert accepted > 100, (accepted, rejected)
    print(f"reciprocal verdicts: {accepted} accepted, {rejected} rejected of {len(sizes)} sizes")


def _raycast_dev(ops, dev, tsdf, grads, wts, fg, co, vox, ray0=None, stats=False, divide=False):
    """divide=False: the march uses the checked reciprocal of the voxel size (the default of the
    class-level path); True: IEEE divisions.  Both must equal the oracle bit for bit."""
    ray = dev_full((H, W), 0.0) if ray0 is None else to_dev(ray0, dev)
    vert = dev_full((H, W, 3), 0.0)
    nrm = dev_full((H, W, 3), 0.0)
    mask = dev_full((H, W), 0, np.uint8)
    st = dev_full((4,), 0, np.uint64) if stats else None
    ops.raycast_tsdf(to_dev(tsdf, dev), None if grads is None else to_dev(grads, dev),
                     to_dev(wts, dev), None if fg is None else to_dev(fg, dev), ray, vert, nrm,
                     mask, co.R32, co.t32, K, vox, 10 * vox, st,
                     rcp_voxel=0.0 if divide else checked_reciprocal(ops, vox))
    dev.synchronize()
    out = [to_np(ray), to_np(vert), to_np(nrm), to_np(mask)]
    return out + [to_np(st)] if stats else out


@pytest.fixture(scope="module")
def bg_state(oracle):
    tsdf, wts = alloc_vol(BG["n"]), alloc_vol(BG["n"])
    for i in range(4):
        cam, depth, _ = frame(i)
        oc = rel_OC(cam, BG["pose"])
        oracle.update_tsdf(depth, np.ones((H, W), np.float32), tsdf, wts, oc.R32, oc.t32, K,
                           BG["vox"], 10 * BG["vox"], MAXW)
    return tsdf, wts


CAMS = {
    "tracked": camera_path(4),
    "rotated": Pose(rot([0.3, 1, 0.2], 14), [0.2, -0.1, 0.15]),
    "inside": Pose(rot([0, 1, 0], -8), [0.0, 0.0, 0.5]),
    "outside_oblique": Pose(rot([1, 0, 0], 20), [0.0, -0.9, -0.3]),
}


@pytest.mark.parametrize("divide", [False, True], ids=["reciprocal", "divide"])
@pytest.mark.parametrize("cam_name", list(CAMS))
@pytest.mark.parametrize("use_grad_volume", [False, True])
def test_raycast_background(oracle, ops, dev, bg_state, cam_name, use_grad_volume, divide):
    tsdf, wts = bg_state
    grads = oracle.compute_tsdf_grads(tsdf) if use_grad_volume else None
    co = rel_CO(CAMS[cam_name], BG["pose"])
    want = oracle.raycast_tsdf(tsdf, grads, wts, None, W, H, co.R32, co.t32, K, BG["vox"],
                               10 * BG["vox"], count_steps=True)
    got = _raycast_dev(ops, dev, tsdf, grads, wts, None, co, BG["vox"], stats=True, divide=divide)
    if cam_name != "outside_oblique":
        assert want[3].sum() > 2000
    assert_parity(got[3], want[3], "mask", exact=True)
    assert_parity(got[0], want[0], "raylengths", exact=True)
    assert_parity(got[1], want[1], "vertices", exact=True)
    assert_parity(got[2], want[2], "normals", exact=True)
    assert int(got[4][0]) == int(want[4].sum()), "march sample count (S of the byte model)"
    assert int(got[4][1]) == int(want[3].sum()), "hit count"


def test_raycast_empty_and_unseen_volume(oracle, ops, dev):
    tsdf, wts = alloc_vol((32, 32, 32)), alloc_vol((32, 32, 32))
    co = rel_CO(Pose(), Pose(t=[0, 0, 0.8]))
    want = oracle.raycast_tsdf(tsdf, None, wts, None, W, H, co.R32, co.t32, K, 0.01, 0.1,
                               count_steps=True)
    got = _raycast_dev(ops, dev, tsdf, None, wts, None, co, 0.01, stats=True)
    assert not got[3].any() and not got[0].any()
    assert int(got[4][0]) == int(want[4].sum()) > 0


def test_raycast_respects_previous_raylength(oracle, ops, dev, bg_state):
    # non-zero incoming raylengths clip the march (TSDF.cu:496-500): hits behind them vanish
    tsdf, wts = bg_state
    co = rel_CO(CAMS["tracked"], BG["pose"])
    ray0 = np.zeros((H, W), np.float32)
    ray0[:, : W // 2] = 1.0
    want = oracle.raycast_tsdf(tsdf, None, wts, None, W, H, co.R32, co.t32, K, BG["vox"],
                               10 * BG["vox"], raylengths=ray0)
    got = _raycast_dev(ops, dev, tsdf, None, wts, None, co, BG["vox"], ray0=ray0)
    assert want[3][:, : W // 2].sum() < want[3][:, W // 2:].sum()
    for g, w_, name in zip(got, want, ["ray", "vert", "normal", "mask"]):
        assert_parity(g, w_, name, exact=True)


def test_raycast_object_with_foreground_mask(oracle, ops, dev):
    cen, r = SPHERES[0]
    res, size = (32, 32, 32), 0.8
    vox = size / 32
    pose = Pose(t=cen)
    tsdf, wts, fgbg = alloc_vol(res), alloc_vol(res), alloc_vol(res, 2)
    for i in range(3):
        cam, depth, ids = frame(i)
        oc = rel_OC(cam, pose)
        oracle.update_tsdf(depth, np.ones((H, W), np.float32), tsdf, wts, oc.R32, oc.t32, K, vox,
                           10 * vox, MAXW)
        oracle.update_fgbg_probs((ids == 1).astype(np.uint8), np.zeros((H, W), np.uint8), tsdf,
                                 wts, fgbg, oc.R32, oc.t32, K, vox)
    probs, vmask = oracle.compute_fg_probs(fgbg)
    assert 0 < (vmask > 0).sum() < vmask.size
    co = rel_CO(camera_path(3), pose)
    want = oracle.raycast_tsdf(tsdf, None, wts, vmask, W, H, co.R32, co.t32, K, vox, 10 * vox)
    got = _raycast_dev(ops, dev, tsdf, None, wts, vmask, co, vox)
    assert want[3].sum() > 200
    for g, w_, name in zip(got, want, ["ray", "vert", "normal", "mask"]):
        assert_parity(g, w_, name, exact=True)
    # the literal reference form (masked weights volume) gives the same image
    d_masked = dev_full(wts.shape, 0.0)
    ops.mask_raycast_weights(to_dev(wts, dev), to_dev(vmask, dev), d_masked)
    assert_parity(to_np(d_masked), oracle.mask_raycast_weights(wts, vmask), "raycastWeights",
                  exact=True)
    got2 = _raycast_dev(ops, dev, tsdf, None, to_np(d_masked), None, co, vox)
    for g, w_, name in zip(got2, want, ["ray", "vert", "normal", "mask"]):
        assert_parity(g, w_, name + " (masked volume)", exact=True)


# ---- a13 / a14 -----------------------------------------------------------------------------------

@pytest.mark.parametrize("res", [(32, 32, 32), (30, 22, 18)])
def test_fgbg_counts_and_fg_probs(oracle, ops, dev, bg_state, res):
    cen, r = SPHERES[0]
    vox = 0.8 / max(res)
    pose = Pose(rot([0, 0, 1], 10), cen)
    tsdf, wts, fgbg = alloc_vol(res), alloc_vol(res), alloc_vol(res, 2)
    d_fgbg = to_dev(fgbg, dev)
    rng = np.random.default_rng(8)
    for i in range(3):
        cam, depth, ids = frame(i)
        oc = rel_OC(cam, pose)
        oracle.update_tsdf(depth, np.ones((H, W), np.float32), tsdf, wts, oc.R32, oc.t32, K, vox,
                           10 * vox, MAXW)
        mask = (ids == 1).astype(np.uint8) * (1 if i % 2 else 255)  # any non-zero is "true"
        occl = (rng.random((H, W)) < 0.2).astype(np.uint8)
        oracle.update_fgbg_probs(mask, occl, tsdf, wts, fgbg, oc.R32, oc.t32, K, vox)
        ops.update_fgbg_probs(to_dev(mask, dev, 3), to_dev(occl, dev, 3), to_dev(tsdf, dev),
                              to_dev(wts, dev), d_fgbg, oc.R32, oc.t32, K, vox)
    assert fgbg[..., 0].max() >= 2 and fgbg[..., 1].max() >= 2
    assert_parity(to_np(d_fgbg), fgbg, "fgBgProbs", exact=True)
    probs, vmask = oracle.compute_fg_probs(fgbg)
    d_probs = dev_full(probs.shape, 3.0)
    d_mask = dev_full(probs.shape, 7, np.uint8)
    ops.compute_fg_probs(d_fgbg, d_probs, d_mask)
    assert_parity(to_np(d_probs), probs, "fgProbs", exact=True)
    assert_parity(to_np(d_mask), vmask, "fgVolMask", exact=True)


# ---- a3-a6 ---------------------------------------------------------------------------------------

def _object_state(oracle, k, res=(32, 32, 32)):
    cen, r = SPHERES[k]
    vox = 0.8 / res[0]
    pose = Pose(rot([0, 1, 0], 5 * k), cen)
    tsdf, wts, fgbg = alloc_vol(res), alloc_vol(res), alloc_vol(res, 2)
    for i in range(3):
        cam, depth, ids = frame(i)
        oc = rel_OC(cam, pose)
        oracle.update_tsdf(depth, np.ones((H, W), np.float32), tsdf, wts, oc.R32, oc.t32, K, vox,
                           10 * vox, MAXW)
        oracle.update_fgbg_probs((ids == k + 1).astype(np.uint8), np.zeros((H, W), np.uint8), tsdf,
                                 wts, fgbg, oc.R32, oc.t32, K, vox)
    probs, vmask = oracle.compute_fg_probs(fgbg)
    return dict(tsdf=tsdf, wts=wts, probs=probs, vmask=vmask, pose=pose, vox=vox)


def test_estep_background_and_objects(oracle, ops, dev, bg_state):
    tsdf, wts = bg_state
    objs = [_object_state(oracle, 0), _object_state(oracle, 1)]
    cam, depth, _ = frame(4)
    pts = oracle.compute_points(depth, K)
    d_pts = to_dev(pts, dev)
    models = [dict(tsdf=tsdf, probs=None, pose=BG["pose"], vox=BG["vox"])] + objs
    raw, d_maps = [], []
    for m in models:
        co = rel_CO(cam, m["pose"])
        raw.append(oracle.compute_association(m["tsdf"], m["probs"], pts, co.R32, co.t32, m["vox"],
                                              10 * m["vox"], SIGMA, ALPHA, PRIOR))
        out = dev_full((H, W), 9.0)
        ops.compute_association(to_dev(m["tsdf"], dev),
                                None if m["probs"] is None else to_dev(m["probs"], dev), d_pts,
                                co.R32, co.t32, m["vox"], 10 * m["vox"], SIGMA, ALPHA, PRIOR, out)
        d_maps.append(out)
    for k, (dm, r) in enumerate(zip(d_maps, raw)):
        # expf differs by an ulp or two between glibc and the device library
        assert_parity(to_np(dm), r, f"un-normalised association {k}", rtol=2e-6)
        assert np.array_equal(to_np(dm) == 0, r == 0), "association mask (exact-zero lookups)"
    assert (raw[0] == 0).any() and (raw[1] > 0.25).any()
    want = [r.copy() for r in raw]
    norm = oracle.normalize_association(want)
    d_norm = dev_full((H, W), 0.0)
    ops.normalize_association(d_maps, norm=d_norm)
    assert_parity(to_np(d_norm), norm, "associationNorm", rtol=2e-6)
    for k, (dm, wv) in enumerate(zip(d_maps, want)):
        assert_parity(to_np(dm), wv, f"association weights {k}", rtol=RTOL_ASSOC)
    total = sum(to_np(dm).astype(np.float64) for dm in d_maps)
    valid = norm != 0
    assert np.allclose(total[valid], 1.0, atol=1e-6) and np.all(total[~valid] == 0)


RTOL_ASSOC = 4e-6


@pytest.mark.parametrize("nmaps", [1, 5, 16, 17, 40])
def test_normalize_exact_and_chunked(oracle, ops, dev, nmaps):
    """The normalisation itself (sequential sum + x/0:=0 divide) is IEEE-exact: bit parity,
    including the multi-launch path for more than 16 models."""
    rng = np.random.default_rng(nmaps)
    maps = [rng.uniform(0, 3, (H, W)).astype(np.float32) for _ in range(nmaps)]
    for m in maps:
        m[:7] = 0
    d_maps = [to_dev(m, dev, 2 if i % 2 else 0) for i, m in enumerate(maps)]
    norm = oracle.normalize_association(maps)
    d_norm = dev_full((H, W), 0.0)
    ops.normalize_association(d_maps, norm=d_norm)
    assert_parity(to_np(d_norm), norm, "norm", exact=True)
    for k in range(nmaps):
        assert_parity(to_np(d_maps[k]), maps[k], f"map {k}", exact=True)


def test_sum_and_normalize_with_remote_partial(oracle, ops, dev):
    """Multi-GPU split of the normaliser: each rank sums its own object maps, the
    partials are all-reduced, and every rank normalises [background, own objects] with
    nsum = 1 and extraSum = the reduced object sum."""
    rng = np.random.default_rng(3)
    bg = rng.uniform(0, 2, (H, W)).astype(np.float32)
    mine = [rng.uniform(0, 2, (H, W)).astype(np.float32) for _ in range(2)]
    theirs = [rng.uniform(0, 2, (H, W)).astype(np.float32) for _ in range(20)]
    for m in [bg] + mine + theirs:
        m[:5] = 0
    d_sum_mine = dev_full((H, W), 0.0)
    d_sum_theirs = dev_full((H, W), 0.0)
    ops.sum_association([to_dev(m, dev) for m in mine], d_sum_mine)
    ops.sum_association([to_dev(m, dev) for m in theirs], d_sum_theirs)  # 20 maps: chunked path
    seq_mine = mine[0] + mine[1]
    seq_theirs = theirs[0].copy()
    for m in theirs[1:]:
        seq_theirs = seq_theirs + m
    assert_parity(to_np(d_sum_mine), seq_mine, "local partial", exact=True)
    assert_parity(to_np(d_sum_theirs), seq_theirs, "remote partial", exact=True)
    reduced = to_dev(to_np(d_sum_mine) + to_np(d_sum_theirs), dev)  # what the all-reduce delivers
    d_maps = [to_dev(m, dev) for m in [bg] + mine]
    d_norm = dev_full((H, W), 0.0)
    ops.normalize_association(d_maps, extra_sum=reduced, norm=d_norm, nsum=1)
    nrm = bg + (seq_mine + seq_theirs)
    assert_parity(to_np(d_norm), nrm, "norm with reduced object sum", exact=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        for k, m in enumerate([bg] + mine):
            want = np.where(nrm != 0, m / nrm, 0).astype(np.float32)
            assert_parity(to_np(d_maps[k]), want, f"map {k}", exact=True)
    # and it stays within a few ulp of the single-GPU sequential order
    allmaps = [bg.copy()] + [m.copy() for m in mine + theirs]
    oracle.normalize_association(allmaps)
    assert_parity(to_np(d_maps[1]), allmaps[1], "vs sequential order", rtol=1e-6)


# ---- a12 -----------------------------------------------------------------------------------------

@pytest.mark.parametrize("nobj", [0, 2, 19])
def test_composite_and_visibility(oracle, ops, dev, nobj):
    rng = np.random.default_rng(40 + nobj)
    ids = list(range(1, nobj + 1))
    if nobj >= 2:
        ids[0], ids[1] = 29, 24  # list order is creation order, not id order
    obj_seg = [(rng.random((H, W)) < 0.3).astype(np.uint8) for _ in ids]
    obj_ray = [(rng.uniform(0.5, 3, (H, W)).astype(np.float32) * s) for s in obj_seg]
    if nobj >= 2:
        obj_ray[1][:20] = obj_ray[0][:20]  # ties: the earlier object keeps the pixel
    obj_vert = [rng.standard_normal((H, W, 3)).astype(np.float32) for _ in ids]
    obj_norm = [rng.standard_normal((H, W, 3)).astype(np.float32) for _ in ids]
    bg_mask = (rng.random((H, W)) < 0.8).astype(np.uint8)
    bg_ray = rng.uniform(0.5, 3, (H, W)).astype(np.float32) * bg_mask
    bg_vert = rng.standard_normal((H, W, 3)).astype(np.float32)
    bg_norm = rng.standard_normal((H, W, 3)).astype(np.float32)
    diff0 = (rng.uniform(-1, 1, (H, W)) * (rng.random((H, W)) < 0.3)).astype(np.float32)
    diff = diff0.copy()
    want = oracle.composite_raycast(ids, obj_ray, obj_vert, obj_norm, obj_seg, bg_ray, bg_vert,
                                    bg_norm, bg_mask, diff, 10)
    d = lambda a: to_dev(a, dev)
    ray = dev_full((H, W), 5.0)
    vert = dev_full((H, W, 3), 5.0)
    nrm = dev_full((H, W, 3), 5.0)
    seg = dev_full((H, W), 5, np.uint8)
    no_obj = dev_full((H, W), 5, np.uint8)
    d_diff = d(diff0)
    vis = dev_full((max(nobj, 1),), -1, np.int32)
    ops.composite_raycast(ids, [d(a) for a in obj_ray], [d(a) for a in obj_vert],
                          [d(a) for a in obj_norm], [d(a) for a in obj_seg], d(bg_ray), d(bg_vert),
                          d(bg_norm), d(bg_mask), ray, vert, nrm, seg, d_diff, no_obj, 10, vis)
    dev.synchronize()
    names = ["ray", "vert", "norm", "seg", "noObj"]
    for g, w_, name in zip([ray, vert, nrm, seg, no_obj], want[:5], names):
        assert_parity(to_np(g), w_, name, exact=True)
    assert_parity(to_np(d_diff), diff, "diffRaylengths", exact=True)
    if nobj:
        assert to_np(vis)[:nobj].tolist() == want[5].tolist()
        assert want[5].sum() > 0
    # the fused pair (the composite's launch counts, the flag launch clears): same images, same numbers
    ray2, vert2, nrm2 = dev_full((H, W), 5.0), dev_full((H, W, 3), 5.0), dev_full((H, W, 3), 5.0)
    seg2, no_obj2, d_diff2 = dev_full((H, W), 5, np.uint8), dev_full((H, W), 5, np.uint8), d(diff0)
    vis2 = dev_full((max(nobj, 1),), 0, np.int32)
    mirror = dev_full((max(nobj, 1),), -7, np.int32)
    visible = dev_full((nobj + 1,), -7, np.int32)
    thresh = int(np.median(want[5])) if nobj else 0
    for _ in range(2):  # twice: the counts are cleared behind the first call
        ops.composite_visibility(ids, [d(a) for a in obj_ray], [d(a) for a in obj_vert], [d(a) for a in obj_norm],
                                 [d(a) for a in obj_seg], d(bg_ray), d(bg_vert), d(bg_norm), d(bg_mask), ray2, vert2,
                                 nrm2, seg2, d_diff2, no_obj2, 10, vis2, thresh, visible, mirror)
        dev.synchronize()
        for g, w_, name in zip([ray2, vert2, nrm2, seg2, no_obj2], want[:5], names):
            assert_parity(to_np(g), w_, name + " (fused)", exact=True)
        assert to_np(visible)[0] == 1
        if nobj:
            assert to_np(mirror)[:nobj].tolist() == want[5].tolist()
            assert to_np(visible)[1:].tolist() == [int(c > thresh) for c in want[5]]
            assert not to_np(vis2)[:nobj].any()
        d_diff2 = d(diff0)


def test_occluded_mask(oracle, ops, dev):
    rng = np.random.default_rng(6)
    obj_seg = (rng.random((H, W)) < 0.5).astype(np.uint8)
    seg = rng.integers(0, 4, (H, W)).astype(np.uint8)
    occ = dev_full((H, W), 9, np.uint8)
    ops.occluded_mask(to_dev(obj_seg, dev, 1), to_dev(seg, dev), 2, occ)
    assert_parity(to_np(occ), oracle.occluded_mask(obj_seg, seg, 2), "occluded", exact=True)


# ---- error behaviour on the device side ----------------------------------------------------------

def test_device_info_reports_gfx950(ops):
    name, arch, cus = ops.device_info()
    assert "gfx950" in arch and cus >= 200, (name, arch, cus)
