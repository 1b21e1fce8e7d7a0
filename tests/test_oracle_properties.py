"""Known-answer and property checks that pin the CPU oracle.

The reference ships no tests or golden vectors for this path and cannot be built in this image
(parity unpinned, see oracle/emf_oracle.h), so the oracle is anchored on what the algorithm must
produce analytically: exact values on linear fields, closed-form SDF of a fronto-parallel plane,
the half-voxel overshoot of the reference's hit interpolation, normalisation
identities, and agreement between equivalent formulations (gradient volume vs forward differences,
masked weights volume vs in-gather foreground mask).
"""
import numpy as np
import pytest

from tests.scenes import Pose, intrinsics, rel_CO, rel_OC, render_depth, rot

W, H = 160, 120
K = intrinsics(W, H)
N = 64
VOX = 0.04  # 2.56 m cube
TRUNC = 10 * VOX
VOL_POSE = Pose(t=[0, 0, N * VOX / 2])


def _integrate_plane(oracle, depth_m=1.5, frames=1, cam=Pose(), assoc=None, n=N, vox=VOX):
    depth = np.full((H, W), depth_m, np.float32)
    assoc = np.ones((H, W), np.float32) if assoc is None else assoc
    tsdf = np.zeros((n, n, n), np.float32)
    wts = np.zeros((n, n, n), np.float32)
    oc = rel_OC(cam, Pose(t=[0, 0, n * vox / 2]))
    for _ in range(frames):
        oracle.update_tsdf(depth, assoc, tsdf, wts, oc.R32, oc.t32, K, vox, 10 * vox, 64.0)
    return tsdf, wts


def test_compute_points_formula(oracle):
    rng = np.random.default_rng(1)
    depth = rng.uniform(0, 4, (H, W)).astype(np.float32)
    depth[rng.random((H, W)) < 0.1] = 0
    pts = oracle.compute_points(depth, K)
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    ex = ((xs - K[0, 2]) * depth) / K[0, 0]
    ey = ((ys - K[1, 2]) * depth) / K[1, 1]
    assert np.array_equal(pts[..., 0], ex.astype(np.float32))
    assert np.array_equal(pts[..., 1], ey.astype(np.float32))
    assert np.array_equal(pts[..., 2], depth)


def test_integrate_plane_matches_closed_form_sdf(oracle):
    D = 1.5
    tsdf, wts = _integrate_plane(oracle, D)
    # voxel centres in camera frame (identity camera): z = (k - (N-1)/2) * vox + N*vox/2
    k = np.arange(N, dtype=np.float64)
    z = (k - (N - 1) / 2) * VOX + N * VOX / 2
    xy = (k - (N - 1) / 2) * VOX
    Z, Y, X = np.meshgrid(z, xy, xy, indexing="ij")
    u = K[0, 0] * X / Z + K[0, 2]
    v = K[1, 1] * Y / Z + K[1, 2]
    inside = (np.rint(u) >= 0) & (np.rint(u) < W) & (np.rint(v) >= 0) & (np.rint(v) < H)
    # projective SDF of a fronto-parallel plane is (D - z) up to the nearest-pixel rounding of
    # lambda, which perturbs it by < 1 % of z here
    expect = np.clip((D - Z) / TRUNC, -1, 1)
    front = inside & ((D - Z) >= -TRUNC * 0.98)
    assert front.sum() > 10000
    assert np.max(np.abs(tsdf[front] - expect[front])) < 0.03
    assert np.all(wts[front] == 1.0)
    # behind the truncation band: never observed -> tsdf = -1, weight 0 (TSDF.cu:398-400)
    behind = inside & ((D - Z) < -TRUNC * 1.02)
    assert np.all(tsdf[behind] == -1.0) and np.all(wts[behind] == 0.0)
    # outside the image: untouched
    assert np.all(tsdf[~inside] == 0.0) and np.all(wts[~inside] == 0.0)


def test_weight_cap_and_running_average(oracle):
    tsdf1, _ = _integrate_plane(oracle, 1.5, frames=1)
    tsdf70, w70 = _integrate_plane(oracle, 1.5, frames=70)
    seen = w70 > 0
    assert w70.max() == 64.0 and np.all(w70[seen] == 64.0)
    # averaging identical samples leaves the value unchanged up to rounding
    assert np.max(np.abs(tsdf70[seen] - tsdf1[seen])) < 1e-5


def test_zero_association_keeps_surface_voxels_unobserved_but_carves_free_space(oracle):
    # Q8: voxels with sdf >= truncdist fuse with weight 1 regardless of association
    assoc = np.zeros((H, W), np.float32)
    tsdf, wts = _integrate_plane(oracle, 1.5, assoc=assoc)
    assert np.all(tsdf[wts > 0] == 1.0)
    tsdf_full, wts_full = _integrate_plane(oracle, 1.5)
    band = (wts_full > 0) & (np.abs(tsdf_full) < 1)
    assert band.sum() > 1000 and np.all(wts[band] == 0) and np.all(tsdf[band] == 0)


def test_invalid_depth_resets_unseen_voxels_only(oracle):
    depth = np.zeros((H, W), np.float32)
    assoc = np.ones((H, W), np.float32)
    tsdf = np.full((N, N, N), 0.25, np.float32)
    wts = np.zeros((N, N, N), np.float32)
    wts[:, :, ::2] = 3.0
    oc = rel_OC(Pose(), VOL_POSE)
    t0 = tsdf.copy()
    oracle.update_tsdf(depth, assoc, tsdf, wts, oc.R32, oc.t32, K, VOX, TRUNC, 64.0)
    changed = tsdf != t0
    assert changed.any()
    assert np.all(tsdf[changed] == 0) and np.all(wts[changed] == 0)
    assert not changed[:, :, ::2].any()


def test_gradients_of_linear_field(oracle):
    z, y, x = np.meshgrid(np.arange(20), np.arange(24), np.arange(28), indexing="ij")
    tsdf = (0.5 * x + 0.25 * y - 0.125 * z).astype(np.float32)  # exactly representable
    g = oracle.compute_tsdf_grads(tsdf)
    assert g.shape == (20, 24, 28, 3)
    inner = g[:-1, :-1, :-1]
    assert np.all(inner[..., 0] == 0.5) and np.all(inner[..., 1] == 0.25)
    assert np.all(inner[..., 2] == -0.125)
    assert np.all(g[-1] == 0) and np.all(g[:, -1] == 0) and np.all(g[:, :, -1] == 0)  # Q10


def test_trilinear_lookup_is_exact_on_linear_field(oracle):
    n = 32
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    vol = (0.5 * x + 0.25 * y - 0.125 * z + 3).astype(np.float32)
    rng = np.random.default_rng(2)
    pts = rng.uniform(-0.5, 0.5, (H, W, 3)).astype(np.float32)
    pts[..., 2] += 1.0
    vox = 0.05
    co = rel_CO(Pose(rot([0, 1, 0], 10), [0.05, 0, 0]), Pose(t=[0, 0, 1.0]))
    vals = oracle.get_volume_vals(vol, pts, co.R32, co.t32, vox)
    p = pts.reshape(-1, 3).astype(np.float64) @ co.R.T + co.t
    v = p / vox + (n - 1) / 2
    ok = np.all((v >= 0) & (v + 1 < n), axis=1)
    expect = 0.5 * v[:, 0] + 0.25 * v[:, 1] - 0.125 * v[:, 2] + 3
    got = vals.reshape(-1)
    safe = np.all((v >= 0.01) & (v + 1.01 < n), axis=1)  # away from the validity boundary
    assert safe.sum() > 5000
    assert np.max(np.abs(got[safe] - expect[safe])) < 1e-4
    assert np.all(got[~ok & ~safe] == 0)


def test_lookup_skips_invalid_points_and_three_channels(oracle):
    n = 16
    rng = np.random.default_rng(3)
    vol3 = rng.standard_normal((n, n, n, 3)).astype(np.float32)
    pts = rng.uniform(-0.3, 0.3, (H, W, 3)).astype(np.float32)
    pts[..., 2] = np.where(rng.random((H, W)) < 0.3, 0.0, pts[..., 2] + 0.5)
    co = rel_CO(Pose(), Pose(t=[0, 0, 0.5]))
    vals3 = oracle.get_volume_vals(vol3, pts, co.R32, co.t32, 0.05)
    assert np.all(vals3[pts[..., 2] <= 0] == 0)
    for c in range(3):
        v1 = oracle.get_volume_vals(np.ascontiguousarray(vol3[..., c]), pts, co.R32, co.t32, 0.05)
        assert np.array_equal(v1, vals3[..., c])


def test_raycast_plane_hits_half_voxel_behind_and_faces_camera(oracle):
    # the reference's t* uses the updated (half-voxel) step, so a fronto-parallel
    # plane at depth D raycasts to about D + voxel/2.
    D = 1.5
    tsdf, wts = _integrate_plane(oracle, D, frames=2)
    grads = oracle.compute_tsdf_grads(tsdf)
    co = rel_CO(Pose(), VOL_POSE)
    ray, vert, nrm, mask = oracle.raycast_tsdf(tsdf, grads, wts, None, W, H, co.R32, co.t32, K,
                                               VOX, TRUNC)
    inner = np.zeros((H, W), bool)
    inner[20:-20, 20:-20] = True
    assert mask[inner].all()
    z = vert[..., 2][inner]
    assert np.all(np.abs(z - (D + VOX / 2)) < 0.35 * VOX)
    n = nrm[inner]
    assert np.all(n[:, 2] < -0.99)
    # vertex = raylength * unit ray direction
    cy, cx = H // 2, W // 2
    assert abs(np.linalg.norm(vert[cy, cx]) - ray[cy, cx]) < 1e-5
    # masks are 0/1 (Q11)
    assert set(np.unique(mask)) <= {0, 1}


def test_raycast_empty_volume_hits_nothing_and_counts_half_voxel_march(oracle):
    n = 32
    tsdf = np.zeros((n, n, n), np.float32)
    wts = np.zeros((n, n, n), np.float32)
    co = rel_CO(Pose(), Pose(t=[0, 0, 1.0]))
    ray, vert, nrm, mask, steps = oracle.raycast_tsdf(tsdf, None, wts, None, W, H, co.R32, co.t32,
                                                      K, 0.01, 0.1, count_steps=True)
    assert not mask.any() and not ray.any()
    # unseen space (tsdf == 0) is marched at half-voxel steps through the whole box
    assert 40 <= steps[H // 2, W // 2] <= 2 * n


def test_raycast_gradient_volume_equals_on_the_fly_differences(oracle):
    depth, _ = render_depth(W, H, K, Pose(), spheres=[((0.1, 0.0, 1.2), 0.3)])
    tsdf = np.zeros((N, N, N), np.float32)
    wts = np.zeros((N, N, N), np.float32)
    assoc = np.ones((H, W), np.float32)
    oc = rel_OC(Pose(), VOL_POSE)
    oracle.update_tsdf(depth, assoc, tsdf, wts, oc.R32, oc.t32, K, VOX, TRUNC, 64.0)
    cam = Pose(rot([0, 1, 0], 4), [0.03, -0.02, 0.01])
    co = rel_CO(cam, VOL_POSE)
    a = oracle.raycast_tsdf(tsdf, oracle.compute_tsdf_grads(tsdf), wts, None, W, H, co.R32,
                            co.t32, K, VOX, TRUNC)
    b = oracle.raycast_tsdf(tsdf, None, wts, None, W, H, co.R32, co.t32, K, VOX, TRUNC)
    assert a[3].sum() > 3000
    for u, v in zip(a, b):
        assert np.array_equal(u, v, equal_nan=True)


def test_raycast_fg_mask_equals_masked_weight_volume(oracle):
    depth, _ = render_depth(W, H, K, Pose(), spheres=[((0.0, 0.0, 1.2), 0.3)])
    tsdf = np.zeros((N, N, N), np.float32)
    wts = np.zeros((N, N, N), np.float32)
    oc = rel_OC(Pose(), VOL_POSE)
    oracle.update_tsdf(depth, np.ones((H, W), np.float32), tsdf, wts, oc.R32, oc.t32, K, VOX,
                       TRUNC, 64.0)
    rng = np.random.default_rng(5)
    fg = (rng.random((N, N, N)) < 0.7).astype(np.uint8) * 255
    fg[:, :, N // 2:] = 0  # a half-space that is not foreground at all
    co = rel_CO(Pose(), VOL_POSE)
    masked = oracle.mask_raycast_weights(wts, fg)
    a = oracle.raycast_tsdf(tsdf, None, masked, None, W, H, co.R32, co.t32, K, VOX, TRUNC)
    b = oracle.raycast_tsdf(tsdf, None, wts, fg, W, H, co.R32, co.t32, K, VOX, TRUNC)
    full = oracle.raycast_tsdf(tsdf, None, wts, None, W, H, co.R32, co.t32, K, VOX, TRUNC)
    assert 0 < a[3].sum() < full[3].sum()
    for u, v in zip(a, b):
        assert np.array_equal(u, v, equal_nan=True)


def test_association_closed_form_and_mask(oracle):
    n = 16
    tsdf = np.full((n, n, n), 0.5, np.float32)
    tsdf[:, :, : n // 2] = 0.0  # exact-zero lookups are invalid (Q6)
    pts = np.zeros((H, W, 3), np.float32)
    pts[..., 2] = 0.4
    pts[:, : W // 2, 0] = -0.2
    pts[:, W // 2:, 0] = 0.2
    pts[0, :, 2] = 0.0  # invalid depth
    co = rel_CO(Pose(), Pose(t=[0, 0, 0.4]))
    sigma, alpha, prior, trunc = 0.02, 0.8, 1.0, 0.1
    out = oracle.compute_association(tsdf, None, pts, co.R32, co.t32, 0.05, trunc, sigma, alpha,
                                     prior)
    expect = alpha * np.exp(-0.5 * trunc / sigma) / (2 * sigma) + (1 - alpha) * prior
    assert np.all(out[0] == 0)
    assert np.all(out[1:, : W // 2] == 0)
    assert np.allclose(out[1:, W // 2:], expect, rtol=1e-6)
    # foreground probability scales the Laplace term only
    fg = np.full((n, n, n), 0.25, np.float32)
    out_fg = oracle.compute_association(tsdf, fg, pts, co.R32, co.t32, 0.05, trunc, sigma, alpha,
                                        prior)
    expect_fg = alpha * 0.25 * np.exp(-0.5 * trunc / sigma) / (2 * sigma) + (1 - alpha) * prior
    assert np.allclose(out_fg[1:, W // 2:], expect_fg, rtol=1e-6)


def test_normalisation_identities(oracle):
    rng = np.random.default_rng(7)
    maps = [rng.uniform(0, 2, (H, W)).astype(np.float32) for _ in range(5)]
    for m in maps:
        m[:10] = 0  # rows where every model is invalid
    raw = [m.copy() for m in maps]
    norm = oracle.normalize_association(maps)
    seq = raw[0].copy()
    for m in raw[1:]:
        seq = seq + m
    assert np.array_equal(norm, seq)
    assert all(np.all(m[:10] == 0) for m in maps)  # x / 0 := 0 (Q7)
    total = np.sum(np.stack(maps), 0)
    assert np.allclose(total[10:], 1.0, atol=1e-6)
    assert np.array_equal(maps[2][10:], raw[2][10:] / seq[10:])


def test_fg_probs(oracle):
    fgbg = np.zeros((4, 4, 4, 2), np.float32)
    fgbg[0, 0, 0] = (3, 1)
    fgbg[0, 0, 1] = (1, 1)
    fgbg[0, 0, 2] = (0, 5)
    probs, mask = oracle.compute_fg_probs(fgbg)
    assert probs[0, 0, 0] == 0.75 and mask[0, 0, 0] == 255
    assert probs[0, 0, 1] == 0.5 and mask[0, 0, 1] == 0  # strict > 0.5
    assert probs[0, 0, 2] == 0 and probs[1, 1, 1] == 0  # 0 / 0 := 0
    assert mask.sum() == 255


def test_update_fgbg_counts(oracle):
    tsdf, wts = _integrate_plane(oracle, 1.5)
    fgbg = np.zeros((N, N, N, 2), np.float32)
    mask = np.zeros((H, W), np.uint8)
    mask[:, : W // 2] = 1
    occl = np.zeros((H, W), np.uint8)
    occl[: H // 4] = 1
    oc = rel_OC(Pose(), VOL_POSE)
    for _ in range(2):
        oracle.update_fgbg_probs(mask, occl, tsdf, wts, fgbg, oc.R32, oc.t32, K, VOX)
    touched = fgbg.sum(-1) > 0
    band = (np.abs(tsdf) < 1) & (wts != 0)
    assert touched.any() and not (touched & ~band).any()
    assert set(np.unique(fgbg[touched].sum(-1))) == {2.0}  # one count per call, fg xor bg
    assert (fgbg[..., 0] > 0).any() and (fgbg[..., 1] > 0).any()
    assert not ((fgbg[..., 0] > 0) & (fgbg[..., 1] > 0)).any()


def test_composite_order_override_and_visibility(oracle):
    h, w = 60, 80
    z = lambda v: np.full((h, w), v, np.float32)
    z3 = lambda v: np.full((h, w, 3), v, np.float32)
    seg_a = np.zeros((h, w), np.uint8); seg_a[:, :50] = 1
    seg_b = np.zeros((h, w), np.uint8); seg_b[:, 30:] = 1
    ray_a, ray_b = z(1.0) * seg_a, z(1.0) * seg_b  # equal depth in the overlap: first wins (Q15)
    ray_b[:, 40:] = 0.5 * seg_b[:, 40:]            # nearer from column 40 on
    bg_ray = z(2.0); bg_ray[:10] = 0.4             # background >= 10 cm in front on the top rows
    bg_mask = np.ones((h, w), np.uint8); bg_mask[-5:] = 0
    diff = np.zeros((h, w), np.float32); diff[-5:] = 1.0  # stale value where bg_mask == 0 (Q12)
    ray, vert, nrm, seg, no_obj, vis = oracle.composite_raycast(
        [3, 7], [ray_a, ray_b], [z3(1), z3(2)], [z3(-1), z3(-2)], [seg_a, seg_b], bg_ray, z3(9),
        z3(-9), bg_mask, diff, 5)
    body = slice(10, h - 5)
    assert np.all(seg[body, :40] == 3) and np.all(seg[body, 40:] == 7)
    assert np.all(seg[:10] == 0)  # background override: composite - bg > 0.05
    assert np.all(ray[:10, :40] == 1.0)  # ... but the composite raylength is not replaced
    assert np.all(seg[-5:] == 0) and np.all(diff[-5:] == 1.0)  # stale diff still overrides
    assert np.all(no_obj[seg == 0] == 255) and np.all(no_obj[seg != 0] == 0)
    assert np.all(vert[seg == 0] == 9) and np.all(vert[seg == 3] == 1) and np.all(vert[seg == 7] == 2)
    inner = np.zeros((h, w), bool); inner[5:-5, 5:-5] = True
    assert vis.tolist() == [int(((seg == 3) & inner).sum()), int(((seg == 7) & inner).sum())]


def test_occluded_mask(oracle):
    obj_seg = np.array([[1, 1, 0, 0]], np.uint8)
    seg = np.array([[4, 2, 4, 0]], np.uint8)
    occ = oracle.occluded_mask(obj_seg, seg, 4)
    assert occ.tolist() == [[0, 1, 0, 0]]


def test_fma_build_differs_only_within_noise_floor(oracle):
    """The contraction-enabled build of the same source stays within the parity tolerance for the
    overwhelming majority of elements: quantifies the floor any
    cross-compiler comparison against the nvcc-built reference would see."""
    depth, _ = render_depth(W, H, K, Pose(), spheres=[((0.1, 0.0, 1.2), 0.3)], noise=0.002, seed=3)
    oc = rel_OC(Pose(), VOL_POSE)
    out = []
    for fma in (False, True):
        tsdf = np.zeros((N, N, N), np.float32)
        wts = np.zeros((N, N, N), np.float32)
        oracle.update_tsdf(depth, np.ones((H, W), np.float32), tsdf, wts, oc.R32, oc.t32, K, VOX,
                           TRUNC, 64.0, fma=fma)
        out.append((tsdf, wts))
    a, b = out[0][0], out[1][0]
    rel = np.abs(a - b) > 1e-4 * np.maximum(np.abs(a), np.abs(b)) + 1e-6
    assert rel.mean() < 0.01
