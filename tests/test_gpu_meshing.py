"""Marching cubes: emf_hip_meshCount / emf_hip_meshEmit against the oracle's restatement
of cuda::TSDF::marchingCubes -- same vertices, normals and triangles, element for element."""
import numpy as np
import pytest

from tests.parity_util import to_dev
from tests.scenes import Pose, camera_path, intrinsics, render_depth, rel_OC

pytestmark = pytest.mark.gpu

W, H = 160, 120
K = intrinsics(W, H)
SPHERES = [((0.25, 0.05, 1.3), 0.22), ((-0.3, -0.1, 1.6), 0.18)]


@pytest.fixture(scope="module")
def ops(dev):
    from emfusion_amd import ops
    return ops


def fused_volume(oracle, res, vox, pose, frames=3, seed=50):
    f32 = np.float32
    nx, ny, nz = res
    tsdf, wts = np.zeros((nz, ny, nx), f32), np.zeros((nz, ny, nx), f32)
    for i in range(frames):
        cam = camera_path(i)
        depth, _ = render_depth(W, H, K, cam, SPHERES, noise=0.002, dropout=0.01, seed=seed + i)
        oc = rel_OC(cam, pose)
        oracle.update_tsdf(depth, np.ones((H, W), f32), tsdf, wts, oc.R32, oc.t32, K, vox, 10 * vox, 64.0)
    return tsdf, wts


def same_mesh(got, want):
    for g, w, name in zip(got, want, ("vertices", "normals", "triangles")):
        assert g.shape == w.shape, (name, g.shape, w.shape)
        assert g.tobytes() == w.tobytes(), name


@pytest.mark.parametrize("res", [(32, 32, 32), (30, 22, 37), (64, 48, 40)])
def test_mesh_equals_the_reference_restatement(oracle, ops, dev, res):
    vox = 0.64 / res[0]
    tsdf, wts = fused_volume(oracle, res, vox, Pose(t=SPHERES[0][0]))
    want = oracle.marching_cubes(tsdf, wts, vox)
    assert len(want[0]) > 500 and len(want[2]) > 200
    same_mesh(ops.extract_mesh(to_dev(tsdf), to_dev(wts), vox), want)
    # every triangle refers to vertices of its own cube: indices in range, none degenerate by index
    tri = want[2]
    assert np.all(tri[:, 0] == 3) and tri[:, 1:].min() >= 0 and tri[:, 1:].max() < len(want[0])
    # normals are the raw interpolated gradients (Q19): not unit length
    n = np.linalg.norm(want[1], axis=1)
    assert np.isfinite(n).all() and np.abs(n - 1).max() > 0.5


def test_mesh_with_foreground_mask_and_gradient_volume(oracle, ops, dev):
    res, vox = (40, 36, 32), 0.016
    tsdf, wts = fused_volume(oracle, res, vox, Pose(t=SPHERES[0][0]))
    rng = np.random.default_rng(8)
    fg = (rng.uniform(size=tsdf.shape) < 0.93).astype(np.uint8) * 255
    grads = oracle.compute_tsdf_grads(tsdf)
    want = oracle.marching_cubes(tsdf, wts, vox, fg=fg)
    assert 100 < len(want[0]) < len(oracle.marching_cubes(tsdf, wts, vox)[0])
    same_mesh(ops.extract_mesh(to_dev(tsdf), to_dev(wts), vox, fg_mask=to_dev(fg)), want)
    # the materialised gradient volume gives the same normals as the on-the-fly differences
    same_mesh(ops.extract_mesh(to_dev(tsdf), to_dev(wts), vox, fg_mask=to_dev(fg), grads=to_dev(grads)), want)
    same_mesh(oracle.marching_cubes(tsdf, wts, vox, fg=fg, grads=grads), want)
    # a gradient volume that is NOT the forward difference is used as given
    g2 = (grads * np.float32(3)).astype(np.float32)
    got = ops.extract_mesh(to_dev(tsdf), to_dev(wts), vox, fg_mask=to_dev(fg), grads=to_dev(g2))
    same_mesh(got, oracle.marching_cubes(tsdf, wts, vox, fg=fg, grads=g2))


def test_empty_and_degenerate_volumes(oracle, ops, dev):
    f32 = np.float32
    z = np.zeros((8, 8, 8), f32)
    v, n, t = ops.extract_mesh(to_dev(z), to_dev(z), 0.01)          # nothing observed
    assert v.shape == (0, 3) and t.shape == (0, 4)
    ones = np.ones((8, 8, 8), f32)
    v, n, t = ops.extract_mesh(to_dev(ones), to_dev(ones), 0.01)    # observed, no sign change
    assert v.shape == (0, 3) and t.shape == (0, 4)
    v, n, t = ops.extract_mesh(to_dev(-ones), to_dev(ones), 0.01)   # all inside (class 255)
    assert v.shape == (0, 3) and t.shape == (0, 4)
    # exact zeros on voxels: vertexInterp's |val| < 1e-5 branches return the corner itself
    plane = np.zeros((6, 6, 6), f32)
    plane[:, :, :3] = -0.5
    plane[:, :, 3] = 0.0
    plane[:, :, 4:] = 0.5
    want = oracle.marching_cubes(plane, np.ones_like(plane), 0.02)
    assert len(want[0]) > 0
    same_mesh(ops.extract_mesh(to_dev(plane), to_dev(np.ones_like(plane)), 0.02), want)
    tiny = np.array([[[-1, 1], [1, 1]], [[1, 1], [1, 1]]], f32)     # 2^3: a single cube
    want = oracle.marching_cubes(tiny, np.ones_like(tiny), 1.0)
    assert want[0].shape == (3, 3) and want[2].tolist() == [[3, 0, 2, 1]]
    same_mesh(ops.extract_mesh(to_dev(tiny), to_dev(np.ones_like(tiny)), 1.0), want)


def test_mesh_is_watertight_inside_the_observed_region(oracle, ops, dev):
    """Geometric sanity independent of the oracle: welded by position, every edge of the surface of a
    fully observed analytic sphere is shared by exactly two triangles."""
    f32 = np.float32
    n, vox = 24, 0.05
    c = (np.arange(n, dtype=f32) - f32(n - 1) / 2) * f32(vox)
    zz, yy, xx = np.meshgrid(c, c, c, indexing="ij")
    sdf = (np.sqrt(xx * xx + yy * yy + zz * zz) - f32(0.37)).astype(f32)
    v, nrm, t = ops.extract_mesh(to_dev(sdf), to_dev(np.ones_like(sdf)), vox)
    assert abs(np.linalg.norm(v, axis=1) - 0.37).max() < 0.01
    _, weld = np.unique(np.round(v / 1e-5).astype(np.int64), axis=0, return_inverse=True)
    tri = weld.reshape(-1)[t[:, 1:]]
    e = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
    e = e[e[:, 0] != e[:, 1]]
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert np.all(cnt == 2)
    # normals point outwards (gradient of a distance field), un-normalised: |g| ~ voxel size
    cosang = np.einsum("ij,ij->i", nrm, v) / (np.linalg.norm(nrm, axis=1) * np.linalg.norm(v, axis=1))
    assert cosang.min() > 0.9 and abs(np.linalg.norm(nrm, axis=1).mean() - vox) < 0.2 * vox
