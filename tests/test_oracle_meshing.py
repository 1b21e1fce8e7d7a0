"""Known-answer properties of the oracle's marching cubes (oracle/emf_oracle.c, f-4) -- CPU only."""
import numpy as np


def sphere_sdf(n, vox, r):
    f32 = np.float32
    c = (np.arange(n, dtype=f32) - f32(n - 1) / 2) * f32(vox)
    zz, yy, xx = np.meshgrid(c, c, c, indexing="ij")
    return (np.sqrt(xx * xx + yy * yy + zz * zz) - f32(r)).astype(f32)


def test_single_cube_known_answer(oracle):
    f32 = np.float32
    tiny = np.array([[[-1, 1], [1, 1]], [[1, 1], [1, 1]]], f32)  # corner 0 inside: class 1 -> edges 0, 8, 3
    v, n, t = oracle.marching_cubes(tiny, np.ones_like(tiny), 1.0)
    assert t.tolist() == [[3, 0, 2, 1]]  # triTable[1] = {0, 8, 3}: vertex slots in edge-bit order 0, 3, 8
    # vertices in edge-bit order: edge 0 (x), edge 3 (z), edge 8 (y); mid-points since |-1| = |1|
    assert np.array_equal(v, np.array([[0, -.5, -.5], [-.5, -.5, 0], [-.5, 0, -.5]], f32))
    # raw forward-difference gradient at corner 0 is (2, 2, 2); the other corners lie on last planes
    # (gradient 0): the interpolated, un-normalised normal is half of it (Q19)
    assert np.array_equal(n, np.full((3, 3), 1.0, f32))


def test_masked_cubes_produce_nothing(oracle):
    sdf = sphere_sdf(12, 0.1, 0.35)
    w = np.ones_like(sdf)
    full = oracle.marching_cubes(sdf, w, 0.1)
    w[:, :, 6:] = 0  # unobserved half
    half = oracle.marching_cubes(sdf, w, 0.1)
    assert 0 < len(half[0]) < len(full[0]) and half[0][:, 0].max() <= 0.0 + 1e-6
    fg = np.zeros(sdf.shape, np.uint8)
    assert len(oracle.marching_cubes(sdf, np.ones_like(sdf), 0.1, fg=fg)[0]) == 0


def test_sphere_is_closed_and_oriented(oracle):
    vox, r = 0.05, 0.37
    sdf = sphere_sdf(24, vox, r)
    v, nrm, t = oracle.marching_cubes(sdf, np.ones_like(sdf), vox)
    assert abs(np.linalg.norm(v, axis=1) - r).max() < 0.01
    _, weld = np.unique(np.round(v / 1e-5).astype(np.int64), axis=0, return_inverse=True)
    tri = weld.reshape(-1)[t[:, 1:]]
    e = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
    e = e[e[:, 0] != e[:, 1]]
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert np.all(cnt == 2)  # watertight
    # Euler characteristic of a sphere
    V = len(np.unique(weld))
    good = (tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])
    assert V - len(cnt) + int(good.sum()) == 2
    # triangle winding agrees with the gradient (outward) direction
    a, b, c = v[t[:, 1]], v[t[:, 2]], v[t[:, 3]]
    face = np.cross(b - a, c - a)
    s = np.sign(np.einsum("ij,ij->i", face, a + b + c))
    assert abs(s[np.linalg.norm(face, axis=1) > 1e-9].mean()) > 0.99
