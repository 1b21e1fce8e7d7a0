"""The two arithmetic shortcuts of the tiled integration (emfusion_amd/csrc/device_core.hpp, EMF_INT_FAST,
on by default) proven rather than sampled through scenes:

  * round_pixel<true>: the pixel a voxel projects to (reference TSDF.cu:360-361, __float2int_rn of an
    IEEE quotient) as round(x * v_rcp_f32(z)), the division kept only for quotients within 2^-21 |q| of
    a rounding tie;
  * band_decision<true>: the side of the truncation band (TSDF.cu:380-400) from v_sqrt_f32, the IEEE
    square root and the division by truncdist kept only within (|d| + |t|) 2^-20 of +-truncdist.

Both rest on "v_rcp_f32 / v_sqrt_f32 are accurate to 1 ulp".  That premise is swept over ALL 2^32 inputs on
the device; the margins derived from it are then attacked with quotients placed within a few ulp of every
tie k + 1/2 and distances within a few ulp of +-truncdist, comparing the shortcut with the IEEE form of the
very device functions the kernels call."""
import ctypes as C

import numpy as np
import pytest

from tests.parity_util import dev_full, to_dev, to_np

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def lib(dev):
    from emfusion_amd import _lib
    return _lib.load()


def test_reciprocal_and_square_root_are_within_one_ulp_for_every_input(lib, dev):
    out = dev_full((4,), 1, np.uint64)
    assert lib.emf_hip_sweepFastPathPremises(C.c_void_p(out.ptr), None) == 0
    bad_rcp, bad_sqrt, worst_rcp, worst_sqrt = (int(v) for v in to_np(out))
    worst_rcp = float(np.array([worst_rcp], np.uint32).view(f32)[0])
    worst_sqrt = float(np.array([worst_sqrt], np.uint32).view(f32)[0])
    # every z with 2^-126 <= |z| <= 2^126 (all of them: 2 x 252 x 2^23 + 2 inputs), every n >= 2^-126
    assert bad_rcp == 0, (bad_rcp, worst_rcp)
    assert bad_sqrt == 0, (bad_sqrt, worst_sqrt)
    assert 0 < worst_rcp <= 2.0 ** -23 and 0 <= worst_sqrt <= 2.0 ** -23, (worst_rcp, worst_sqrt)


def _ulp_steps(x, steps):
    """x moved by `steps` units in the last place (both float32 arrays of one shape)."""
    i = x.view(np.int32).astype(np.int64)
    i = np.where(i < 0, np.int64(-2 ** 31) - i, i)  # monotone integer line
    i = i + steps
    i = np.where(i < 0, np.int64(-2 ** 31) - i, i)
    return i.astype(np.int32).view(f32)


def _pixels(lib, num, den):
    n = num.size
    d_num, d_den = to_dev(num.reshape(-1)), to_dev(den.reshape(-1))
    fast, exact = dev_full((n,), -777, np.int32), dev_full((n,), -777, np.int32)
    assert lib.emf_hip_debugPixelRounding(C.c_void_p(d_num.ptr), C.c_void_p(d_den.ptr), n, C.c_void_p(fast.ptr),
                                          C.c_void_p(exact.ptr), None) == 0
    return to_np(fast), to_np(exact)


def test_pixel_rounding_next_to_every_tie(lib, dev):
    rng = np.random.default_rng(8)
    ks = np.arange(-4, 1300, dtype=np.float64) + 0.5          # every tie of a 1280-pixel row, and a few outside
    zs = np.concatenate([rng.uniform(0.05, 40.0, 700), 2.0 ** rng.integers(-6, 6, 60),
                         rng.uniform(1e-3, 1e4, 40)]).astype(f32)
    steps = np.arange(-4, 5, dtype=np.int64)
    num = (ks[:, None] * zs[None, :].astype(np.float64)).astype(f32)              # x with x / z ~ k + 1/2
    num = _ulp_steps(num[:, :, None].repeat(len(steps), 2), steps[None, None, :])  # ... and its neighbours
    den = np.broadcast_to(zs[None, :, None], num.shape).copy()
    assert num.size > 9e6
    fast, exact = _pixels(lib, num, den)
    q32 = (num.reshape(-1) / den.reshape(-1)).astype(f32)
    assert np.array_equal(exact, np.rint(q32).astype(np.int32))
    assert np.array_equal(fast, exact), f"{int((fast != exact).sum())} quotients round differently"
    # the attack is real: a good part of the quotients sit exactly on a tie or within an ulp of one
    assert (np.abs(q32.astype(np.float64) - (np.floor(q32) + 0.5)) <= np.abs(q32) * 2.0 ** -22).mean() > 0.3


def test_pixel_rounding_random_and_special_operands(lib, dev):
    rng = np.random.default_rng(9)
    n = 1 << 22
    num = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 6, n)).astype(f32)
    den = (10 ** rng.uniform(-3, 3, n)).astype(f32)
    # special numerators over every kind of denominator INSIDE the premise's domain 2^-126 <= z <= 2^126 -- the
    # camera-frame depth of a voxel of a finite volume; beyond it (z = 3e38: the reciprocal is a denormal, which
    # v_rcp_f32 flushes) the shortcut is not claimed, and 0 / denormal z make the quotient non-finite or huge,
    # which quotient_is_risky() hands to the division
    sp = np.array([0.0, -0.0, 1e-45, 1e-38, 1.17549435e-38, 3e38, np.inf, -np.inf, np.nan, 0.5, 1.5, 2.5, -0.5, 2 ** 20,
                   2 ** 20 + 0.5, 2 ** 23, 2 ** 31, -2 ** 31, 1e30, 8e37], f32)
    dn = np.array([0.0, 1e-45, 1e-38, 1.17549435e-38, 2.0 ** -126, 1e-30, 0.5, 1.5, 2.5, 2 ** 20, 2 ** 23, 1e30, 8e37,
                   2.0 ** 126], f32)
    a, b = np.meshgrid(sp, dn)
    num, den = np.concatenate([num, a.reshape(-1)]), np.concatenate([den, b.reshape(-1)])
    fast, exact = _pixels(lib, num, den)
    bad = np.flatnonzero(fast != exact)
    assert bad.size == 0, [(num[i], den[i], fast[i], exact[i]) for i in bad[:8]]


def test_band_decision_next_to_the_truncation_distance(lib, dev):
    rng = np.random.default_rng(10)
    steps = np.arange(-6, 7, dtype=np.int64)
    for trunc in (f32(0.1), f32(0.05), f32(0.32), f32(0.0234375)):
        m = 120000
        il = rng.uniform(0.78, 1.0, m).astype(f32)               # 1 / |(u, v, 1)| over a 60-degree field of view
        n2 = (rng.uniform(0.2, 12.0, m).astype(f32)) ** 2
        t = (il.astype(np.float64) * np.sqrt(n2.astype(np.float64)))
        side = rng.choice([-1.0, 1.0, 0.0, 0.999, -0.999], m)    # both edges of the band, its middle, just inside
        d0 = (t + side * float(trunc)).astype(f32)
        d = _ulp_steps(d0[:, None].repeat(len(steps), 1), steps[None, :]).reshape(-1)
        il_r, n2_r = il.repeat(len(steps)), n2.repeat(len(steps))
        d = np.concatenate([d, rng.uniform(0.1, 12.0, m).astype(f32)])  # and plain free space / far behind
        il_r, n2_r = np.concatenate([il_r, il]), np.concatenate([n2_r, n2])
        n = d.size
        bufs = [to_dev(x) for x in (d, il_r, n2_r)]
        kf, ke = dev_full((n,), -1, np.int32), dev_full((n,), -1, np.int32)
        sf, se = dev_full((n,), 9.0), dev_full((n,), 9.0)
        assert lib.emf_hip_debugBandDecision(*[C.c_void_p(b.ptr) for b in bufs], n, C.c_float(trunc), C.c_void_p(kf.ptr),
                                             C.c_void_p(sf.ptr), C.c_void_p(ke.ptr), C.c_void_p(se.ptr), None) == 0
        kf, ke, sf, se = to_np(kf), to_np(ke), to_np(sf), to_np(se)
        assert np.array_equal(kf, ke), (float(trunc), int((kf != ke).sum()))
        assert sf.tobytes() == se.tobytes(), float(trunc)
        kinds = set(np.unique(ke).tolist())
        assert {2, 3, 3 | 16} <= kinds, kinds  # behind the band, free space, inside the band: all occur
        # the attack is real: many of the distances are decided by the last bits of the IEEE square root
        sdf = d[: m * len(steps)].astype(np.float64) - il_r[: m * len(steps)].astype(np.float64) * np.sqrt(n2_r[: m * len(steps)].astype(np.float64))
        assert (np.abs(np.abs(sdf) - float(trunc)) < 4e-7).mean() > 0.05
