"""Helpers for the GPU parity tests: host<->device movement and the acceptance metric."""
from __future__ import annotations

import numpy as np

# north_star tolerance: outputs within 1e-4 relative of the reference path.
RTOL = 1e-4


def to_dev(a: np.ndarray, dev=None, pad_cols: int = 0):
    """numpy -> device array (product HIP runtime).  pad_cols > 0 allocates padded rows
    (pitch > width * elemsize, padding poisoned) to exercise pitched images."""
    from emfusion_amd.devmem import DeviceArray
    return DeviceArray.from_numpy(a, pad_cols)


def dev_full(shape, value, dtype=np.float32, pad_cols: int = 0):
    from emfusion_amd.devmem import DeviceArray
    return DeviceArray.full(shape, value, dtype, pad_cols)


def to_np(t) -> np.ndarray:
    """Synchronise the device, then copy to the host."""
    return t.numpy()


def mismatch(a: np.ndarray, b: np.ndarray, rtol: float = RTOL, atol: float = 0.0) -> np.ndarray:
    """Boolean map of elements violating |a-b| <= rtol*max(|a|,|b|) + atol (NaN == NaN passes)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    with np.errstate(invalid="ignore"):
        bad = np.abs(a - b) > rtol * np.maximum(np.abs(a), np.abs(b)) + atol
    bad |= np.isnan(a) ^ np.isnan(b)
    return bad & ~both_nan


def assert_parity(got, want, what: str, rtol: float = RTOL, atol: float = 0.0,
                  budget: float = 0.0, exact: bool = False):
    """got (HIP) vs want (oracle).  `budget` = allowed fraction of elements outside tolerance.
    exact=True demands bit-identical arrays (integer / mask / index outputs, and float kernels
    whose arithmetic is IEEE-exact on both sides)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    if exact:
        same = (got == want) | (np.isnan(got.astype(np.float64)) & np.isnan(want.astype(np.float64)))
        n = int((~same).sum())
        assert n == 0, f"{what}: {n} of {same.size} elements differ bitwise (first at " \
                       f"{tuple(np.argwhere(~same)[0])}: {got[~same][0]!r} vs {want[~same][0]!r})"
        return 1.0
    bad = mismatch(got, want, rtol, atol)
    frac = bad.mean() if bad.size else 0.0
    exact_frac = float(np.mean((got == want) | (np.isnan(got) & np.isnan(want)))) if got.size else 1.0
    assert frac <= budget, (f"{what}: {int(bad.sum())} of {bad.size} elements ({frac:.3%}) outside "
                            f"rtol={rtol} atol={atol}; worst abs diff "
                            f"{np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64)))}")
    return exact_frac
