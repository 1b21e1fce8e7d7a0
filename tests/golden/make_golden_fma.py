"""The kernel vectors once more with a*b+c contraction ON ("each with both FMA
settings so the test tolerance is data-driven").

PROVENANCE: outputs of THIS repository's CPU oracle built with `-ffp-contract=fast -mfma`
(oracle/libemf_oracle_fma.so) on the inputs of kernels_v1.npz -- the same source with the contraction nvcc
applies to the reference by default (which no other compiler reproduces instruction for instruction; gcc's
choice of what to fuse is one sample of it).  Not reference outputs: the reference cannot be built here.
Only the OUTPUT arrays are stored; the inputs are those of kernels_v1.npz (same generator, same seeds).

Usage (repo root, CPU only):  python tests/golden/make_golden_fma.py  ->  tests/golden/kernels_fma_v1.npz
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import binding as orc  # noqa: E402
from tests.golden import make_golden  # noqa: E402

OUTPUTS = ("tsdf0", "wts0", "tsdf2", "wts2", "grads", "vals1", "vals3", "fgbg", "probs", "vmask", "points") + tuple(
    f"{k}{j}{s}" for j in range(3) for s in (("", "_fg") if j == 0 else ("",)) for k in ("ray", "vert", "nrm", "hit", "steps"))

if __name__ == "__main__":
    orc.lib(True)
    orc.set_threads(4, fma=True)
    plain = dict(np.load(make_golden.OUT / "kernels_v1.npz"))
    g = make_golden.kernels(fma=True, save=False)
    out = {}
    for tag in ("cube", "ragged"):
        for k in OUTPUTS:
            out[f"{tag}_{k}"] = g[f"{tag}_{k}"]
        for k in g:  # the inputs must be those of kernels_v1.npz: nothing of them may depend on the build
            if k.startswith(tag) and k[len(tag) + 1:] not in OUTPUTS:
                assert np.array_equal(g[k], plain[k]), k
    np.savez_compressed(make_golden.OUT / "kernels_fma_v1.npz", **out)
    n_diff = sum(int(not np.array_equal(out[k], plain[k])) for k in out)
    print("kernels_fma_v1.npz", (make_golden.OUT / "kernels_fma_v1.npz").stat().st_size, "bytes;", len(out),
          "arrays,", n_diff, "differ from the contraction-off vectors")
