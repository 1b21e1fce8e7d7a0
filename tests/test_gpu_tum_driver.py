"""apps/run_tum.py end to end on a TUM-format sequence staged from the synthetic stream: depth PNGs
(x 5000), associations.txt and Mask%04d.plk files go in; tracked poses and volume dumps come out."""
import pickle
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_run_tum_on_a_staged_synthetic_sequence(tmp_path, dev):
    from tests import tum_staging as T
    seq_dir, masks, truth = T.stage(tmp_path)
    N = T.N
    from pathlib import Path as _P
    seq = _P(seq_dir)
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, str(ROOT / "apps" / "run_tum.py"), seq_dir, "--masks", str(masks),
                        "--out", str(out), "--bg-res", "64", "--bg-voxel", "0.04", "--obj-res", "32",
                        "--visibility-thresh", "100", "--mask-frames", "2", "--volumes"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [line.split() for line in (out / "poses-cam.txt").read_text().strip().splitlines()]
    assert [int(x[0]) for x in rows] == list(range(N))
    # the driver's world frame is the first camera: compare camera MOTION with the stream's
    est = np.array([[float(v) for v in x[1:4]] for x in rows])
    assert np.abs(est[0]).max() == 0
    assert np.linalg.norm(est[-1]) < 0.05  # the synthetic camera moves on a 5 cm circle
    assert (out / "poses-1.txt").exists() and (out / "poses-2.txt").exists()  # spawned from the masks
    assert (out / "tsdfs" / "bg_tsdf.bin").stat().st_size == 24 + 64 ** 3 * 4
    assert "objects [1, 2]" in r.stdout or "objects [1]" in r.stdout or "objects []" in r.stdout
