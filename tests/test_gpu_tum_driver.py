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
    from emfusion_amd import pipeline, readers
    W, H, N = 160, 120, 6
    prm = pipeline.make_params(W, H, 64, 0.04, 32)
    synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), 2, seed=0xE3F5)
    seq, masks = tmp_path / "seq", tmp_path / "masks"
    (seq / "depth").mkdir(parents=True)
    masks.mkdir()
    lines, truth = [], []
    for f in range(N):
        depth, sid = synth.render(f)
        truth.append(synth.camera_pose(f)[1])
        readers.write_png_gray16(seq / "depth" / f"{f:04d}.png", np.round(depth * 5000).astype(np.uint16))
        lines.append(f"{f / 30:.6f} rgb/{f:04d}.png {f / 30:.6f} depth/{f:04d}.png")
        if f % 2 == 0:
            m = [(sid == 1).astype(np.uint8), (sid == 2).astype(np.uint8)]  # generate_result's lists
            with open(masks / f"Mask{f:04d}.plk", "wb") as fh:
                pickle.dump(([[0, 0, 1, 1]] * 2, m, np.zeros((2, 81)).tolist()), fh, protocol=2)
    (seq / "associations.txt").write_text("\n".join(lines) + "\n")
    synth.close()
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, str(ROOT / "apps" / "run_tum.py"), str(seq) + "/", "--masks", str(masks),
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
