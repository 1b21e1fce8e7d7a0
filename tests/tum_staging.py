"""Stages a TUM-format sequence from the synthetic stream for the two sequence drivers (apps/run_tum.py,
apps/emfusion_synth --sequence): depth PNGs (x 5000), associations.txt, Mask%04d.plk files in the layout the reference's
preprocessing writes (apps/maskrcnn.in.py:188-206).  Test infrastructure; no dataset ships with the repository."""
import pickle

import numpy as np

W, H, N, MASK_EVERY = 160, 120, 6, 2
SMALL = ["--bg-res", "64", "--bg-voxel", "0.04", "--obj-res", "32", "--visibility-thresh", "100", "--mask-frames", str(MASK_EVERY)]


def stage(tmp_path):
    """Returns (sequence dir with trailing slash, mask dir, ground-truth camera translations)."""
    from emfusion_amd import pipeline, readers
    prm = pipeline.make_params(W, H, 64, 0.04, 32)
    synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), 2, seed=0xE3F5)
    seq, masks = tmp_path / "seq", tmp_path / "masks"
    (seq / "depth").mkdir(parents=True)
    masks.mkdir()
    lines, truth = [], []
    for f in range(N):
        depth, sid = synth.render(f)
        truth.append(synth.camera_pose(f)[1])
        readers.write_png_gray16(seq / "depth" / f"{f:04d}.png", np.round(depth * 5000).astype(np.uint16),
                                 filters=np.arange(H) % 5)  # adaptive filters, as libpng writes them
        lines.append(f"{f / 30:.6f} rgb/{f:04d}.png {f / 30:.6f} depth/{f:04d}.png")
        if f % MASK_EVERY == 0:
            seg = np.stack([sid == 1, sid == 2], axis=2)  # (H, W, N) bool, sliced like generate_result does
            with open(masks / f"Mask{f:04d}.plk", "wb") as fh:
                pickle.dump(([[0, 0, 1, 1]] * 2, [seg[:, :, 0], seg[:, :, 1]], np.zeros((2, 81)).tolist()), fh,
                            protocol=pickle.HIGHEST_PROTOCOL if f else 2)
    (seq / "associations.txt").write_text("\n".join(lines) + "\n")
    synth.close()
    return str(seq) + "/", str(masks), truth
