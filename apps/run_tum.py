#!/usr/bin/env python3
"""The reference's main loop (apps/EM-Fusion.cpp:139-156) for a TUM RGB-D sequence with preprocessed
Mask R-CNN results (BASELINE.json configs[2]: `config/tum.cfg`, `--preproc-masks`), on the
MI355X-native classes:

    python apps/run_tum.py /data/rgbd_dataset_freiburg3_walking_xyz/ --masks /data/masks/ --out results/

Every frame: depth PNG (/5000) -> bilateral pre-filter -> E-step / LM-ICP tracking of camera and
objects / E-step -> raycast -> (every maskRCNNFrames-th frame: match the instance masks to the
models, spawn volumes for unmatched ones) -> weighted integration -> mask integration -> clean-up.
Writes poses-cam.txt / poses-<id>.txt (TUM format) and the volume dumps.  Needs an MI355X and a
staged dataset; neither the sequence nor the masks ship with this repository.
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("sequence", help="TUM sequence directory (associations.txt, depth/*.png)")
    ap.add_argument("--cofusion", nargs=2, metavar=("COLORDIR", "DEPTHDIR"),
                    help="the sequence is a Co-Fusion style dataset (reference ImageReader): "
                         "<sequence>/COLORDIR/ColorNNNN.png, <sequence>/DEPTHDIR/DepthNNNN.exr")
    ap.add_argument("--intrinsics", nargs=4, type=float, metavar=("FX", "FY", "CX", "CY"),
                    help="camera intrinsics (default: 525 px focal length scaled to the image width)")
    ap.add_argument("--ignore-person", action="store_true",
                    help="Params.ignore_person of config/tum.cfg: person objects stay out of renderings and meshes")
    ap.add_argument("--masks", help="directory with Mask%%04d.plk files of the reference's preprocessing")
    ap.add_argument("--out", default="emfusion_out")
    ap.add_argument("--frames", type=int, default=0, help="0 = all")
    ap.add_argument("--bg-res", type=int, default=512)
    ap.add_argument("--bg-voxel", type=float, default=0.01)
    ap.add_argument("--obj-res", type=int, default=128)
    ap.add_argument("--volumes", action="store_true", help="also dump the TSDF volumes")
    ap.add_argument("--visibility-thresh", type=int, default=0, help="0 = 1600 scaled by the image area")
    ap.add_argument("--mask-frames", type=int, default=30, help="Mask R-CNN every n-th frame (maskRCNNFrames)")
    args = ap.parse_args()

    import torch  # noqa: F401  (one HIP runtime, see bench.py)
    from emfusion_amd import pipeline, readers
    from emfusion_amd.devmem import DeviceArray
    from emfusion_amd.ops import image_view

    if args.cofusion:
        reader = readers.ImageReader(args.sequence, *args.cofusion)
        index0 = reader.first
    else:
        reader = readers.TUMReader(args.sequence)
        index0 = 0
    n = len(reader) if args.frames <= 0 else min(args.frames, len(reader))
    first = reader.depth(index0)
    h, w = first.shape
    scale = w / 640.0
    prm = pipeline.make_params(w, h, args.bg_res, args.bg_voxel, args.obj_res,
                               visibility_thresh=args.visibility_thresh or int(round(1600 * scale * scale)),
                               boundary=int(round(20 * scale)), mask_frames=args.mask_frames)
    if args.intrinsics:
        fx, fy, cx, cy = args.intrinsics
        prm.K[:] = [fx, 0, cx, 0, fy, cy, 0, 0, 1]
    fus = pipeline.Fusion(prm, None)
    fus.set_ignore_person(args.ignore_person)
    fus.set_preprocess(True)
    fus.set_cleanup(True)
    fus.setup_output(False, args.volumes)  # EMFusion::setupOutput of the reference app (apps/EM-Fusion.cpp:112)
    eye, zero = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)
    t0 = time.time()
    for f in range(n):
        depth = np.ascontiguousarray(reader.depth(index0 + f), np.float32)
        depth[~np.isfinite(depth)] = 0
        d = DeviceArray.from_numpy(depth)
        keep = [d]
        if args.masks and f % prm.mask_frames == 0:
            plk = Path(args.masks) / f"Mask{f:04d}.plk"  # numbered by frameCount (EMFusion.cpp:384-386)
            if plk.exists():
                _, masks, scores = readers.load_preprocessed_masks(plk)
                dev_masks = [DeviceArray.from_numpy(m) for m in masks]
                keep += dev_masks
                fus.queue_instance_masks([image_view(m) for m in dev_masks])
                fus.queue_instance_scores(scores)
        if f == 1:
            fus.set_tracking(camera=True, objects=True)  # frame 0 defines the world frame
        fus.process_frame(image_view(d), eye, zero, {}, {}, False)
        fus.synchronize()
        if f % 50 == 0:
            r = fus.track_result(0) if f else None
            print(f"frame {f}/{n}: objects {sorted(fus.visible_objects())}"
                  + (f", camera LM steps {r['iterations']} ({r['accepted']} accepted)" if r else ""),
                  flush=True)
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    fus.write_results(out, volumes=args.volumes)
    print(f"{n} frames in {time.time() - t0:.1f} s (incl. PNG decoding on the host); results in {out}/")
    fus.close()


if __name__ == "__main__":
    main()
