"""One rank of a full-size rehearsal of BASELINE.json configs[3] (8 x MI355X: 64 object volumes 128^3 sharded
8 per GPU + a replicated 512^3 background, 640 x 480) with ALL ranks sharing the one GPU of the box:

    python -m torch.distributed.run --nproc-per-node 8 tests/rehearsal_worker.py --out DIR [--frames 6]

one process per rank, as on a node; the ranks' receive buffers are mapped into each other with hipIpc and the
exchanges are the direct peer-write ones (emf::makePeerCommunicator; waitInFront mode: a one-wave wait in front
of every consumer).  Every rank writes DIR/rank<r>.npz: digests of its background replica and of the joint images,
its visible set, and -- rank 0 only -- the joint images themselves.  `--world 1` runs the same 64-object scene in
ONE process without a communicator as the reference of the comparison: on the batched path, three chunks of the model
table (round 6; the per-volume path, which this run fell back to through round 5, with EMF_PER_VOLUME=1).
Test infrastructure (tests/test_gpu_config3_rehearsal.py); SURVEY.md 8(e)."""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

W, H, BG, VOX, OBJ = 640, 480, 512, 0.01, 128


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--objects", type=int, default=64)
    ap.add_argument("--world", type=int, default=0, help="1: single process, no communicator")
    args = ap.parse_args()
    import torch  # noqa: F401  (first: one HIP runtime per process, emfusion_amd/devmem.py)
    import xxhash
    dist = None
    rank, world = 0, 1
    if args.world != 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
    from emfusion_amd import devmem, pipeline
    from emfusion_amd.devmem import DeviceArray
    from emfusion_amd.ops import image_view
    devmem.set_device(0)
    prm = pipeline.make_params(W, H, BG, VOX, OBJ)
    K = np.array(prm.K, np.float32)
    synth = pipeline.SyntheticStream(W, H, K, args.objects, seed=0xE3F5)
    comm = pipeline.Communicator.peer(dist, W * H * 16) if dist is not None else None
    fus = pipeline.Fusion(prm, comm)
    if comm is not None:
        fus.set_depth_broadcast(0)
    ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(args.objects)]
    mine = [i for i in ids if fus.owns_object(i)]
    keep, vis = [], []
    for f in range(args.frames):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        # with the broadcast on, only rank 0 holds the frame: the others start from zeros
        d = DeviceArray.from_numpy(depth if rank == 0 else np.zeros_like(depth))
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in mine}
        masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in mine} if f == 0 else {}
        keep += [d, masks]
        fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, f == 0)
        fus.synchronize()
        vis.append(sorted(fus.visible_objects()))

    def dg(a):
        return xxhash.xxh3_128(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()
    out = dict(rank=rank, world=world, mine=np.array(mine), visible=np.array(vis[-1]), chunks=fus.batched_chunks(),
               visible_per_frame=np.array([",".join(map(str, v)) for v in vis]))
    joint = {k: fus.image(k) for k in ("segmentation", "raylengths", "bg_raylengths", "bg_assoc", "assoc_norm")}
    for k, a in joint.items():
        out["digest_" + k] = dg(a)
    out["digest_bg_tsdf"] = dg(fus.volume("tsdf", 0))
    out["digest_bg_weights"] = dg(fus.volume("weights", 0))
    if rank == 0:
        out.update({"img_" + k: a for k, a in joint.items()})
        w = fus.volume("weights", 0)
        out["bg_seen"] = int((w > 0).sum())
        out["bg_weights_sum"] = float(w.sum(dtype=np.float64))
        out["bg_tsdf_sample"] = fus.volume("tsdf", 0)[::8, ::8, ::8].copy()
    for i in mine[:2]:
        out[f"digest_obj{i}_tsdf"] = dg(fus.volume("tsdf", i))
    Path(args.out).mkdir(parents=True, exist_ok=True)
    np.savez(Path(args.out) / f"rank{rank}.npz", **out)
    fus.close()
    synth.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
