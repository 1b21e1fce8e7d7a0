#!/usr/bin/env python
"""Headline benchmark: frames/s of EM-Fusion's per-frame volumetric hot path on MI355X.

One "step" = one frame of the schedule emf::EMFusion::processFrame runs:
compute_points + 3 x E-step (association likelihood of every model + normalisation) + raycast of
every model + compositing/visibility + association-weighted TSDF integration of the background
and every visible object (+ fg/bg mask integration on every 30th frame, as in the reference),
on a deterministic synthetic RGB-D stream whose depth maps are resident in HBM before the timed
region starts.  Poses and masks are supplied (tracking and Mask R-CNN are outside this path).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 runs BASELINE.json configs[1]: background 512^3 @ 1 cm + 4 object volumes 128^3, 640x480.
N > 1 (launched by torch.distributed.run, one rank per GPU) is the object-sharded layout of
configs[3]: every rank holds a replica of the background and `--objects-per-gpu` object volumes
of its own; ranks exchange one RCCL all-reduce per E-step (normaliser) and one per raycast
(nearest-hit merge).  Weak scaling: per-GPU work is fixed, the scene grows with N.

The timed region holds the product path plus a HIP-event pair around every 4th launch of the long kernels
(raycast, background integration, tracking stage; --event-stride) and nothing else: the march-sample counters the byte model of `roofline` needs are
collected afterwards, in an untimed replay of the same frames from a cleared state (--no-stats-replay skips it).

Behind the headline, in the same line (bench_extras.py): `target_config`, `steady_state`, `entry_point` (the reference's
processFrame(RGBD) on a staged TUM-layout scene), `strong_scaling` (configs[3]'s 64-object scene, fixed, split over the
ranks) and -- N > 1 -- `rccl` (ranks / devices the transport saw), `sharded_parity` (replicas and joint images against a
single-rank re-run) and `per_rank_roofline`.  An RCCL failure exits non-zero (--allow-fallback: gloo, labelled).

Rank 0 prints ONE JSON line (see README / DESIGN.md for the `roofline` and `cpu_baseline` objects).
torch is used for torch.distributed only (gloo rendezvous, barriers, max-reduce of the time); all
device memory and streams belong to the product's own HIP runtime (emfusion_amd/devmem.py).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--bg-res", type=int, default=512)
    ap.add_argument("--bg-voxel", type=float, default=0.01)
    ap.add_argument("--obj-res", type=int, default=128)
    ap.add_argument("--objects-per-gpu", type=int, default=4)
    ap.add_argument("--grads", choices=["onthefly", "materialize"], default="onthefly",
                    help="surface normals from on-the-fly TSDF differences (default) or from the "
                         "reference's materialised gradient volume (rebuilt every frame)")
    ap.add_argument("--no-stats-replay", action="store_true",
                    help="skip the untimed replay that counts march samples (roofline.achieved of the raycast is "
                         "then null): for profiler passes that should see each launch once")
    ap.add_argument("--track", action="store_true",
                    help="the tracked hot path on an OBSERVABLE scene: the staged TUM-layout sequence (tests/tum_scene.py, "
                         "config/tum.cfg's values, BASELINE.json configs[2]) with depth maps and masks resident in HBM, camera "
                         "and objects tracked (LM-ICP, SURVEY f-1); changes metric and workload -- the headline metric of "
                         "BASELINE.json is measured WITHOUT this flag")
    ap.add_argument("--entry", action="store_true",
                    help="the reference's entry point, EMFusion::processFrame(const RGBD&), on the same staged sequence: host "
                         "depth maps (double-buffered pinned upload), bilateral pre-filter, masks from Mask%%04d.plk, object "
                         "life cycle, camera + object tracking, clean-up: configs[2]'s frames/s")
    ap.add_argument("--track-spheres", action="store_true",
                    help="rounds 2-5's tracked figure: configs[1]'s sphere scene with poses tracked instead of supplied "
                         "(under-constrained: an object stage often burns its whole budget on unobservable motion; a footnote)")
    ap.add_argument("--no-entry", action="store_true", help="skip the `entry_point` sub-line of the default N = 1 run")
    ap.add_argument("--no-strong", action="store_true", help="skip the `strong_scaling` sub-run (configs[3]'s scene, fixed, "
                                                             "split over the ranks)")
    ap.add_argument("--strong-objects", type=int, default=None,
                    help="objects of the strong-scaling scene (default: configs[3]'s 64 when the run has configs[1] / [3]'s "
                         "geometry -- 640x480, background 512^3, objects 128^3 --, none for other geometries unless given)")
    ap.add_argument("--no-parity", action="store_true", help="N > 1: skip `sharded_parity` (replicas / joint images against a "
                                                             "single-rank re-run on rank 0)")
    ap.add_argument("--parity-frames", type=int, default=4)
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if RCCL cannot be brought up, fall back to host-staged collectives over gloo (loudly labelled) "
                         "instead of exiting non-zero")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-target", action="store_true",
                    help="skip the two further measurements the default N = 1 run appends: `target_config` (north_star's "
                         "target configuration, bg 512^3 + 8 x 128^3) and `steady_state` (the headline's workload over "
                         "frames 25..124, a window that holds four mask frames)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target CPU time for the oracle baseline sample")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="bracket EVERY launch with a HIP event pair (full per-kernel table; the event "
                         "records around the ~5 us kernels cost ~5 %% of the frame).  Default: only the "
                         "large kernels (raycast, integrate, tracking) are bracketed")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="bracket every N-th launch of the long kernels with a HIP-event pair (default 4; 1 = every "
                         "launch).  The records sit on the streams they time: all of them cost 2.6 %% of the frame rate")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket kernel launches with HIP events in the timed region")
    ap.add_argument("--no-depth-broadcast", action="store_true",
                    help="multi-GPU: every rank already holds the frame; skip the per-frame "
                         "ncclBroadcast of the depth image from rank 0")
    ap.add_argument("--comm", choices=["rccl", "gloo", "peer"], default="rccl",
                    help="rccl: the product's transport (one GPU per rank).  gloo: rehearsal -- the "
                         "collectives are staged through host memory and carried by torch.distributed, "
                         "so N ranks can share fewer than N GPUs; not a measurement.  peer: the direct peer-write "
                         "exchanges (hipIpc-mapped receive buffers, three small launches per exchange, no library "
                         "collective; handles travel over gloo) -- works with ranks sharing a GPU too")
    ap.add_argument("--force-sharded", action="store_true",
                    help="debug: run the cross-rank exchange path (RCCL all-reduces) even with one "
                         "rank, to exercise the multi-GPU code on a single GPU")
    return ap.parse_args()


def dist_setup(n_gpus):
    """Returns (rank, world, local_rank, dist module or None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and "RANK" not in os.environ:
        if n_gpus != 1:
            raise SystemExit("--gpus N > 1 needs one process per GPU: launch with "
                             "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        return 0, 1, 0, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # gloo: host-side control plane only (rendezvous, barriers, max of the elapsed time); the
    # data-path collectives are RCCL calls issued by the C++ communicator on its own HIP streams
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE is {world}")
    return rank, world, int(os.environ.get("LOCAL_RANK", rank)), dist


def main():
    args = parse_args()
    # torch FIRST: its wheel bundles libamdhip64.so.7 / librccl.so.1 under the same SONAMEs as
    # /opt/rocm, so whichever copy is loaded first serves the whole process.  Importing torch
    # before the product libraries gives one HIP runtime (and one RCCL) for torch.distributed,
    # the kernels and the communicator alike -- see emfusion_amd/devmem.py.
    import torch  # noqa: F401
    rank, world, local_rank, dist = dist_setup(args.gpus)

    from emfusion_amd import devmem, ops, pipeline
    from emfusion_amd.devmem import DeviceArray

    if devmem.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    devmem.set_device(local_rank % devmem.device_count())
    dev_name, arch, cus = ops.device_info()
    if args.track or args.entry:
        # the tracked hot path / the reference's entry point on the staged TUM-layout scene: a line of its own (bench_extras.py)
        if world != 1:
            raise SystemExit("--track / --entry measure one RGB-D stream on one GPU (--gpus 1)")
        import bench_extras
        result = bench_extras.tum_line(args, pipeline, ops, DeviceArray, roofline, workload_key,
                                       f"{dev_name or 'MI355X'} {arch} {cus} CUs", "entry" if args.entry else "track")
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), flush=True)
        return
    args.track = args.track_spheres  # from here on: the sphere scene with free poses (footnote figure)
    if torch.cuda.is_available():
        # torch initialises its CUDA state lazily, on the first torch.cuda call: have that happen HERE and not in
        # the torch.cuda.synchronize() that brackets the timed region (measured, round 4: with the lazy
        # initialisation right in front of it, the first timed frame's enqueue took 30-50 ms of host time in
        # most runs of --steps 60 --warmup 20)
        torch.cuda.set_device(local_rank % devmem.device_count())
        torch.cuda.synchronize()

    W, H = args.width, args.height
    P = W * H
    nobj_total = args.objects_per_gpu * world
    prm = pipeline.make_params(W, H, args.bg_res, args.bg_voxel, args.obj_res,
                               materialize_gradients=(args.grads == "materialize"))
    # visibility threshold / boundary scale with the image area (reference values are for VGA)
    scale = (W / 640.0)
    prm.visibility_thresh = int(round(1600 * scale * scale))
    prm.boundary = int(round(20 * scale))
    K = np.array(prm.K, np.float32)

    comm = None
    if world > 1 or args.force_sharded:
        if args.force_sharded:
            os.environ["EMF_FORCE_SHARDED"] = "1"
        transport = "rccl"
        if args.comm == "gloo" and dist is not None:
            comm = pipeline.Communicator.host_staged(dist)
            transport = "gloo (rehearsal, --comm gloo)"
        elif args.comm == "peer" and dist is None:  # --force-sharded with one rank: what the exchanges themselves cost
            comm = pipeline.Communicator.local_group(1, transport="peer", max_bytes=W * H * 16)[0]
            transport = "peer-write (one rank, --force-sharded)"
        elif args.comm == "peer" and dist is not None:
            comm = pipeline.Communicator.peer(dist, W * H * 16)
            transport = "peer-write (direct stores into hipIpc-mapped peer buffers, --comm peer)" + \
                ("" if devmem.device_count() >= world else "; ranks share a GPU: rehearsal")
        else:
            err = None
            try:
                uid = [pipeline.Communicator.unique_id() if rank == 0 else None]
                if dist is not None:
                    dist.broadcast_object_list(uid, src=0)  # ncclUniqueId travels over the gloo group
                comm = pipeline.Communicator(uid[0], rank, world)
            except Exception as e:  # noqa: BLE001 - reported below, loudly
                err = repr(e)
            if dist is not None:
                # every rank must take the same road: agree on whether RCCL came up everywhere
                flags = [None] * world
                dist.all_gather_object(flags, err)
                bad = [f for f in flags if f]
                if bad and not args.allow_fallback:
                    # a multi-GPU line that silently measured something else is worse than none
                    if comm is not None:
                        comm.close()
                    dist.barrier()
                    dist.destroy_process_group()
                    raise SystemExit(f"bench.py: RCCL communicator could NOT be created ({bad[0]}); refusing to measure "
                                     "over another transport (--allow-fallback runs host-staged collectives over gloo, "
                                     "--comm gloo / --comm peer select a rehearsal transport explicitly)")
                if bad:
                    print(f"bench.py: RCCL communicator could NOT be created ({bad[0]}); falling back to "
                          "host-staged collectives over gloo (--allow-fallback) -- the multi-GPU numbers of this run are NOT "
                          "those of the product's transport", file=sys.stderr, flush=True)
                    if comm is not None:
                        comm.close()
                    comm = pipeline.Communicator.host_staged(dist)
                    transport = "gloo FALLBACK (RCCL init failed: %s)" % bad[0][:120]
            elif err:
                raise SystemExit("bench.py: RCCL communicator could not be created: " + err)

    def synth_factory(n):
        return pipeline.SyntheticStream(W, H, K, n, seed=0xE3F5)
    synth = synth_factory(nobj_total)
    fus = pipeline.Fusion(prm, comm)
    depth_broadcast = comm is not None and not args.no_depth_broadcast
    if depth_broadcast:
        # north_star: "depth maps are broadcast once per frame" -- rank 0's image is the source;
        # the other ranks' (identical, synthetic) copies are overwritten by it inside the timed step
        fus.set_depth_broadcast(0)
    ids = []
    for k in range(nobj_total):
        c, r, vs = synth.sphere(k, 0)
        ids.append(fus.add_object(c, vs))
    mine = [i for i in ids if fus.owns_object(i)]

    # ---- synthetic inputs, resident in HBM before anything is timed ------------------------------
    nframes = args.warmup + args.steps
    mask_every = prm.mask_frames
    t_gen = time.time()
    depth_dev, frames = [], []
    for f in range(nframes):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0])
                 for i in mine}
        run_masks = f % mask_every == 0
        d_masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in mine} \
            if run_masks else {}
        d = DeviceArray.from_numpy(depth)
        depth_dev.append((d, d_masks))
        frames.append(dict(view=ops.image_view(d), R=R, t=t, poses=poses,
                           masks={i: ops.image_view(m) for i, m in d_masks.items()},
                           run_masks=run_masks))
    t_gen = time.time() - t_gen

    def step(f):
        fr = frames[f]
        fus.process_frame(fr["view"], fr["R"], fr["t"], fr["poses"], fr["masks"], fr["run_masks"])

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- warm-up (untimed): populates the volumes, saturates weights ---------------------------
    for f in range(args.warmup):
        step(f)
    fus.synchronize()

    # ---- timed region: exactly K frames, barrier + device sync on both sides ---------------------
    launches_per_frame = 8 * (1 + len(mine)) + 12
    if not args.no_kernel_events:
        fus.kernel_timers_enable(launches_per_frame * args.steps + 64)
        if not args.all_kernel_events:
            fus.kernel_timers_select(["raycast", "integrate_bg", "track"] if fus.background_overlap() else ["raycast", "integrate", "track"])
            fus.kernel_timers_stride(max(1, args.event_stride))
    if args.track:
        fus.set_tracking(camera=True, objects=True)
    barrier()
    fus.synchronize()
    if torch.cuda.is_available() and not os.environ.get("BENCH_NO_TORCH_SYNC"):
        torch.cuda.synchronize()  # same HIP runtime as the product libraries (imported first)
    # The harness is Python, the product is not: a generation-2 garbage collection of the interpreter (torch's import
    # leaves ~10^6 tracked objects) stops the enqueuing thread for 35-45 ms -- measured in round 4 as ONE slow
    # process_frame call in most runs of --steps 60 --warmup 20, i.e. half the frame rate of that run.  Collect now,
    # keep the collector out of the timed region, switch it back on behind it.
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    per_step = []
    for f in range(args.warmup, nframes):
        ts = time.perf_counter()
        step(f)
        per_step.append(time.perf_counter() - ts)
    issued = time.perf_counter() - t0  # the host is done enqueuing; the rest of `elapsed` is the device catching up
    fus.synchronize()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if os.environ.get("BENCH_PER_STEP"):
        print("PER_STEP_US " + " ".join("%.0f" % (1e6 * v) for v in per_step), file=sys.stderr, flush=True)
    kern = None if args.no_kernel_events else fus.kernel_timers_collect()
    visible = fus.visible_objects()
    chunks = fus.batched_chunks()


    track_steps = []

    def replay_with_counters():
        """March samples and hits of exactly the timed launches, counted in an UNTIMED replay (the byte model of
        the raycast needs them; the per-wave counters cost the raycast and the sweep beside it ~13 % when they
        run inside the timed region, as they did until round 2).  The stream and every kernel are
        deterministic: same frames from a cleared state, same launches."""
        if args.no_stats_replay:
            return None
        if not args.no_kernel_events:
            fus.kernel_timers_enable(0)  # no event pairs in the replay
        if args.track:
            fus.set_tracking(camera=False, objects=False)
        fus.reset()
        for k in range(nobj_total):
            c, r, vs = synth.sphere(k, 0)
            assert fus.add_object(c, vs) == ids[k]
        for f in range(nframes):
            if f == args.warmup:
                fus.synchronize()
                fus.enable_raycast_stats(True)
                if args.track:
                    fus.set_tracking(camera=True, objects=True)
            step(f)
            if args.track and f >= args.warmup:  # LM steps of the replayed (= the timed) stages: what frames/s is made of
                res = [fus.track_result(i) for i in [0] + list(mine)]
                track_steps.append((res[0]["iterations"], res[0]["accepted"], max(r["iterations"] for r in res[1:]) if mine else 0))
        fus.synchronize()
        return fus.raycast_stats()

    # the copy-bandwidth probe comes first: its kernel also marks, in a kernel trace of this command, where
    # the measured run ends and the replay begins (scripts/summarize_profile.py)
    copy_gbs = copy_bandwidth(devmem, ops)  # every rank: its own device (the rehearsal's ranks share one)
    l1p = l1_probe(devmem, ops)
    barrier()
    stats = replay_with_counters()
    barrier()

    # ---- every rank prices ITS kernels with the committed profile of its share; rank 0 gathers the summaries ----
    share_key = workload_key(W, H, args.bg_res, args.obj_res, len(mine), args.track)
    my_roof = my_rows = None
    if kern is not None:
        my_roof, my_rows = roofline(kern, stats, P, copy_gbs, share_key, args.steps, l1p)
    per_rank = None
    if comm is not None:
        ray = next((r for r in (my_rows or []) if r["kind"] == "raycast"), None)
        integ = next((r for r in (my_rows or []) if r["kind"] in ("integrate_bg", "integrate")), None)
        brief = {"rank": rank, "objects": len(mine), "workload_key": share_key,
                 "dominant_kernel": my_roof["kernel"] if my_roof else None,
                 "avg_launch_ms": my_roof["avg_launch_ms"] if my_roof else None,
                 "bound": my_roof.get("bound") if my_roof else None, "frac": my_roof.get("frac") if my_roof else None,
                 "hbm_frac": ((my_roof.get("resources") or {}).get("hbm") or {}).get("frac") if my_roof else None,
                 "raycast_ms": ray["avg_ms"] if ray else None, "integrate_ms": integ["avg_ms"] if integ else None,
                 "hbm_copy_GBs": copy_gbs, "counters_from": my_roof.get("counters_from") if my_roof else None}
        per_rank = [brief]
        if dist is not None:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, brief)

    result = None
    if rank == 0:
        fps = args.steps / elapsed
        result = {
            "metric": "frames/sec (integrate+raycast+EM-assoc)" if not args.track else
                      "frames/sec (integrate+raycast+EM-assoc+LM-ICP tracking)",
            "value": round(fps, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "host_issue_ms_per_step": round(1e3 * issued / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + (" (REHEARSAL: collectives over gloo, ranks may share a GPU -- not a "
                                   "measurement)" if comm is not None and (transport.startswith("gloo") or
                                                                             "rehearsal" in transport) else ""),
            "config": {
                "workload": (f"bg {args.bg_res}^3 @ {args.bg_voxel * 100:g} cm + {nobj_total} obj "
                             f"{args.obj_res}^3, {W}x{H}, full EM association + weighted fusion"
                             + (" (BASELINE.json configs[1])"
                                if (world, nobj_total, args.bg_res, args.obj_res, W, H) ==
                                (1, 4, 512, 128, 640, 480) else "")),
                "objects_total": nobj_total,
                "objects_per_gpu": args.objects_per_gpu,
                "background": "replicated" if world > 1 else "single",
                "gradients": args.grads,
                "estep_per_frame": 3,
                "transport": "none (one rank)" if comm is None else transport,
                "collectives_per_frame": ("none" if comm is None else
                                          ("broadcast(depth) + " if depth_broadcast else "") +
                                          "3 x all-reduce(sum f32, normaliser) + ONE grouped exchange per raycast "
                                          "(all-reduce(min u64, nearest-hit keys) + broadcasts of the background "
                                          "raycast's row bands): 5 exchanges; with a 30 us latency model about two of "
                                          "them are exposed (tests/test_gpu_exchange_latency.py)"),
                "tracking": "camera + objects, weighted LM-ICP, <= 100 iterations" if args.track
                            else "none (poses supplied, SURVEY 8d)",
                "mask_frames_every": mask_every,
                "mask_frames_in_timed_window": [f for f in range(args.warmup, nframes) if f % mask_every == 0],
                "visible_objects_last_frame": len(visible),
                "path": "batched" if chunks else "per-volume",
                "launches_per_stage": chunks,
                "device": f"{dev_name or 'MI355X'} {arch} {cus} CUs",
            },
        }
        if args.track and track_steps:
            n = len(track_steps)
            result["tracking_steps"] = {
                "note": "LM steps of the timed frames' stages, counted in the untimed replay: one launch per step, +1 for the sums "
                        "at the stage's first pose, +1 for the last verdict, + the launches the host had queued when the stage ended",
                "camera_per_frame": round(sum(t[0] for t in track_steps) / n, 1),
                "camera_accepted_per_frame": round(sum(t[1] for t in track_steps) / n, 1),
                "objects_longest_per_frame": round(sum(t[2] for t in track_steps) / n, 1),
                "frames_in_which_an_object_used_the_whole_budget": sum(1 for t in track_steps if t[2] >= prm.max_tracking_iter),
            }
        # priced by the newest committed PMC summary of THIS workload (profiles/*_counters.json carry a workload key);
        # with N > 1 every rank is priced with the profile of its SHARE (its objects + the background on one GPU)
        result["config"]["workload_key"] = share_key
        result["roofline"] = my_roof
        if my_rows is not None:
            result["kernels"] = my_rows
        if world > 1 or comm is not None:
            result["per_rank_roofline"] = per_rank
            if my_roof is not None:
                my_roof["share_note"] = ("counters of the committed profile of this rank's share run UNSHARDED on one GPU "
                                         f"({share_key}); the sharded launch marches 1/{world} of the background's rows and adds "
                                         "the exchanges, durations are this run's HIP events on this rank")
        result["hbm_copy_GBs"] = copy_gbs  # attainable D2D stream bandwidth of THIS box (read + write)
        # work aggregate for the scaling curves: every rank's volumes advance one frame per step
        result["volume_frames_per_s"] = round(fps * (world + nobj_total), 1)
        result["scaling_note"] = ("weak scaling of ONE RGB-D stream: every GPU adds --objects-per-gpu "
                                  "object volumes to the scene and re-integrates / raycasts its replica of "
                                  "the background, so frames/s of the joint scene is flat by design (ideal "
                                  "= the N=1 value); volume_frames_per_s is the aggregate work rate, which "
                                  "grows with N")
    fus.close()
    import bench_extras
    if args.strong_objects is None:
        args.strong_objects = 64 if (W, H, args.bg_res, args.obj_res) == (640, 480, 512, 128) else 0
    if comm is not None:
        # what the transport itself saw, and the line's own proof that the sharded frames are the single-GPU frames
        rep = bench_extras.transport_report(comm, dist, world, args.comm)
        par = None if args.no_parity else bench_extras.sharded_parity(pipeline, ops, DeviceArray, prm, comm, dist, rank, world,
                                                                        nobj_total, synth_factory, depth_broadcast,
                                                                        frames=args.parity_frames)
        if rank == 0:
            result["rccl"] = rep
            result["sharded_parity"] = par
            if par is not None and not par.get("ok"):
                print("bench.py: sharded_parity FAILED -- the sharded frames differ from the single-rank frames: " + json.dumps(par),
                      file=sys.stderr, flush=True)
    if not args.no_strong and not args.track and args.strong_objects > 0 and comm is not None and \
            args.strong_objects > 32 * world:
        if rank == 0:  # (one rank with --force-sharded: 64 local objects)
            result["strong_scaling"] = {"skipped": f"the sharded path takes <= 32 objects per rank ({args.strong_objects} over {world})"}
    elif not args.no_strong and not args.track and args.strong_objects > 0:
        strong = bench_extras.strong_scaling(args, pipeline, ops, DeviceArray, prm, comm, dist, rank, world, synth_factory,
                                             depth_broadcast, args.strong_objects)
        if rank == 0:
            result["strong_scaling"] = strong
    is_headline = (world, nobj_total, args.bg_res, args.obj_res, W, H, args.track, comm) == (1, 4, 512, 128, 640, 480, False, None)
    if rank == 0 and is_headline and not args.no_target:
        # north_star's own target (>= 30 frames/s with 8 object volumes), same protocol, same process, behind the headline
        result["target_config"] = measure_target(args, pipeline, ops, DeviceArray, fus_params=prm)
        result["steady_state"] = measure_steady_state(args, pipeline, ops, DeviceArray, prm, nobj_total)
    if rank == 0 and is_headline and not args.no_entry:
        # the reference's own entry point, processFrame(RGBD), on the staged TUM-layout scene (configs[2]'s throughput)
        result["entry_point"] = bench_extras.entry_point(pipeline, ops, DeviceArray)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, prm, K, synth, ids)

    synth.close()
    if comm is not None:
        comm.close()
    if rank == 0:
        # libraries (RCCL's version banner) write to C stdio: flush that first so that the JSON line
        # is the last line of the output whatever the buffering
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measure_window(args, pipeline, ops, DeviceArray, prm, nobj, first, count):
    """Frames [first, first + count) of the synthetic stream with `nobj` objects, timed between two device
    synchronisations after frames [0, first) have run untimed, in a fresh emf::EMFusion of this process: the headline's
    protocol (inputs resident in HBM, the interpreter's collector kept out) on another scene or another window."""
    import gc
    W, H = args.width, args.height
    K = np.array(prm.K, np.float32)
    synth = pipeline.SyntheticStream(W, H, K, nobj, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(nobj)]
    frames, keep = [], []
    for f in range(first + count):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
        rm = f % prm.mask_frames == 0
        masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if rm else {}
        d = DeviceArray.from_numpy(depth)
        keep.append((d, masks))
        frames.append((ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, rm))
    for f in range(first):
        fus.process_frame(*frames[f])
    fus.synchronize()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for f in range(first, first + count):
        fus.process_frame(*frames[f])
    fus.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    visible = len(fus.visible_objects())
    fus.close()
    synth.close()
    return elapsed, visible, [f for f in range(first, first + count) if f % prm.mask_frames == 0]


def measure_target(args, pipeline, ops, DeviceArray, fus_params, nobj=8):
    """north_star: ">= 30 frames/sec integrate+raycast+EM-update at 640x480 with 1 background (512^3) + 8 object (128^3)
    volumes on 1 MI355X".  The headline's protocol (W warm-up frames, K timed ones) on that scene."""
    W, H = args.width, args.height
    elapsed, visible, _ = measure_window(args, pipeline, ops, DeviceArray, fus_params, nobj, args.warmup, args.steps)
    return {"workload": f"bg {args.bg_res}^3 @ {args.bg_voxel * 100:g} cm + {nobj} obj {args.obj_res}^3, {W}x{H}, full EM "
                        "association + weighted fusion (north_star's single-GPU target; also one GPU's share of "
                        "BASELINE.json configs[3])",
            "workload_key": workload_key(W, H, args.bg_res, args.obj_res, nobj),
            "value": round(args.steps / elapsed, 3), "unit": "frames/s", "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "steps": args.steps, "warmup": args.warmup, "target": 30.0, "visible_objects_last_frame": visible,
            "note": "no HIP-event pairs in this second run (the headline's run has them around every 4th long launch)"}


STEADY_FIRST, STEADY_COUNT = 25, 100


def measure_steady_state(args, pipeline, ops, DeviceArray, fus_params, nobj):
    """The headline's workload over frames 25..124 of the same stream: volumes populated, weights saturating, and
    four mask frames (30, 60, 90, 120: `integrateMasks`, EMFusion.cpp:891-906, every 30th frame) inside the window --
    the driver's `--steps 20 --warmup 5` window (frames 5..24) holds none and is still filling the volumes."""
    elapsed, visible, mask_frames = measure_window(args, pipeline, ops, DeviceArray, fus_params, nobj, STEADY_FIRST, STEADY_COUNT)
    return {"frames": [STEADY_FIRST, STEADY_FIRST + STEADY_COUNT - 1], "value": round(STEADY_COUNT / elapsed, 3), "unit": "frames/s",
            "ms_per_step": round(1e3 * elapsed / STEADY_COUNT, 4), "mask_frames_in_window": mask_frames,
            "visible_objects_last_frame": visible,
            "note": "same workload and protocol as the headline line, later and longer window; no HIP-event pairs"}


# kernel kind -> (HIP kernel symbol for the rocprof cross-check, per-unit algorithmic bytes note)
KERNEL_NAMES = {
    "integrate": "k_integrate_cull + k_integrate_listed",
    "integrate_bg": "k_integrate_cull + k_integrate_listed<out of place> (background, beside k_raycast)",
    "raycast": "k_raycast",
    "assoc": "k_assoc",
    "normalize": "k_assoc_normalize",
    "composite": "k_composite (+k_vis_counts)",
    "points": "k_compute_points",
    "grads": "k_tsdf_grads",
    "fgbg": "k_update_fgbg (+k_fg_probs)",
    "track": "k_track_* (one stage: prepare + LM iterations until convergence)",
}


def algorithmic_bytes(kind, summ, stats, P, steps=None):
    """Algorithmic bytes summed over the BRACKETED launches of one kernel kind -- SURVEY.md section 8(d),
    restated in DESIGN.md "Byte model".  `units` = voxels (sweeps) or pixels (image kernels) of those launches;
    the march samples are counted over all `steps` timed frames (untimed replay) and scaled to them."""
    u, n = summ["units"], max(summ["launches"], 1)
    share = n / float(steps) if steps else 1.0  # bracketed launches / timed launches (--event-stride)
    if kind in ("integrate", "integrate_bg"):   # B_int: read tsdf + weight, write tsdf + weight per voxel
        return 16.0 * u
    if kind == "grads":       # B_grad: 4 B read + 12 B written per voxel
        return 16.0 * u
    if kind == "raycast":     # B_ray: 16 gathers per march sample, gradient blend at hits, outputs
        if stats is None:     # --no-stats-replay: the samples were not counted
            return 0.0
        S, hits = stats[0] * share, stats[1] * share
        return 64.0 * S + 96.0 * hits + 29.0 * u
    if kind == "assoc":       # B_em per model and pixel: 12 B point + 32 B tsdf gather + 4 B out
        return 48.0 * u       # (objects add a 32 B fg gather; counted at the background's rate)
    if kind == "normalize":   # every map read and written once
        return 8.0 * u
    if kind == "composite":   # 29 B per model and pixel in, 30 B per pixel out
        return 29.0 * u
    if kind == "points":      # 4 B in, 12 B out
        return 16.0 * u
    if kind == "fgbg":        # tsdf, weight, counts RMW + probability / mask out
        return 37.0 * u
    return 0.0


def workload_key(W, H, bg_res, obj_res, nobj, track=False):
    """What a committed profile must have been taken on to price this run's kernels (profiles/*_counters.json)."""
    return f"{W}x{H}_bg{bg_res}_obj{nobj}x{obj_res}" + ("_track" if track else "")


HEADLINE_KEY = workload_key(640, 480, 512, 128, 4)


def committed_profile(key=HEADLINE_KEY):
    """The newest committed PMC summary of the workload `key` under profiles/ (scripts/profile_round.sh +
    summarize_profile.py): `*_traffic.json` (HBM-side bytes per launch from the sized TCC_EA0 request
    counters) and `*_counters.json` (issue / L1 / L2 counters per launch), both over the timed launches of
    the driver's protocol.  The counters cannot be read from inside this process: they are those of the
    profiled run of the same command, and the line says which one (`counters_from`)."""
    out = {"tag": None, "traffic": {}, "counters": {}, "protocol": None, "file": None}
    for f in sorted((ROOT / "profiles").glob("*_counters.json"), reverse=True):
        try:
            c = json.loads(f.read_text())
            if c.get("workload_key", HEADLINE_KEY) != key:  # (profiles older than round 4 carry no key: configs[1])
                continue
            out.update(tag=c.get("tag"), counters=c.get("kernels", {}), protocol=c.get("protocol"), file=f.name)
            t = f.with_name(f.name.replace("_counters.json", "_traffic.json"))
            if t.exists():
                out["traffic"] = json.loads(t.read_text()).get("kernels", {})
            return out
        except (OSError, ValueError):
            continue
    return out


# Peaks the resource fractions are priced against (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 at 2.4 GHz max
# clock; a wave64 VALU instruction occupies its SIMD for 2 cycles; L2 ~34.5 TB/s; HBM3E 8 TB/s spec).  The
# vector L1 looks up one 64-byte tag per cycle and CU: bench's own copy kernel moves 1 KiB per wave
# instruction in 16 TCP_TOTAL_CACHE_ACCESSES, and a 64-lane 8-byte gather costs 16 accesses (one per 4
# lanes) however few lines it touches -- the march's bound while the chip is full.
CLOCK_HZ = 2.4e9
PEAKS = {
    "valu": ("G wave-instr/s", 256 * 4 * CLOCK_HZ / 2.0 / 1e9),
    "l1": ("G tag lookups/s", 256 * CLOCK_HZ / 1e9),
    "l2": ("GB/s", 34500.0),
    "hbm": ("GB/s", HBM_PEAK_GBS),
}


def resource_fractions(kind, avg_ms, prof):
    """Achieved rate and fraction of peak of every resource the committed counters price, for one kernel
    kind at the launch duration measured LIVE in this run (HIP events): {resource: {achieved, peak, unit,
    frac}}.  frac <= 1 by construction of the peaks (a counter-backed utilisation, not a byte model)."""
    c = prof["counters"].get(kind)
    if not c or avg_ms <= 0:
        return None
    t = avg_ms * 1e-3

    def per(name):
        return c.get(name, {}).get("per_launch")

    # request sizes at the L2, calibrated in the same passes by the 1 GiB copy kernel (known bytes)
    cal = prof["counters"].get("stream_copy", {})
    rd_req = cal.get("TCC_READ_sum", {}).get("per_launch")
    wr_req = cal.get("TCC_WRITE_sum", {}).get("per_launch")
    rd_bytes = (2.0 ** 30) / rd_req if rd_req else 128.0
    wr_bytes = (2.0 ** 30) / wr_req if wr_req else 64.0
    rates = {}
    if per("SQ_INSTS_VALU") is not None:
        rates["valu"] = per("SQ_INSTS_VALU") / t / 1e9
    if per("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
        rates["l1"] = per("TCP_TOTAL_CACHE_ACCESSES_sum") / t / 1e9
    if per("TCC_READ_sum") is not None and per("TCC_WRITE_sum") is not None:
        rates["l2"] = (per("TCC_READ_sum") * rd_bytes + per("TCC_WRITE_sum") * wr_bytes) / t / 1e9
    tr = prof["traffic"].get(kind)
    if tr:
        rates["hbm"] = tr["hbm_bytes_per_launch"] / t / 1e9
    out = {}
    for r, a in rates.items():
        unit, peak = PEAKS[r]
        out[r] = {"achieved": round(a, 2), "peak": round(peak, 1), "unit": unit, "frac": round(a / peak, 4)}
    waves = per("SQ_WAVE_CYCLES")
    if waves is not None:  # quad-cycles summed over the waves -> mean resident waves per SIMD
        out["mean_waves_per_simd"] = round(waves * 4.0 / (t * CLOCK_HZ * 1024), 2)
    return out


def copy_bandwidth(devmem, ops, mib=1024, reps=10):
    """GB/s (bytes read + bytes written) of the plain 16-byte-per-lane copy kernel between two
    `mib` MiB buffers: what this box's HBM sustains for a pure stream."""
    from emfusion_amd.devmem import DeviceArray, Event
    n = mib * 1024 * 1024 // 4
    src, dst = DeviceArray.zeros((n,), np.float32), DeviceArray.zeros((n,), np.float32)
    for _ in range(3):
        ops.stream_copy(dst, src)
    e0, e1 = Event(), Event()
    e0.record()
    for _ in range(reps):
        ops.stream_copy(dst, src)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_ms(e1) / reps
    del src, dst
    return round(2.0 * n * 4 / (ms * 1e-3) / 1e9, 1)


L1_PROBE = dict(footprint=16384, iterations=2048, workgroups=2048)  # 8192 waves x 2048 gathers = 16.8 M wave-instr


def l1_probe(devmem, ops, reps=5):
    """Gather instructions per second of emf_hip_l1GatherProbe on an L1-resident footprint at full occupancy, for 64 / 4 /
    1 distinct lines per 64-lane 8-byte load: {lines: {"ms": launch time, "G_instr_s": rate}} (HIP events, this run)."""
    from emfusion_amd.devmem import DeviceArray, Event
    buf = DeviceArray.zeros((L1_PROBE["footprint"] // 4,), np.float32)
    sink = DeviceArray.zeros((2,), np.float32)
    out = {}
    for lines in (64, 4, 1):
        def run():
            ops.l1_gather_probe(buf, L1_PROBE["footprint"], lines, L1_PROBE["iterations"], L1_PROBE["workgroups"], sink)
        run()
        e0, e1 = Event(), Event()
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_ms(e1) / reps
        n = L1_PROBE["workgroups"] * 4 * L1_PROBE["iterations"]
        out[lines] = {"ms": round(ms, 4), "G_instr_s": round(n / (ms * 1e-3) / 1e9, 2)}
    return out


def l1_measured_peak(probe, prof):
    """Tag look-ups per second the probe reaches: TCP_TOTAL_CACHE_ACCESSES per launch of the same probe kernel in the
    committed PMC profile over its launch time in THIS run; the largest of the three access patterns.  None without
    committed counters of the probe (profiles older than round 4)."""
    best = None
    for lines, r in probe.items():
        c = prof["counters"].get(f"l1_probe_{lines}", {}).get("TCP_TOTAL_CACHE_ACCESSES_sum", {}).get("per_launch")
        if c and r["ms"] > 0:
            rate = c / (r["ms"] * 1e-3) / 1e9
            best = rate if best is None or rate > best else best
    return round(best, 1) if best else None


def l1_probe_report(probe, prof):
    rep = {"kernel": "k_l1_probe<lines>: 8-byte gathers, 16 KiB footprint, 8192 resident waves", "assumed_peak": 614.4,
           "unit": "G tag lookups/s", "patterns": {}}
    for lines, r in probe.items():
        c = prof["counters"].get(f"l1_probe_{lines}", {}).get("TCP_TOTAL_CACHE_ACCESSES_sum", {}).get("per_launch")
        n = L1_PROBE["workgroups"] * 4 * L1_PROBE["iterations"]
        rep["patterns"][f"{lines}_lines_per_instr"] = {
            "launch_ms": r["ms"], "G_gather_instr_s": r["G_instr_s"],
            "lookups_per_instr": round(c / n, 2) if c else None,
            "G_tag_lookups_s": round(c / (r["ms"] * 1e-3) / 1e9, 1) if c else None}
    rep["value"] = l1_measured_peak(probe, prof)
    rep["counters_from"] = prof.get("file")
    return rep


def roofline(kern, stats, P, copy_gbs=None, profiled=True, steps=None, l1_probe=None):
    """`roofline` of the JSON line: the dominant kernel against the resource that binds it.

    bound / achieved / peak / frac come from hardware counters (resource_fractions): VALU issue, vector-L1
    tag lookups (the gather path), L2 bytes, HBM bytes -- the largest fraction is the bound.  SURVEY 8(d)'s
    algorithmic byte model stays in the line as `model_GBs` (an upper bound on gather bytes that the caches
    beat, NOT a fraction of anything)."""
    key = HEADLINE_KEY if profiled is True else (None if profiled is False else profiled)
    none = {"tag": None, "traffic": {}, "counters": {}, "protocol": None, "file": None}
    prof = committed_profile(key) if key else none
    if l1_probe:  # quote the L1 against what the probe kernel reaches on this box, not against one look-up per clock
        peak = l1_measured_peak(l1_probe, prof if prof["counters"] else committed_profile())
        if peak:
            PEAKS["l1"] = ("G tag lookups/s", peak)
    rows = []
    for kind, summ in kern.items():
        if kind.startswith("_") or summ["launches"] == 0:
            continue
        total_b = algorithmic_bytes(kind, summ, stats, P, steps if kind == "raycast" else None)
        n = summ["launches"]
        avg_ms = summ["total_ms"] / n
        row = {
            "kernel": KERNEL_NAMES[kind], "kind": kind, "launches": n,
            "avg_ms": round(avg_ms, 5), "total_ms": round(summ["total_ms"], 3),
            "alg_bytes_per_launch": round(total_b / n, 1),
            "model_GBs": round(total_b / n / (avg_ms * 1e-3) / 1e9, 2) if avg_ms > 0 else None,
        }
        res = resource_fractions(kind, avg_ms, prof)
        if res:
            row["resources"] = res
        rows.append(row)
    rows.sort(key=lambda r: -r["total_ms"])
    # The dominant kernel is the longest one on the frame's critical stream.  The background's integration
    # runs beside the raycast on a second, lowest-priority stream and the main stream only joins it after the
    # composite and the objects' integration (~0.06 ms later): its launch is as long as the gaps the raycast
    # leaves it (0.23 ms alone), so it only counts when it really is what the frame waits for.
    ray = next((r for r in rows if r["kind"] == "raycast"), None)
    dom = next(r for r in rows
               if not (r["kind"] == "integrate_bg" and ray is not None and r["avg_ms"] < 1.1 * ray["avg_ms"]))
    res = {k: v for k, v in (dom.get("resources") or {}).items() if isinstance(v, dict)}
    roof = {"kernel": dom["kernel"], "avg_launch_ms": dom["avg_ms"],
            "timed_launches": f"{dom['launches']} of {steps} bracketed by HIP events (--event-stride)" if steps else dom["launches"],
            "dropped_launches": kern.get("_dropped", 0)}
    if res:
        bound = max(res, key=lambda k: res[k]["frac"])
        roof.update(bound=bound, achieved=res[bound]["achieved"], peak=res[bound]["peak"], unit=res[bound]["unit"],
                    frac=res[bound]["frac"], resources=dom["resources"])
        tr = prof["traffic"].get(dom["kind"])
        roof["traffic"] = round(tr["hbm_bytes_per_launch"], 1) if tr else None
        roof["counters_from"] = (f"profiles/{prof['tag']}_counters.json + _traffic.json: rocprofv3 --pmc passes of "
                                 f"{prof['protocol']}; durations are this run's HIP events")
    else:  # no committed counters for this workload: nothing to price the kernel against
        roof.update(bound=None, achieved=None, peak=None, unit=None, frac=None, traffic=None,
                    counters_from=f"none committed for this workload ({key}; profiles/ holds "
                                  f"{sorted(set(json.loads(f.read_text()).get('workload_key', HEADLINE_KEY) for f in (ROOT / 'profiles').glob('*_counters.json')))})")
    roof["model_GBs"] = dom["model_GBs"]
    roof["alg_bytes_per_launch"] = dom["alg_bytes_per_launch"]
    roof["model_note"] = ("model_GBs = SURVEY 8(d)'s algorithmic bytes per launch over the launch duration; for the "
                          "raycast that is 64 B per march sample + 96 B per hit + 29 B per pixel and model, an upper "
                          "bound on gather bytes that L1 / L2 serve (hit rate 0.86), so it can exceed the HBM peak -- "
                          "it is a rate of the byte model, not a roofline fraction")
    if copy_gbs:
        roof["hbm_copy_GBs"] = copy_gbs
    if l1_probe:
        roof["l1_tag_peak_measured"] = l1_probe_report(l1_probe, prof if prof["counters"] else committed_profile())
    # the kernel that runs BESIDE the dominant one shares its resources: the chip's utilisation while the
    # raycast runs is the sum of both
    integ = next((r for r in rows if r["kind"] == "integrate_bg"), None)
    overlapped = integ is not None
    if integ is None:
        integ = next((r for r in rows if r["kind"] == "integrate"), None)
    if integ:
        ires = {k: v for k, v in (integ.get("resources") or {}).items() if isinstance(v, dict)}
        entry = {"kernel": integ["kernel"], "concurrent_with_raycast": overlapped, "avg_launch_ms": integ["avg_ms"],
                 "model_GBs": integ["model_GBs"], "alg_bytes_per_launch": integ["alg_bytes_per_launch"]}
        if ires:
            b = max(ires, key=lambda k: ires[k]["frac"])
            tr = prof["traffic"].get(integ["kind"])
            entry.update(bound=b, frac=ires[b]["frac"], resources=integ["resources"],
                         traffic=round(tr["hbm_bytes_per_launch"], 1) if tr else None)
        entry["note"] = ("model_GBs = 16 B x every voxel of the integrated volumes (SURVEY 8d) over the launch time: "
                         "full-sweep equivalent -- boxes outside the view cone are culled before they are touched and "
                         "unseen tiles are integrated without being read, so `traffic` is what really moves")
        roof["integrate_stream"] = entry
        if overlapped and res and ires and dom["kind"] == "raycast":
            roof["chip_while_raycast_runs"] = {
                k: round(res[k]["frac"] + ires[k]["frac"] * min(1.0, integ["avg_ms"] / dom["avg_ms"]), 4)
                for k in res if k in ires}
    if dom["kind"] == "raycast" and stats is not None:
        roof["march_samples_per_launch"] = round(stats[0] / max(steps or dom["launches"], 1), 1)
        # samples gathered incl. the speculative ones of the lanes-per-ray march that were dropped (= samples with one lane per ray)
        roof["march_gathered_per_launch"] = round(stats[2] / max(steps or dom["launches"], 1), 1)
    return roof, rows


def cpu_baseline(args, prm, K, synth, ids):
    """The oracle (our CPU restatement, kind "port") on the host cores of this box, as SURVEY 8(d) defines
    the CPU baseline: built -O3 -march=native ON this host (same unfused arithmetic, same bits), OpenMP over
    the threads this process may really use (affinity mask capped by the cgroup's CPU quota), on a bounded
    sample of the SAME workload -- frame 0 untimed (first integration), frame 1 on ONE thread, then whole
    frames of the schedule on all threads until about --cpu-seconds are spent.  `value` is the all-thread
    figure; the 1-thread figure stands beside it."""
    from oracle import binding as oracle
    from tests.oracle_pipeline import Affine32, OraclePipeline

    native = oracle.use_native(True)
    cores = oracle.host_threads()
    W, H = args.width, args.height
    orc = OraclePipeline(oracle, W, H, K, args.bg_res, args.bg_voxel, list(prm.volume_pose_t),
                         args.obj_res, visibility_thresh=prm.visibility_thresh,
                         boundary=prm.boundary)
    for k, i in enumerate(ids):
        c, r, vs = synth.sphere(k, 0)
        orc.add_object(c, vs)

    def run(f):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        poses = {i: Affine32(t=synth.sphere(i - 1, f)[0]) for i in ids}
        masks = {i: (sid == i).astype(np.uint8) for i in ids} if f % prm.mask_frames == 0 else {}
        t0 = time.perf_counter()
        orc.process_frame(depth, Affine32(R.reshape(3, 3), t), poses, masks, bool(masks))
        return time.perf_counter() - t0

    oracle.set_threads(cores)
    run(0)
    oracle.set_threads(1)
    one = run(1)  # one whole frame of the schedule on one thread
    oracle.set_threads(cores)
    spent, n = 0.0, 0
    while spent < args.cpu_seconds and n < 20:
        spent += run(2 + n)
        n += 1
    oracle.use_native(False)
    cpu_model = ""
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {
        "value": round(n / spent, 4),
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "one_thread": {"value": round(1.0 / one, 4), "unit": "frames/s", "cores": 1,
                       "sample": f"frame 1 of the same stream, {one:.1f} s"},
        "speedup_over_one_thread": round((n / spent) * one, 1),
        "build": ("gcc -O3 -march=native -ffp-contract=off -fopenmp, built on this host" if native else
                  "gcc -O2 -ffp-contract=off -fopenmp (portable build: the native build failed on this host)"),
        "host": f"{cpu_model}; {os.cpu_count()} logical CPUs, {cores} usable (affinity mask and cgroup quota)",
        "sample": (f"frames 2..{1 + n} of the same synthetic stream (frame 0 untimed, frame 1 on one thread), "
                   f"full schedule, bg {args.bg_res}^3 + {len(ids)} obj {args.obj_res}^3, {W}x{H}; OpenMP over "
                   f"{cores} host threads, {spent:.1f} s"),
    }


if __name__ == "__main__":
    main()
