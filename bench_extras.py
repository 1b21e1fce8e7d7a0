"""The measurements bench.py makes beside its headline (kept out of bench.py for length; same process, same protocol):

  * `tum_line` / `entry_point`   the reference's entry point, EMFusion::processFrame(const RGBD&) (reference
        src/core/EMFusion.cpp:70-129: upload -> bilateral pre-filter -> E-step -> camera tracking -> E-step -> object
        tracking -> E-step -> raycast -> masks / life cycle -> integrate -> clean-up), on the staged TUM-layout scene of
        tests/tum_scene.py with config/tum.cfg's values -- BASELINE.json configs[2]'s throughput -- and the same stream
        with its inputs resident in HBM (`bench.py --track`);
  * `strong_scaling`             configs[3]'s scene with a FIXED number of objects split over the ranks (frames/s rises
        with N; the N = 1 point is the chunked batched path);
  * `sharded_parity`             the multi-GPU line's own proof: replicas identical on every rank, joint images equal to a
        single-rank re-run of the same frames;
  * `transport_report`           what the transport itself says about ranks and devices.

Nothing here is timed inside the headline's timed region."""
from __future__ import annotations

import gc
import os
import pickle
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
TUM_CFG = ROOT / "tests" / "golden" / "tum_fullsize.cfg"  # the two values config/tum.cfg changes; everything else = Params' defaults


# --------------------------------------------------------------------------------------------------------------------
# the staged TUM-layout stream (tests/tum_scene.py), in memory: depth as the 16-bit PNG would hold it, Mask%04d.plk on disk
# --------------------------------------------------------------------------------------------------------------------
class TumStream:
    def __init__(self, frames, seed=0x7A5C):
        from tests import tum_scene as S
        self.S = S
        self.tmp = tempfile.TemporaryDirectory(prefix="emf_bench_tum_")
        self.masks_dir = Path(self.tmp.name) / "masks"
        self.masks_dir.mkdir()
        rng = np.random.default_rng(seed)
        self.depth, self.masks, self.scores, self.truth = [], {}, {}, []
        t0 = time.time()
        for f in range(frames):
            depth, ids = S.render(f, rng)
            q16 = np.round(depth * 5000.0).astype(np.uint16)                # TUM depth PNG: 1 / 5000 m
            self.depth.append(q16.astype(np.float32) * np.float32(1.0 / 5000.0))  # TUMRGBDReader's floats
            self.truth.append(S.camera_pose(f))
            if f % S.MASK_EVERY == 0:
                m = ids == 1
                ys, xs = np.nonzero(m)
                box = [int(ys.min()), int(xs.min()), int(ys.max()) + 1, int(xs.max()) + 1] if m.any() else [0, 0, 1, 1]
                scores = np.full(81, 0.001)
                scores[S.PERSON_CLASS] = 0.92
                with open(self.masks_dir / f"Mask{f:04d}.plk", "wb") as fh:
                    pickle.dump(([box], [m], [scores.tolist()]), fh, protocol=2)
                self.masks[f], self.scores[f] = m.astype(np.uint8), scores.tolist()
        self.render_seconds = time.time() - t0
        self.frames = frames

    def close(self):
        self.tmp.cleanup()


def _tum_run(pipeline, ops, DeviceArray, stream, mode, warmup, steps, env=None, kernel_events=None, raycast_stats=False,
             collect_track=True):
    """One pass over frames [0, warmup + steps) of `stream` in a fresh emf::EMFusion built from tests/golden/tum_fullsize.cfg;
    frames [warmup, warmup + steps) are timed between two device synchronisations.
      mode "entry"   : process_rgbd(host depth) -- EMFusion::processFrame(const RGBD&): upload, bilateral pre-filter, masks
                       read from <masks>/Mask%04d.plk on every 30th frame, object life cycle, camera + object tracking, clean-up
      mode "resident": the same frames with depth maps and instance masks already in HBM (process_frame on device views,
                       pre-filter on, the instances queued on mask frames): the tracked hot path without the host's I/O
    kernel_events: None, or (kinds, stride) for the per-launch HIP-event timers.
    collect_track: read the stages' results back after every frame (a dozen calls through the handle API per frame, on the
    critical path of a host that is in lock-step with the device: the run whose time is REPORTED leaves it to an untimed
    pass over the same -- deterministic -- frames)."""
    saved = {}
    for k, v in (env or {}).items():
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        fus = pipeline.Fusion.from_config(TUM_CFG)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    prm = fus.params
    fus.set_cleanup(True)
    fus.set_tracking(camera=True, objects=True)  # (frame 0 has nothing to track against: EMFusion.cpp:76)
    n = warmup + steps
    keep = []
    if mode == "entry":
        fus.use_preproc_masks(str(stream.masks_dir))
    else:
        fus.set_preprocess(True)
        views = []
        for f in range(n):
            d = DeviceArray.from_numpy(stream.depth[f])
            keep.append(d)
            views.append(ops.image_view(d))
        dmasks = {f: DeviceArray.from_numpy(m) for f, m in stream.masks.items() if f < n}  # (carved in place: per run)
    eye, zero = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, np.float32)

    def step(f):
        if mode == "entry":
            fus.process_rgbd(stream.depth[f])
        else:
            if f in dmasks:
                fus.queue_instance_masks([ops.image_view(dmasks[f])])
                fus.queue_instance_scores([stream.scores[f]])
            fus.process_frame(views[f], eye, zero, {}, {}, False)

    for f in range(warmup):
        step(f)
    fus.synchronize()
    if kernel_events:
        fus.kernel_timers_enable(64 * steps + 64)
        fus.kernel_timers_select(kernel_events[0])
        fus.kernel_timers_stride(max(1, kernel_events[1]))
    if raycast_stats:
        fus.enable_raycast_stats(True)
    up0 = fus.upload_host_time()
    track = []
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for f in range(warmup, n):
        step(f)
        if collect_track:  # results of the stages that just ran (host values: the tracking driver has read them back already)
            ids = fus.object_ids()
            res = [fus.track_result(0)] + [fus.track_result(i) for i in ids]
            track.append((res[0]["iterations"], res[0]["accepted"], max([r["iterations"] for r in res[1:]] or [0]), len(ids)))
    issued = time.perf_counter() - t0
    fus.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    up1 = fus.upload_host_time()
    out = dict(elapsed=elapsed, issued=issued, track=track, max_iter=int(prm.max_tracking_iter),
               upload_host_ms=1e3 * (up1[0] - up0[0]) / max(up1[1] - up0[1], 1) if mode == "entry" else None,
               kern=fus.kernel_timers_collect() if kernel_events else None,
               stats=fus.raycast_stats() if raycast_stats else None,
               objects=fus.object_ids(), visible=sorted(fus.visible_objects()), chunks=fus.batched_chunks(),
               overlap=fus.background_overlap(), bg_res=int(prm.bg_res[0]), obj_res=int(prm.obj_res[0]),
               size=(int(prm.width), int(prm.height)), mask_frames=int(prm.mask_frames),
               pose=fus.pose(0))
    fus.close()
    del keep
    return out


def _tracking_steps(track, max_iter):
    n = max(len(track), 1)
    return {
        "note": "LM steps of the timed frames' stages as the tracking driver read them back (one launch per step, +1 for the sums "
                "at the stage's first pose, +1 for the last verdict, + the launches the host had queued when the stage ended)",
        "camera_per_frame": round(sum(t[0] for t in track) / n, 1),
        "camera_accepted_per_frame": round(sum(t[1] for t in track) / n, 1),
        "objects_longest_per_frame": round(sum(t[2] for t in track) / n, 1),
        "frames_in_which_a_stage_used_the_whole_budget": sum(1 for t in track if t[0] >= max_iter or t[2] >= max_iter),
        "live_objects": sorted(set(t[3] for t in track)),
    }


def _trajectory_error(stream, run, last_frame):
    """final camera translation against the scene's ground truth (the closed-loop accuracy tests are
    tests/test_gpu_tum_fullsize.py; this is a sanity figure that the timed run tracked the scene)"""
    t = np.asarray(run["pose"][1], np.float64)
    return round(float(np.linalg.norm(t - stream.truth[last_frame][1])) * 1e3, 3)


TUM_WORKLOAD = ("staged TUM-layout sequence (tests/tum_scene.py: 640x480, three non-parallel planes + furniture + a walking "
                "`person`, 0.2 % depth noise, 1 % drop-outs, PNG quantisation; masks every 30 frames) with config/tum.cfg's "
                "values: background 512^3 @ 1 cm + dynamic objects 64^3 -- BASELINE.json configs[2] on a stand-in for "
                "fr3/walking_xyz (the dataset cannot be staged here)")


def entry_point(pipeline, ops, DeviceArray, warmup=2, steps=40, stream=None):
    """Sub-line `entry_point` of the default run: processFrame(RGBD) timed with the double-buffered pinned upload and, for
    the before / after the verdict of round 5 asked for, with the synchronous pageable upload of rounds 1-5."""
    own = stream is None
    if own:
        stream = TumStream(warmup + steps)
    sync = _tum_run(pipeline, ops, DeviceArray, stream, "entry", warmup, steps, env={"EMF_ASYNC_UPLOAD": "0"}, collect_track=False)
    run = _tum_run(pipeline, ops, DeviceArray, stream, "entry", warmup, steps, collect_track=False)
    # (the same frames, the same stages: their results are read back in a third, unreported pass)
    run["track"] = _tum_run(pipeline, ops, DeviceArray, stream, "entry", warmup, steps)["track"]
    out = {
        "entry": "EMFusion::processFrame(const RGBD&) (reference src/core/EMFusion.cpp:70-129) through emf_fusion_process_rgbd: "
                 "host depth -> double-buffered pinned upload -> bilateral pre-filter -> E-step -> camera LM-ICP -> E-step -> "
                 "object LM-ICP -> E-step -> raycast -> masks (Mask%04d.plk every 30th frame) / object life cycle -> integrate "
                 "-> clean-up",
        "workload": TUM_WORKLOAD,
        "workload_key": "640x480_bg512_tumscene_entry",
        "value": round(steps / run["elapsed"], 2), "unit": "frames/s",
        "ms_per_step": round(1e3 * run["elapsed"] / steps, 4),
        "host_issue_ms_per_step": round(1e3 * run["issued"] / steps, 4),
        "steps": steps, "warmup": warmup,
        "frames": [warmup, warmup + steps - 1],
        "mask_frames_in_window": [f for f in range(warmup, warmup + steps) if f % run["mask_frames"] == 0],
        "tracking_steps": _tracking_steps(run["track"], run["max_iter"]),
        "upload": {
            "pinned_double_buffered": {"host_ms_per_frame": round(run["upload_host_ms"], 4),
                                       "frames_per_s": round(steps / run["elapsed"], 2)},
            "pageable_synchronous (EMF_ASYNC_UPLOAD=0, rounds 1-5)": {"host_ms_per_frame": round(sync["upload_host_ms"], 4),
                                                                      "frames_per_s": round(steps / sync["elapsed"], 2)},
            "note": "host_ms_per_frame = host time inside processFrame(RGBD) until the depth map is on its way: staging memcpy "
                    "+ enqueue on the copy stream vs. the runtime's own staging of a pageable source, which blocks the caller",
        },
        "final_camera_error_mm": _trajectory_error(stream, run, warmup + steps - 1),
        "objects_alive": run["objects"], "path": "batched" if run["chunks"] else "per-volume",
        "data": "synthetic (rendered in this process: %.1f s, untimed)" % stream.render_seconds,
    }
    if own:
        stream.close()
    return out


def tum_line(args, pipeline, ops, DeviceArray, roofline, workload_key_of, dev_desc, mode):
    """The whole JSON line of `bench.py --entry` / `bench.py --track` (N = 1)."""
    warmup, steps = args.warmup, args.steps
    stream = TumStream(warmup + steps)
    kinds = ["raycast", "integrate_bg", "track"]
    run = _tum_run(pipeline, ops, DeviceArray, stream, "entry" if mode == "entry" else "resident", warmup, steps,
                   kernel_events=None if args.no_kernel_events else (kinds, args.event_stride), collect_track=False)
    stats = None
    if not args.no_stats_replay:  # untimed replay of the same frames with the march counters on; reads the stages' results back
        rep = _tum_run(pipeline, ops, DeviceArray, stream, "entry" if mode == "entry" else "resident", warmup, steps,
                       raycast_stats=True)
        stats, run["track"] = rep["stats"], rep["track"]
    W, H = run["size"]
    key = f"{W}x{H}_bg{run['bg_res']}_tumscene_{'entry' if mode == 'entry' else 'track'}"
    fps = steps / run["elapsed"]
    result = {
        "metric": ("frames/sec (processFrame(RGBD): upload + pre-filter + EM-assoc + LM-ICP tracking + raycast + life cycle + integrate)"
                   if mode == "entry" else "frames/sec (integrate+raycast+EM-assoc+LM-ICP tracking)"),
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(1e3 * run["elapsed"] / steps, 4),
        "host_issue_ms_per_step": round(1e3 * run["issued"] / steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (staged TUM-layout scene, rendered in this process)",
        "config": {
            "workload": TUM_WORKLOAD,
            "inputs": ("host depth maps through processFrame(RGBD), masks from Mask%04d.plk" if mode == "entry" else
                       "depth maps and instance masks resident in HBM before the timed region; bilateral pre-filter, life cycle "
                       "and clean-up inside it"),
            "tracking": "camera + objects, weighted LM-ICP, <= 100 iterations",
            "objects_total": len(run["objects"]), "background": "single", "gradients": "onthefly", "estep_per_frame": 3,
            "mask_frames_every": run["mask_frames"],
            "mask_frames_in_timed_window": [f for f in range(warmup, warmup + steps) if f % run["mask_frames"] == 0],
            "path": "batched" if run["chunks"] else "per-volume", "launches_per_stage": run["chunks"],
            "visible_objects_last_frame": len(run["visible"]), "device": dev_desc, "workload_key": key,
        },
        "tracking_steps": _tracking_steps(run["track"], run["max_iter"]) if run["track"] else None,
        "final_camera_error_mm": _trajectory_error(stream, run, warmup + steps - 1),
    }
    if mode == "entry":
        result["upload_host_ms_per_frame"] = round(run["upload_host_ms"], 4)
    if run["kern"] is not None:
        result["roofline"], result["kernels"] = roofline(run["kern"], stats, W * H, None, key, steps, None)
    else:
        result["roofline"] = None
    stream.close()
    return result


# --------------------------------------------------------------------------------------------------------------------
# multi-GPU: what the transport saw, the line's own parity proof, and the strong-scaling sub-run
# --------------------------------------------------------------------------------------------------------------------
def transport_report(comm, dist, world, requested):
    """`rccl` of the N > 1 line: one describe() per rank, asked of the transport (ncclCommCount / ncclCommCuDevice / PCI bus
    id / ncclGetVersion) -- not of the launcher's environment."""
    mine = comm.describe()
    mine["pid"] = os.getpid()
    mine["hip_visible_devices"] = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    every = [mine]
    if dist is not None:
        every = [None] * world
        dist.all_gather_object(every, mine)
    buses = [e.get("pci_bus_id") for e in every]
    return {
        "requested": requested,
        "transport": every[0].get("transport"),
        "ranks": every[0].get("ranks"),
        "ranks_agree": len({e.get("ranks") for e in every}) == 1 and every[0].get("ranks") == world,
        "version": every[0].get("version"),
        "devices": [{k: e.get(k) for k in ("rank", "device", "pci_bus_id", "pid", "hip_visible_devices")} for e in every],
        "distinct_devices": len(set(buses)),
        "one_device_per_rank": len(set(buses)) == world,
        "system_fences": every[0].get("system_fences"),
    }


def _digest(a):
    import xxhash
    return xxhash.xxh3_128(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).hexdigest()


def _outside(a, b, rtol=1e-4, atol=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) > rtol * np.abs(b) + atol).mean())


def sharded_parity(pipeline, ops, DeviceArray, prm, comm, dist, rank, world, nobj_total, synth_factory, depth_broadcast,
                   frames=4):
    """Run `frames` frames of the line's own scene from a cleared state on the sharded path, then -- rank 0, alone -- the same
    frames on ONE rank without any exchange, and compare:
      * the digests of the replicated background (tsdf, weights) and of every joint image are equal on all ranks;
      * rank 0's joint segmentation / ray lengths / association normaliser equal the single-rank run's (bit-identical
        where the normaliser's re-ordered sum allows it: the sharded sum adds the ranks' partials, reference
        EMFusion.cpp:653-665 adds map after map) within north_star's 1e-4.
    Untimed; every rank must call it."""
    W, H = prm.width, prm.height
    K = np.array(prm.K, np.float32)
    synth = synth_factory(nobj_total)
    eye = np.eye(3, dtype=np.float32).reshape(-1)

    def run(c, bcast):
        fus = pipeline.Fusion(prm, c)
        if c is not None and bcast:
            fus.set_depth_broadcast(0)
        ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(nobj_total)]
        mine = [i for i in ids if fus.owns_object(i)]
        keep = []
        for f in range(frames):
            depth, sid = synth.render(f)
            R, t = synth.camera_pose(f)
            # with the broadcast on only rank 0 holds the frame: the others must get it through the exchange
            d = DeviceArray.from_numpy(depth if (c is None or not bcast or rank == 0) else np.zeros_like(depth))
            poses = {i: (eye, synth.sphere(i - 1, f)[0]) for i in mine}
            masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in mine} if f == 0 else {}
            keep += [d, masks]
            fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, f == 0)
        fus.synchronize()
        imgs = {k: fus.image(k) for k in ("segmentation", "raylengths", "bg_raylengths", "assoc_norm", "bg_assoc")}
        dg = {k: _digest(v) for k, v in imgs.items()}
        dg["bg_tsdf"] = _digest(fus.volume("tsdf", 0))
        dg["bg_weights"] = _digest(fus.volume("weights", 0))
        vis = sorted(fus.visible_objects())
        chunks = fus.batched_chunks()
        fus.close()
        return imgs, dg, vis, chunks

    imgs, dg, vis, chunks = run(comm, depth_broadcast)
    every = [(dg, vis)]
    if dist is not None:
        every = [None] * world
        dist.all_gather_object(every, (dg, vis))
    out = {"frames": frames, "objects_total": nobj_total, "launches_per_stage": chunks}
    if rank == 0:
        differing = sorted({k for e in every for k in dg if e[0][k] != every[0][0][k]})
        out["replicas_and_joint_images_identical_on_all_ranks"] = not differing
        out["differing"] = differing
        out["visible_sets_identical_on_all_ranks"] = all(e[1] == every[0][1] for e in every)
        simgs, _, svis, schunks = run(None, False)  # one rank, no exchange, the whole scene
        seg_bad = float((imgs["segmentation"] != simgs["segmentation"]).mean())
        same = imgs["segmentation"] == simgs["segmentation"]
        cmp = {
            "single_rank_launches_per_stage": schunks,
            "visible_sets_equal": vis == svis,
            "segmentation_mismatch_frac": round(seg_bad, 6),
            "raylengths_outside_1e-4": round(_outside(imgs["raylengths"][same], simgs["raylengths"][same]), 6),
            "bg_raylengths_outside_1e-4": round(_outside(imgs["bg_raylengths"], simgs["bg_raylengths"]), 6),
            "normaliser_outside_1e-4": round(_outside(imgs["assoc_norm"], simgs["assoc_norm"]), 6),
            "bg_association_outside_1e-4": round(_outside(imgs["bg_assoc"], simgs["bg_assoc"], atol=1e-7), 6),
            "bit_identical": {k: bool(np.array_equal(imgs[k], simgs[k])) for k in imgs},
            "labels_in_segmentation": int(len(np.unique(simgs["segmentation"])) - 1),
        }
        out["vs_single_rank"] = cmp
        out["ok"] = bool(out["replicas_and_joint_images_identical_on_all_ranks"] and out["visible_sets_identical_on_all_ranks"] and
                         cmp["visible_sets_equal"] and seg_bad < 2e-3 and cmp["raylengths_outside_1e-4"] < 5e-3 and
                         cmp["bg_raylengths_outside_1e-4"] < 5e-3 and cmp["normaliser_outside_1e-4"] < 1e-3 and
                         cmp["bg_association_outside_1e-4"] < 1e-3)
        out["bounds"] = ("segmentation < 2e-3, ray lengths < 5e-3, normaliser / background association < 1e-3 of the pixels "
                         "outside 1e-4 relative (tests/test_gpu_config3_rehearsal.py's bounds)")
    synth.close()
    if dist is not None:
        dist.barrier()
    return out


def strong_scaling(args, pipeline, ops, DeviceArray, prm, comm, dist, rank, world, synth_factory, depth_broadcast, total):
    """BASELINE.json configs[3]'s scene -- `total` object volumes + the background -- FIXED, its objects split round-robin over
    the ranks: frames/s of the same joint scene at N = 1, 2, 4, 8 (the weak line adds objects with every GPU and is flat by
    design).  Same protocol as the headline: W untimed frames, K timed ones between barrier + device synchronisation, the
    maximum over ranks.  N = 1 runs the whole scene on one GPU (the batched path in chunks of 32 table slots)."""
    import torch
    synth = synth_factory(total)
    fus = pipeline.Fusion(prm, comm)
    if comm is not None and depth_broadcast:
        fus.set_depth_broadcast(0)
    ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(total)]
    mine = [i for i in ids if fus.owns_object(i)]
    eye = np.eye(3, dtype=np.float32).reshape(-1)
    n = args.warmup + args.steps
    frames, keep = [], []
    for f in range(n):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        poses = {i: (eye, synth.sphere(i - 1, f)[0]) for i in mine}
        rm = f % prm.mask_frames == 0
        masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in mine} if rm else {}
        d = DeviceArray.from_numpy(depth)
        keep.append((d, masks))
        frames.append((ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, rm))

    def barrier():
        if dist is not None:
            dist.barrier()
    for f in range(args.warmup):
        fus.process_frame(*frames[f])
    fus.synchronize()
    barrier()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for f in range(args.warmup, n):
        fus.process_frame(*frames[f])
    fus.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    out = {
        "workload": (f"bg {args.bg_res}^3 @ {args.bg_voxel * 100:g} cm + {total} obj {args.obj_res}^3, {prm.width}x{prm.height}, full EM "
                     "association + weighted fusion: the scene of BASELINE.json configs[3], FIXED; its objects split round-robin "
                     f"over {world} rank(s)"),
        "scaling": "strong", "objects_total": total, "objects_per_gpu": [len(mine)] if dist is None else None,
        "n_gpus": world, "value": round(args.steps / elapsed, 3), "unit": "frames/s",
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "steps": args.steps, "warmup": args.warmup,
        "path": "batched" if fus.batched_chunks() else "per-volume", "launches_per_stage": fus.batched_chunks(),
        "visible_objects_last_frame": len(fus.visible_objects()),
        "note": ("frames/s of ONE joint scene; the driver's N = 1 / 2 / 4 / 8 runs each print this object, which makes the "
                 "strong-scaling curve (the builder has never had more than one GPU: no curve has been measured)"),
    }
    if dist is not None:
        per = [None] * world
        dist.all_gather_object(per, len(mine))
        out["objects_per_gpu"] = per
    fus.close()
    synth.close()
    del keep
    return out
