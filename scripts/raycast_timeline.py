"""Timeline of the batched raycast on the bench scene (trace build: make EXTRA=-DEMF_RAY_TRACE).
Runs the bench pipeline for N frames, then fetches the per-wave records of the LAST raycast launch:
start / end (100 MHz clock), model, loop samples (max and sum over lanes), CU.
Usage (GPU box): python scripts/raycast_timeline.py [frames]"""
import ctypes as C
import sys
import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401  (one HIP runtime, see bench.py)
from emfusion_amd import _lib, ops, pipeline
from emfusion_amd.devmem import DeviceArray

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W, H = 640, 480
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, 4, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
keep = []
for f in range(frames):
    depth, sid = synth.render(f)
    R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
    run_masks = f % prm.mask_frames == 0
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if run_masks else {}
    d = DeviceArray.from_numpy(depth)
    keep.append((d, masks))
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, run_masks)
fus.synchronize()

rec_t = np.dtype([("t0", "<u8"), ("t1", "<u8"), ("hw", "<u4"), ("xcc", "<u4"), ("model", "<u4"),
                  ("tile", "<u4"), ("wave", "<u4"), ("smax", "<u4"), ("ssum", "<u4"), ("lanes", "<u4")])
buf = np.zeros(32768, rec_t)
lib = _lib.load()
lib.emf_hip_debugFetchRayTrace.argtypes = [C.c_void_p, C.c_size_t]
rc = lib.emf_hip_debugFetchRayTrace(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
r = buf[buf["t1"] > 0]
r = r[r["t1"] > r["t1"].max() - 200000]  # the last launch only (the grid changes from frame to frame: stale slots)
t0 = r["t0"].min()
span = (r["t1"].max() - t0) / 100.0
print(f"waves recorded {len(r)}, span {span:.1f} us")
dur = (r["t1"] - r["t0"]) / 100.0
for m in range(5):
    s = r["model"] == m
    if not s.any():
        continue
    work = s & (r["smax"] > 0)
    print(f" model {m}: waves {s.sum()}, with samples {work.sum()}, start p50 {np.median((r['t0'][s] - t0) / 100.0):.1f} us, "
          f"end max {((r['t1'][s] - t0) / 100.0).max():.1f} us; working waves: dur p50 {np.median(dur[work]):.1f} p90 {np.percentile(dur[work], 90):.1f} "
          f"max {dur[work].max():.1f} us; smax p50 {np.median(r['smax'][work]):.0f} p90 {np.percentile(r['smax'][work], 90):.0f} max {r['smax'][work].max()}")
w = r["smax"] > 0
ns_per_step = 1e3 * dur[w] / r["smax"][w]
print(f" ns per loop step (wave duration / max samples): p10 {np.percentile(ns_per_step, 10):.0f} p50 {np.median(ns_per_step):.0f} p90 {np.percentile(ns_per_step, 90):.0f}")
long_ = w & (r["smax"] > 450)
if long_.any():
    print(f" waves with > 450 samples: {long_.sum()}, ns/step p50 {np.median(1e3 * dur[long_] / r['smax'][long_]):.0f}, lanes-with-samples p50 {np.median(r['lanes'][long_]):.0f}, "
          f"mean samples per lane {np.mean(r['ssum'][long_] / 64):.0f}")
print(" resident waves over time (all / with samples):")
line = []
for k in range(40):
    t = t0 + int((k + 0.5) / 40 * span * 100)
    a = (r["t0"] <= t) & (t < r["t1"])
    line.append(f"{a.sum()}/{(a & w).sum()}")
print("  " + " ".join(line))
# how long would the kernel be if it were only as long as its longest wave chain at the median step time?
print(f" longest wave: {r['smax'].max()} samples x median {np.median(ns_per_step):.0f} ns = {r['smax'].max() * np.median(ns_per_step) / 1e3:.0f} us")
fus.close(); synth.close()
# lane utilisation: a wave loops until its longest ray is done
bgw = (r["model"] == 0) & (r["smax"] > 0)
print(f" background: sum of lane samples {int(r['ssum'][bgw].sum())}, sum over waves of 64 x max samples {int(64 * r['smax'][bgw].sum())}: "
      f"lane utilisation {r['ssum'][bgw].sum() / (64.0 * r['smax'][bgw].sum()):.2f}; wave-steps {int(r['smax'][bgw].sum())}")
# where the long background waves sit in the image (16 x 16 tiles, 40 x 30 of them at VGA)
lw = np.nonzero((r["model"] == 0) & (r["smax"] > 400))[0]
tx, ty = r["tile"][lw] % 40, r["tile"][lw] // 40
print(f" background waves with > 400 samples: {len(lw)}; tile columns {np.bincount(tx, minlength=40).tolist()}")
print(f"   tile rows {np.bincount(ty, minlength=30).tolist()}")
order = np.argsort(-r["smax"][lw])[:12]
print("   longest:", [(int(tx[i]), int(ty[i]), int(r["wave"][lw][i]), int(r["smax"][lw][i])) for i in order])

# the last waves to finish: who are they?
order = np.argsort(r["t1"])[::-1][:24]
print(" last waves to finish (end us, start us, model, tile x, tile y, sub-tile, max samples, ns/step):")
tilesX = (W + 15) // 16
for k in order:
    print(f"   {(r['t1'][k] - t0) / 100.0:7.1f} {(r['t0'][k] - t0) / 100.0:7.1f}  m{r['model'][k]} ({r['tile'][k] % tilesX:2d},{r['tile'][k] // tilesX:2d}) w{r['wave'][k]} "
          f"{r['smax'][k]:4d} {1e3 * dur[k] / max(r['smax'][k], 1):6.0f}")
