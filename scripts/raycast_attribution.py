"""Where a step of the product's ray march spends its clocks, in situ.
Attribution build: make EXTRA="-DEMF_RAY_TRACE -DEMF_MARCH_STAMP" (scripts/run_attribution.sh does it on the GPU
box).  Every wave of k_raycast_batched stamps s_memtime at the top of each loop iteration, when its four corner
gathers have been issued, and when they have all returned (an explicit s_waitcnt vmcnt(0)); the per-wave sums and a
histogram of the wait come back through the trace records of scripts/raycast_timeline.py.  The stamps cost the step
~3 s_memtime + ~25 VALU instructions: the build's step is longer than the product's (both are printed).
Usage (GPU box): python scripts/raycast_attribution.py [frames] [--no-bg-overlap]"""
import ctypes as C
import os
import sys
import numpy as np

if "--no-bg-overlap" in sys.argv:
    os.environ["EMF_BG_OVERLAP"] = "0"
    sys.argv.remove("--no-bg-overlap")
sys.path.insert(0, ".")
import torch  # noqa: F401,E402  (one HIP runtime, see bench.py)
from emfusion_amd import _lib, ops, pipeline  # noqa: E402
from emfusion_amd.devmem import DeviceArray  # noqa: E402

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
                  ("tile", "<u4"), ("wave", "<u4"), ("smax", "<u4"), ("ssum", "<u4"), ("lanes", "<u4"),
                  ("issue", "<u8"), ("wait", "<u8"), ("rest", "<u8"), ("lissue", "<u8"), ("lwait", "<u8"),
                  ("lrest", "<u8"), ("iters", "<u4"), ("hist", "<u4", 6), ("pad", "<u4")])
buf = np.zeros(32768, rec_t)
lib = _lib.load()
lib.emf_hip_debugFetchRayTrace.argtypes = [C.c_void_p, C.c_size_t]
rc = lib.emf_hip_debugFetchRayTrace(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
r = buf[buf["t1"] > 0]
r = r[r["t1"] > r["t1"].max() - 200000]  # the last launch only
r = r[r["iters"] > 0]
t0 = r["t0"].min()
print(f"waves with loop iterations {len(r)}, launch span {(r['t1'].max() - t0) / 100.0:.1f} us "
      f"(EMF_BG_OVERLAP={os.environ.get('EMF_BG_OVERLAP', 'default')})")
tot = (r["issue"] + r["wait"] + r["rest"]).astype(np.float64)
dur_us = (r["t1"] - r["t0"]) / 100.0
big = r["iters"] > 100
print(f"s_memtime ticks per us of wave life (waves > 100 iterations; loop clocks / whole wave duration): "
      f"p50 {np.median(tot[big] / dur_us[big]):.0f}")
TX, TY = W // 16, H // 16
ty, tx = r["tile"] // TX, r["tile"] % TX
border = (r["model"] == 0) & ((tx == 0) | (tx == TX - 1) | (ty == 0) | (ty == TY - 1))
groups = [("background, border tiles", border), ("background, interior", (r["model"] == 0) & ~border),
          ("objects", r["model"] > 0), ("background waves > 400 iterations", (r["model"] == 0) & (r["iters"] > 400))]
edges = ["<200", "200-400", "400-700", "700-1200", "1200-2500", ">=2500"]
print(f"{'group':36s} {'waves':>6s} {'iters':>9s} | clocks per iteration: {'issue':>6s} {'wait':>6s} {'rest':>6s} {'total':>6s} "
      f"| late (iteration >= 256): issue wait rest | wait histogram {' '.join(edges)}")
for name, sel in groups:
    if not sel.any():
        continue
    it = r["iters"][sel].sum()
    li = np.maximum(r["iters"][sel].astype(np.int64) - 256, 0).sum()
    h = r["hist"][sel].sum(axis=0) / it
    late = (f"{r['lissue'][sel].sum() / li:6.0f} {r['lwait'][sel].sum() / li:6.0f} {r['lrest'][sel].sum() / li:6.0f}"
            if li else "     -      -      -")
    print(f"{name:36s} {sel.sum():6d} {it:9d} | {r['issue'][sel].sum() / it:28.0f} {r['wait'][sel].sum() / it:6.0f} "
          f"{r['rest'][sel].sum() / it:6.0f} {tot[sel].sum() / it:6.0f} | {late} | " + " ".join(f"{x:.3f}" for x in h))
# per launch phase: waves grouped by when they ended
end_us = (r["t1"] - t0) / 100.0
print("by end time of the wave (us from the launch's first wave):")
for lo, hi in [(0, 100), (100, 200), (200, 300), (300, 400), (400, 1e9)]:
    sel = (end_us >= lo) & (end_us < hi) & (r["model"] == 0)
    if not sel.any():
        continue
    it = r["iters"][sel].sum()
    print(f"  {lo:4.0f}-{hi if hi < 1e9 else 0:4.0f}: waves {sel.sum():5d}, clocks per iteration issue {r['issue'][sel].sum() / it:5.0f} "
          f"wait {r['wait'][sel].sum() / it:5.0f} rest {r['rest'][sel].sum() / it:5.0f}; wave ns per iteration "
          f"{1e3 * dur_us[sel].sum() / it:5.0f}")
fus.close(); synth.close()
