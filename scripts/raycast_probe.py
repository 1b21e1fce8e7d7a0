"""Diagnostic: isolate the background raycast on a warmed-up bench scene and histogram the
march-steps per ray with the oracle.  Not part of the product or the tests."""
import sys, time, ctypes as C
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from emfusion_amd import devmem, ops, pipeline
from emfusion_amd.devmem import DeviceArray, DeviceView, Event

W, H, NOBJ, WARM = 640, 480, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 40
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, NOBJ)
fus = pipeline.Fusion(prm)
ids = [fus.add_object(synth.sphere(k, 0)[0], synth.sphere(k, 0)[2]) for k in range(NOBJ)]
keep = []
for f in range(WARM):
    depth, sid = synth.render(f)
    R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32), synth.sphere(i - 1, f)[0]) for i in ids}
    d = DeviceArray.from_numpy(depth); keep.append(d)
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if f % 30 == 0 else {}
    keep.append(masks)
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, f % 30 == 0)
fus.synchronize()

def vol(which, oid, dtype=np.float32):
    ptr = C.c_void_p(); res = (C.c_int32 * 3)()
    pipeline._check("get_volume", pipeline.load().emf_fusion_get_volume(fus._h, pipeline.VOL[which], oid, C.byref(ptr), res))
    return DeviceView(ptr.value, (res[2], res[1], res[0]), dtype)

R, t = synth.camera_pose(WARM - 1)
Rm = R.reshape(3, 3).astype(np.float64); tv = t.astype(np.float64)
def rel_co(pose_t):  # volume pose = identity rotation + translation
    return Rm.astype(np.float32).reshape(-1), (tv - np.asarray(pose_t, np.float64)).astype(np.float32)

def time_raycast(tsdf, wts, fg, Rco, tco, vox, reps=10, flags=None):
    ray = DeviceArray.zeros((H, W)); vert = DeviceArray.zeros((H, W, 3)); nrm = DeviceArray.zeros((H, W, 3)); mask = DeviceArray.zeros((H, W), np.uint8)
    st = DeviceArray.zeros((4,), np.uint64)
    ops.raycast_tsdf(tsdf, None, wts, fg, ray, vert, nrm, mask, Rco, tco, K, vox, 10 * vox, st, brick_flags=flags)
    devmem.synchronize()
    s = st.numpy()
    e0, e1 = Event(), Event()
    times = []
    for _ in range(reps):
        ray.zero_(); vert.zero_(); nrm.zero_(); mask.zero_()
        devmem.synchronize()
        e0.record(); ops.raycast_tsdf(tsdf, None, wts, fg, ray, vert, nrm, mask, Rco, tco, K, vox, 10 * vox, None, brick_flags=flags); e1.record(); e1.synchronize()
        times.append(e0.elapsed_ms(e1))
    print('   stats samples=%d hits=%d gathered=%d fastfwd=%d' % tuple(int(v) for v in s))
    return np.median(times), int(s[0]), int(s[1])

bg_t, bg_w = vol("tsdf", 0), vol("weights", 0)
Rco, tco = rel_co(list(prm.volume_pose_t))
ms, S, hits = time_raycast(bg_t, bg_w, None, Rco, tco, 0.01)
print(f"bg raycast alone: {ms:.3f} ms  samples={S} ({S/(W*H):.1f}/ray) hits={hits}")
bg_f = vol("bricks", 0, np.uint8)  # raw half; ops need the buffer base = same pointer
fl = bg_f.numpy()
print("bg brick classes: mixed %.4f zero %.4f one %.4f neg %.4f" % tuple((fl == c).mean() for c in (0, 1, 2, 4)))
ms, S, hits = time_raycast(bg_t, bg_w, None, Rco, tco, 0.01, flags=bg_f)
print(f"bg raycast alone WITH flags: {ms:.3f} ms")
for i in ids:
    c, r, vs = synth.sphere(i - 1, WARM - 1)
    Rco_o, tco_o = rel_co(c)
    ms_o, S_o, h_o = time_raycast(vol("tsdf", i), vol("weights", i), vol("fgmask", i, np.uint8), Rco_o, tco_o, np.float32(vs) / np.float32(128))
    ms_f, _, _ = time_raycast(vol("tsdf", i), vol("weights", i), vol("fgmask", i, np.uint8), Rco_o, tco_o, np.float32(vs) / np.float32(128), flags=vol("bricks", i, np.uint8))
    flo = vol("bricks", i, np.uint8).numpy()
    print(f"obj {i} raycast alone: {ms_o:.3f} ms, with flags {ms_f:.3f} ms samples={S_o} hits={h_o}; bricks mixed %.3f zero %.3f one %.3f neg %.3f" % tuple((flo == c).mean() for c in (0, 1, 2, 4)))

# stage timings inside the pipeline (concurrent streams)
e0, e1 = Event(), Event()
for name, fn in [("estep", fus.stage_estep), ("raycast", fus.stage_raycast), ("integrate", fus.stage_integrate)]:
    ts = []
    for _ in range(10):
        devmem.synchronize(); t0 = time.perf_counter(); fn(); fus.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"stage {name}: host wall median {np.median(ts):.3f} ms min {min(ts):.3f}")

if len(sys.argv) > 2:
    from oracle import binding as oracle
    oracle.set_threads(0)
    tsdf = bg_t.numpy(); wts = bg_w.numpy()
    out = oracle.raycast_tsdf(tsdf, None, wts, None, W, H, Rco, tco, K, np.float32(0.01), np.float32(0.1), count_steps=True)
    st = out[4]
    print("oracle steps/ray: mean %.1f median %d p90 %d p99 %d max %d; hits %d" % (st.mean(), np.median(st), np.percentile(st, 90), np.percentile(st, 99), st.max(), out[3].sum()))
    hist, edges = np.histogram(st, bins=[0, 1, 25, 50, 100, 200, 300, 400, 600, 800, 1200, 4000])
    print(list(zip(edges[:-1].tolist(), hist.tolist())))
    # per 8x8 tile max (what a wave pays) vs mean
    t8 = st[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8)
    print("per-wave: mean of tile max %.1f, mean of tile mean %.1f" % (t8.max((1, 3)).mean(), t8.mean((1, 3)).mean()))
    print("tsdf==1 fraction", (tsdf == 1).mean(), "tsdf==0", (tsdf == 0).mean(), "tsdf==-1", (tsdf == -1).mean())

if len(sys.argv) > 3 and sys.argv[3] == "dump":
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    ptr = C.c_void_p(); res = (C.c_int32 * 3)()
    pipeline._check("get_volume", pipeline.load().emf_fusion_get_volume(fus._h, pipeline.VOL["bricks"], 0, C.byref(ptr), res))
    both = DeviceView(ptr.value, (2, res[2], res[1], res[0]), np.uint8).numpy()
    np.save("gpurun_out/bg_flags.npy", both)
    t = bg_t.numpy()
    # per-z-slab statistics of exact ones / near ones
    np.save("gpurun_out/bg_tsdf_slice_y256.npy", t[:, 256, :])
    np.save("gpurun_out/bg_tsdf_slice_z100.npy", t[100, :, :])
    print("dumped")

if len(sys.argv) > 2 and sys.argv[2] == "tail":
    from oracle import binding as oracle
    oracle.set_threads(0)
    tsdf = bg_t.numpy(); wts = bg_w.numpy()
    out = oracle.raycast_tsdf(tsdf, None, wts, None, W, H, Rco, tco, K, np.float32(0.01), np.float32(0.1), count_steps=True)
    st = out[4].astype(np.int64)
    print("steps/ray: mean %.1f p50 %d p90 %d p99 %d p99.9 %d max %d" % (st.mean(), np.percentile(st, 50), np.percentile(st, 90), np.percentile(st, 99), np.percentile(st, 99.9), st.max()))
    t8 = st.reshape(H // 8, 8, W // 8, 8)
    wmax = t8.max((1, 3)); wmean = t8.mean((1, 3))
    print("per-wave max steps: mean %.1f p50 %d p90 %d p99 %d max %d ; sum over waves of max = %d vs sum of all samples/64 = %d" % (wmax.mean(), np.percentile(wmax, 50), np.percentile(wmax, 90), np.percentile(wmax, 99), wmax.max(), wmax.sum(), st.sum() // 64))
    for thr in (300, 400, 500):
        ys, xs = np.nonzero(st > thr)
        if len(ys):
            print(f"rays > {thr} steps: {len(ys)}; x range {xs.min()}..{xs.max()}, y range {ys.min()}..{ys.max()}; "
                  f"within 8 px of border: {np.mean((xs < 8) | (xs >= W - 8) | (ys < 8) | (ys >= H - 8)):.2f}; hit rate {out[3][ys, xs].mean():.2f}; "
                  f"mean depth of hit {out[0][ys, xs][out[3][ys, xs] > 0].mean() if (out[3][ys, xs] > 0).any() else -1:.2f}")
    # which waves are the slowest and what do their rays go through
    iy, ix = np.unravel_index(np.argsort(wmax.ravel())[-8:], wmax.shape)
    for a, b in zip(iy, ix):
        tile = st[a * 8:a * 8 + 8, b * 8:b * 8 + 8]
        print("slow wave tile (%d,%d): max %d mean %.0f min %d; ray length of hits mean %.2f" % (b * 8, a * 8, tile.max(), tile.mean(), tile.min(), out[0][a * 8:a * 8 + 8, b * 8:b * 8 + 8].mean()))
