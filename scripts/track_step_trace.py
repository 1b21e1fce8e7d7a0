"""Where a k_track_step launch spends its time (build with EXTRA=-DEMF_TRACK_TRACE=<workgroup>): stamps of one
workgroup of model 0 for the first launches of a 20-iteration call on tests/test_gpu_tracking.py's world
scaled to 640x480.  Columns: us from kernel entry to [reduce done, barrier, lm_advance done, the pass's barrier (probe
builds overwrite "state stored" with it), first pixel terms done, end, slots tested (pass 1)]."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import binding
from emfusion_amd import devmem, ops, _lib
import tests.test_gpu_tracking as T
binding.lib(); binding.set_threads(8); devmem.set_device(0)
T.W, T.H = 640, 480
T.K = T.intrinsics(T.W, T.H)
world = T.world.__wrapped__(binding)
tr = T.DeviceTracker(ops, world, [int(a) for a in sys.argv[1:]] or [0])
tr.iterate(20)
raw = T.to_np(tr.scratch)[:tr.per_model]
px, nb = T.W * T.H, -(-T.W * T.H // 1216)
off = (4 * px + 2 * nb * 30) * 4 + C.sizeof(_lib.EmfTrackState)
st = raw[off:off + 24 * 64].view(np.int64).reshape(24, 8)
wg = raw[off + 24 * 64:off + 24 * 64 + 16 * nb].view(np.int64).reshape(nb, 2)
if wg[:, 0].any():
    t0 = wg[wg[:, 0] > 0, 0].min()
    s0, e0 = (wg[:, 0] - t0) / 100.0, (wg[:, 1] - t0) / 100.0
    print("launch 5, all %d workgroups: start us min/med/max %.2f %.2f %.2f   end us min/med/max %.2f %.2f %.2f" % (
        nb, s0.min(), np.median(s0), s0.max(), e0.min(), np.median(e0), e0.max()))
    order = np.argsort(e0)
    print("  last to end:", [(int(i), round(float(s0[i]), 2), round(float(e0[i]), 2)) for i in order[-6:]])
    print("  first to end:", [(int(i), round(float(s0[i]), 2), round(float(e0[i]), 2)) for i in order[:4]])
for i, r in enumerate(st):
    if r[0] == 0: continue
    print(i, " ".join("%6.2f" % ((x - r[0]) / 100.0) if x else "   -  " for x in r[1:8]))
wv = raw[off + 24 * 64 + 16 * nb:off + 24 * 64 + 16 * nb + 16 * 8].view(np.int64)
if wv.any() and st[6][0]:
    print("launch 6, waves' arrival at the barrier behind the column sums (us from kernel entry):",
          " ".join("%.2f" % ((x - st[6][0]) / 100.0) for x in wv))
