import sys, threading
import numpy as np
sys.path.insert(0, '.')
from emfusion_amd import devmem, pipeline
from emfusion_amd.devmem import DeviceArray
devmem.set_device(0)
world = 2
H, W = 120, 160
comms = pipeline.Communicator.local_group(world, transport="peer", max_bytes=W * H * 8)
rng = np.random.default_rng(1)
data = [rng.standard_normal((H, W)).astype(np.float32) for _ in range(world)]
got = [None] * world
keep = []
def rank_main(r):
    st = devmem.Stream(non_blocking=True)
    d = DeviceArray.from_numpy(data[r]); keep.append(d)
    comms[r].all_reduce_sum_f32(d, st)
    st.synchronize()
    got[r] = d.numpy_nosync()
ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
[t.start() for t in ths]; [t.join() for t in ths]
want = data[0] + data[1]
for r in range(world):
    g = got[r]
    print("rank", r, "== want", np.array_equal(g, want), "== own", np.array_equal(g, data[r]), "== other", np.array_equal(g, data[1 - r]),
          "== 2*own", np.array_equal(g, data[r] * 2), "nbad", int((g != want).sum()), "first bad", np.argwhere(g != want)[:3].tolist())
    bad = g != want
    if bad.any():
        i = tuple(np.argwhere(bad)[0])
        print("  got", g[i], "want", want[i], "d0", data[0][i], "d1", data[1][i])
