"""Dump the LM states (A, b, err, pose, mu ...) of tests/test_gpu_tracking.py's world after 1, 2, 5 and 100
iterations to gpurun_out/<name>.npy -- to compare two builds of the tracking kernels byte by byte
(python scripts/dump_track_states.py NAME; test infrastructure: uses the oracle to integrate the scene)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import binding
from emfusion_amd import devmem, ops
import tests.test_gpu_tracking as T

binding.lib(); binding.set_threads(8); devmem.set_device(0)
world = T.world.__wrapped__(binding)
out = []
for n in (1, 2, 5, 100):
    tr = T.DeviceTracker(ops, world, [0, 1])
    tr.iterate(n)
    raw = T.to_np(tr.states).copy()
    out.append(raw)
np.save("gpurun_out/%s.npy" % sys.argv[1], np.stack(out))
print(sys.argv[1], [int(x.view(np.uint32).sum()) for x in out])
if len(sys.argv) > 2:  # compare the legacy fields (first 344 bytes of a state) with another dump
    a, b = np.load("gpurun_out/%s.npy" % sys.argv[2]), np.stack(out)
    sa, sb = a.shape[1] // 2, b.shape[1] // 2
    for i, n in enumerate((1, 2, 5, 100)):
        for m in range(2):
            x, y = a[i, m * sa:m * sa + 344], b[i, m * sb:m * sb + 344]
            diff = np.nonzero(x.view(np.uint32) != y.view(np.uint32))[0]
            print("iterations", n, "model", m, "identical" if diff.size == 0 else "DIFFERENT words %s" % diff[:12])
