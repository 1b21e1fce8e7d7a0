#!/usr/bin/env python
"""Which of a process's streams share a hardware queue?  Two one-wave kernels that spin for T microseconds, enqueued
back to back on two streams, take T when the streams are served by different hardware queues and 2 T when they share
one (a queue is in order).  For n = 0 .. 9 foreign streams created first, the script creates what an emf::EMFusion
creates (main: normal priority, aux: lowest, lists: normal), prints the pairwise verdicts among {null, main, aux,
lists, foreign[0]} -- to be read next to scripts/stream_history_probe.py's frame times for the same n."""
import ctypes as C
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
T_US = 300


def run(n):
    from emfusion_amd import _lib, devmem
    devmem.set_device(0)
    hip, lib = devmem._hip, _lib.load()
    foreign = [devmem.Stream(non_blocking=True) for _ in range(n)]
    lo, hi = C.c_int(), C.c_int()
    hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))

    def mk(prio):
        h = C.c_void_p()
        assert hip.hipStreamCreateWithPriority(C.byref(h), 1, prio) == 0
        return h
    mid = (lo.value + hi.value) // 2
    streams = {"null": C.c_void_p(0), "main": mk(mid), "aux": mk(lo.value), "lists": mk(mid)}
    if foreign:
        streams["f0"] = foreign[0].handle
    names = list(streams)

    def pair(a, b):
        best = 1e9
        for _ in range(3):
            devmem.synchronize()
            t0 = time.perf_counter()
            lib.emf_hip_spinDelay(T_US, streams[a])
            lib.emf_hip_spinDelay(T_US, streams[b])
            hip.hipStreamSynchronize(streams[a])
            hip.hipStreamSynchronize(streams[b])
            best = min(best, (time.perf_counter() - t0) * 1e6)
        return best
    out = []
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            us = pair(a, b)
            out.append(f"{a}+{b}:{'SHARED' if us > 1.6 * T_US else 'apart'}({us:.0f})")
    print(f"QUEUES n={n} " + " ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]))
    else:
        for n in range(10):
            r = subprocess.run([sys.executable, __file__, str(n)], capture_output=True, text=True, timeout=300)
            print("\n".join(l for l in r.stdout.splitlines() if l.startswith("QUEUES")) or r.stderr[-400:], flush=True)
