#!/usr/bin/env python
"""How much a frame of configs[1] depends on the process's stream history: the frame's
streams (main / aux / lists) overlap the raycast with the background's sweep, and whether they do depends on
how HIP maps streams to hardware queues.  Every scenario runs in a process of its own:

  fresh        nothing before the Fusion
  cycles8      8 x create + destroy of a stream first
  foreign8     8 live foreign streams (an embedding application: the reference creates one cv::cuda::Stream per
               object, include/EMFusion/core/EMFusion.h:471)
  foreign3     3 live foreign streams
  both         cycles8 + foreign8
  second       a Fusion built, run for 5 frames and closed first; the measured one is the second of the process

    python scripts/stream_history_probe.py            # all scenarios, prints one line each + a JSON summary
"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
SCENARIOS = ("fresh", "cycles8", "foreign8", "foreign3", "both", "second")


def run(scenario, frames=110, warm=30):
    import numpy as np
    from emfusion_amd import devmem, pipeline
    from emfusion_amd.ops import image_view
    from emfusion_amd.devmem import DeviceArray
    devmem.set_device(0)
    keep = []
    if scenario in ("cycles8", "both"):
        for _ in range(8):
            s = devmem.Stream(non_blocking=True)
            del s
    if scenario in ("foreign8", "both"):
        keep = [devmem.Stream(non_blocking=True) for _ in range(8)]
    if os.environ.get("PROBE_TORCH_FIRST"):
        import torch  # noqa: F401  (bench.py's situation: the torch wheel's HIP runtime serves the process)
        torch.cuda.is_available()
    if scenario.startswith("foreign") and scenario not in ("foreign8",):
        keep = [devmem.Stream(non_blocking=True) for _ in range(int(scenario[7:]))]
    W, H = 640, 480
    prm = pipeline.make_params(W, H, 512, 0.01, 128)
    synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), 4, seed=0xE3F5)
    inputs = []
    for f in range(frames):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        inputs.append((DeviceArray.from_numpy(depth), R, t, sid))

    def make():
        fus = pipeline.Fusion(prm, None)
        ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
        return fus, ids

    def step(fus, ids, f):
        d, R, t, sid = inputs[f]
        poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
        masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if f == 0 else {}
        fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, f == 0)
        return masks

    if scenario == "second":
        fus, ids = make()
        for f in range(5):
            step(fus, ids, f)
        fus.synchronize()
        fus.close()
    fus, ids = make()
    if os.environ.get("PROBE_TIMERS"):  # what bench.py does around its timed region
        fus.kernel_timers_enable(2000)
        fus.kernel_timers_select(["raycast", "integrate_bg", "track"])
        fus.kernel_timers_stride(4)
    for f in range(warm):
        step(fus, ids, f)
    fus.synchronize()
    import gc
    if not os.environ.get("PROBE_KEEP_GC"):  # a gen-2 collection inside the loop costs 15-45 ms (bench.py, round 4)
        gc.collect()
        gc.disable()
    t0 = time.perf_counter()
    for f in range(warm, frames):
        step(fus, ids, f)
    fus.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / (frames - warm)
    gc.enable()
    fus.close()
    synth.close()
    del keep
    return ms


if __name__ == "__main__":
    if len(sys.argv) > 1 and (sys.argv[1] in SCENARIOS or sys.argv[1].startswith("foreign")):
        print("PROBE_RESULT %s %.4f" % (sys.argv[1], run(sys.argv[1])), flush=True)
        sys.exit(0)
    out = {}
    todo = [(sc, {}) for sc in SCENARIOS]
    if "--matrix" in sys.argv:  # where the loss starts, and whether a priority class of their own shields the streams
        todo = [("fresh", {})] + [(f"foreign{n}", {}) for n in (4, 5, 6, 7, 8, 12)]
        todo += [("fresh", {"EMF_PRIO_MAIN": "high"}), ("foreign8", {"EMF_PRIO_MAIN": "high"}),
                 ("foreign8", {"EMF_PRIO_MAIN": "high", "EMF_PRIO_LISTS": "high"}),
                 ("foreign8", {"EMF_PRIO_MAIN": "high", "EMF_PRIO_LISTS": "low"}),
                 ("foreign8", {"EMF_PRIO_LISTS": "low"}),
                 ("foreign8", {"GPU_MAX_HW_QUEUES": "8"}), ("foreign8", {"GPU_MAX_HW_QUEUES": "16"})]
    if "--matrix2" in sys.argv:  # every count of foreign streams, with and without a priority class of its own for `main`
        todo = []
        for n in range(0, 10):
            todo += [(f"foreign{n}", {}), (f"foreign{n}", {"EMF_PRIO_MAIN": "high"})]
    if "--matrix3" in sys.argv:  # is there a choice of priority classes that does not care about the history?
        cfgs = [{"EMF_PRIO_AUX": "normal"}, {"EMF_PRIO_LISTS": "low"}, {"EMF_PRIO_MAIN": "high", "EMF_PRIO_LISTS": "high"},
                {"EMF_PRIO_MAIN": "high", "EMF_PRIO_LISTS": "low"}, {"EMF_PRIO_MAIN": "high", "EMF_PRIO_AUX": "normal", "EMF_PRIO_LISTS": "low"}]
        todo = [(f"foreign{n}", c) for c in cfgs for n in range(0, 10)]
    for rep in range(1 if ("--matrix2" in sys.argv or "--matrix3" in sys.argv) else 2):
        for sc, extra in todo:
            r = subprocess.run([sys.executable, __file__, sc], capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, **extra))
            sc = sc + ("" if not extra else " " + ",".join(f"{k}={v}" for k, v in extra.items()))
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("PROBE_RESULT")]
            if not line:
                print(sc, "FAILED", r.stderr[-500:])
                continue
            ms = float(line[-1].split()[2])
            out.setdefault(sc, []).append(ms)
            print(f"{sc:60s} rep {rep}: {ms:.4f} ms/frame  {1e3 / ms:.0f} frames/s", flush=True)
    print("PROBE_SUMMARY " + json.dumps(out))
