"""What the cross-GPU exchanges cost a frame when each of them takes tens of microseconds (SURVEY 8e:
1-5 MB messages over xGMI are latency-, not bandwidth-bound), measured on ONE GPU: the sharded path runs
through a real 1-rank RCCL communicator (EMF_FORCE_SHARDED=1) wrapped in the latency model
(emf::makeDelayedCommunicator: every exchange first keeps its stream busy for N microseconds).

Per frame the schedule issues FIVE exchanges: the depth broadcast, one all-reduce(sum) of the object
normaliser partials per E-step (three), and ONE grouped exchange per raycast (all-reduce(min) of the
nearest-hit keys + the background raycast's row bands).  Measured with 30 us each: the frame grows by two to
five of them (+70 ... +150 us) -- whether the grouped exchange behind the raycast is covered by the background's
sweep on the second stream depends on which of the two streams ends the frame, and that on the stream-to-queue
mapping of the process.  (Moving the last E-step's
all-reduce to a stream of its own beside the raycast was built, measured -- +90 us instead of +70 -- and
removed: the background's sweep needs the normalised weights and is the other half of the critical path.)"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from tests.parity_util import to_dev  # noqa: E402

pytestmark = pytest.mark.gpu
W, H = 640, 480
LATENCY_US = 30


def _run(delay_us, frames=90, warm=10):
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    os.environ["EMF_FORCE_SHARDED"] = "1"
    try:
        base = pipeline.Communicator(pipeline.Communicator.unique_id(), 0, 1)
        comm = base.delayed(delay_us)
        prm = pipeline.make_params(W, H, 512, 0.01, 128)
        synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), 4, seed=0xE3F5)
        fus = pipeline.Fusion(prm, comm)
        fus.set_depth_broadcast(0)
        ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
        inputs = []
        for f in range(frames):
            depth, sid = synth.render(f)
            R, t = synth.camera_pose(f)
            poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
            masks = {i: to_dev((sid == i).astype(np.uint8)) for i in ids} if f == 0 else {}
            inputs.append((to_dev(depth), R, t, poses, masks))

        def step(f):
            d, R, t, poses, masks = inputs[f]
            fus.process_frame(image_view(d), R, t, poses, {i: image_view(m) for i, m in masks.items()}, f == 0)
        for f in range(warm):
            step(f)
        fus.synchronize()
        x0 = comm.exchanges()
        import gc
        gc.collect()
        gc.disable()  # a gen-2 collection inside the timed loop would cost tens of milliseconds (bench.py, round 4)
        t0 = time.perf_counter()
        for f in range(warm, frames):
            step(f)
        fus.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / (frames - warm)
        gc.enable()
        per_frame = (comm.exchanges() - x0) / (frames - warm)
        import xxhash
        out = dict(ms=ms, exchanges=per_frame,
                   digest=[xxhash.xxh3_128(a.tobytes()).hexdigest() for a in
                           (fus.image("raylengths"), fus.image("segmentation"), fus.image("bg_assoc"), fus.volume("tsdf", 1))])
        fus.close()
        comm.close()
        base.close()
        synth.close()
        return out
    finally:
        os.environ.pop("EMF_FORCE_SHARDED", None)


def test_five_exchanges_per_frame_and_what_they_cost(dev):
    # a process of its own: which hardware queue a stream lands on depends on how many streams the process
    # has created before, and two of the frame's streams on one queue serialise the frame
    run = subprocess.run([sys.executable, str(Path(__file__).resolve())], cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    res = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("LATENCY_RESULT ")][-1][15:])
    free, slow = res["free"], res["slow"]
    for r in (free, slow):
        assert r["exchanges"] == 5.0, r["exchanges"]
    assert free["digest"] == slow["digest"]  # the latency model changes no result
    added = slow["ms"] - free["ms"]
    print(f"frame {free['ms']:.3f} ms; +{added * 1e3:.0f} us with {LATENCY_US} us per exchange "
          f"({100 * added / free['ms']:.0f} %)")
    # at most the five of them (+ their launches).  How many are exposed depends on which of the two streams ends the
    # frame: measured between two (the grouped exchange behind the raycast covered by the sweep on the second
    # stream) and all five, from one stream-to-queue mapping to the next
    # (bounds with room for the run-to-run spread of a 0.6-0.8 ms frame: the claim is "no more than the five")
    assert -0.05 < added < 8 * LATENCY_US * 1e-3, (free["ms"], slow["ms"])


if __name__ == "__main__":
    from emfusion_amd import devmem
    devmem.set_device(0)
    out = dict(free=_run(0), slow=_run(LATENCY_US))
    print("LATENCY_RESULT " + json.dumps(out), flush=True)
