"""Accept / reject sequence of every LM stage on the tracked bench workload (run with EMF_TRACK_WINDOW=0
EMF_TRACK_CHUNK=1 EMF_TRACK_LOG=1: the host polls the states after every iteration and the library prints them).
Prints, per model class, the histogram of rejection-run lengths: how many launches a flow that evaluates the
next damping values of a rejected step in the same launch could save."""
import os, sys, re, subprocess, collections
if os.environ.get("EMF_TRACK_LOG") is None:
    env = dict(os.environ, EMF_TRACK_WINDOW="0", EMF_TRACK_CHUNK="1", EMF_TRACK_LOG="1")
    out = subprocess.run([sys.executable, __file__], env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
    pat = re.compile(r"track model (\d+): iterations (\d+) accepted (\d+) rho (\S+) mu (\S+) nu (\S+) converged (\d)")
    last, seq = {}, collections.defaultdict(list)   # per model: current stage's verdicts
    runs = {"camera": collections.Counter(), "objects": collections.Counter()}
    tot = {"camera": [0, 0, 0], "objects": [0, 0, 0]}   # iterations, rejected, rejected in the terminal run
    shown = 0
    def close(m):
        global shown
        v = seq.pop(m, [])
        if not v: return
        cls = "camera" if m == 0 else "objects"
        s = "".join(v)
        if shown < 12: print("model %d: %s" % (m, s)); shown += 1
        for r in re.findall(r"r+", s): runs[cls][len(r)] += 1
        tot[cls][0] += len(s); tot[cls][1] += s.count("r"); tot[cls][2] += len(s) - len(s.rstrip("r"))
    for line in out.splitlines():
        g = pat.match(line)
        if not g: continue
        m, it, acc = int(g.group(1)), int(g.group(2)), int(g.group(3))
        p = last.get(m)
        if p is None or it < p[0]:   # a new stage
            close(m); p = (0, 0)
        if it > p[0]:
            seq[m].extend("a" if acc > p[1] else "r" for _ in range(it - p[0]))
        last[m] = (it, acc)
    for m in list(seq): close(m)
    for cls in runs:
        it, rj, term = tot[cls]
        print("%s: %d iterations, %d rejected (%d of them in the run that ends the stage); rejection runs by length: %s" % (
            cls, it, rj, term, dict(sorted(runs[cls].items()))))
        print("   launches saved if a rejected step's successors were judged in the same launch, K candidates per launch:",
              {K: sum(n * (L - -(-L // K)) for L, n in runs[cls].items()) for K in (2, 3, 4)})
    sys.exit(0)
import numpy as np
sys.path.insert(0, ".")
from emfusion_amd import ops, pipeline
from emfusion_amd.devmem import DeviceArray
W, H = 640, 480
prm = pipeline.make_params(W, H, 512, 0.01, 128)
K = np.array(prm.K, np.float32)
synth = pipeline.SyntheticStream(W, H, K, 4, seed=0xE3F5)
fus = pipeline.Fusion(prm, None)
ids = [fus.add_object(*[synth.sphere(k, 0)[i] for i in (0, 2)]) for k in range(4)]
fus.set_tracking(True, True)
for f in range(int(os.environ.get("FRAMES", "40"))):
    depth, sid = synth.render(f); R, t = synth.camera_pose(f)
    poses = {i: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(i - 1, f)[0]) for i in ids}
    rm = f % prm.mask_frames == 0
    masks = {i: DeviceArray.from_numpy((sid == i).astype(np.uint8)) for i in ids} if rm else {}
    d = DeviceArray.from_numpy(depth)
    fus.process_frame(ops.image_view(d), R, t, poses, {i: ops.image_view(m) for i, m in masks.items()}, rm)
    fus.synchronize()
