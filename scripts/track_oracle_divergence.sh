#!/bin/bash
# GPU box: the tracked pipeline test (tests/test_gpu_tracking_pipeline.py, set NFRAMES there) frame by frame: how far the HIP
# classes' poses are from the oracle's and from the truth, with and without rescaled sums
cd ${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/dev.py <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import tests.test_gpu_tracking_pipeline as T
from oracle import binding
from emfusion_amd import devmem
binding.lib(); devmem.set_device(0)
g = T.run.__wrapped__(binding, devmem)
fus, orc, ids, rows = next(g)
for r in rows[1:]:
    print(r["f"], "cam R %.2e t %.2e" % (np.abs(r["cam"][0] - r["ocam"][0]).max(), np.abs(r["cam"][1] - r["ocam"][1]).max()),
          "obj t", " ".join("%.2e" % np.abs(r["obj"][i][1] - r["oobj"][i][1]).max() for i in ids),
          "true cam %.4f" % np.linalg.norm(r["cam"][1] - r["cam_true"][1]), "steps", r["res"][0]["iterations"], [r["res"][i]["iterations"] for i in ids])
PY
for V in 1 0; do
  echo "== EMF_TRACK_RESCALE=$V"; EMF_TRACK_RESCALE=$V python /tmp/dev.py 2>&1 | tail -12
done
