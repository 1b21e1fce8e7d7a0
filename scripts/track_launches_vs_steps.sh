#!/bin/bash
# GPU box: launches of every tracking stage of the tracked bench against the LM steps it judged (EMF_TRACK_LOG), in tens of frames
cd ${GRAFT_REPO_ROOT:-/root/repo}
EMF_TRACK_LOG=1 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --no-target --track 2>&1 | grep "track stage" > /tmp/stages.txt
python - <<'PY'
import re
cam, obj = [], []
for l in open("/tmp/stages.txt"):
    g = re.search(r"first (\d+) count (\d+): launches (\d+), most steps (\d+), most accepted (\d+)", l)
    (cam if g.group(1) == "0" else obj).append(tuple(int(g.group(i)) for i in (3, 4, 5)))
for name, v in (("camera", cam), ("objects", obj)):
    print(name, "stages", len(v))
    for a in range(0, len(v), 10):
        w = v[a:a + 10]
        print("  frames %3d..%3d: launches %5.1f  steps %5.1f  accepted %5.1f   launches - steps %5.1f" % (
            a, a + len(w) - 1, sum(x[0] for x in w) / len(w), sum(x[1] for x in w) / len(w), sum(x[2] for x in w) / len(w),
            sum(x[0] - x[1] for x in w) / len(w)))
PY
