#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() {
  python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --no-target --track --event-stride ${STRIDE:-4} 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = [x for x in d['kernels'] if x['kind'] == 'track']
print('$1: %.1f frames/s  %.4f ms/frame%s' % (d['value'], d['ms_per_step'], '  stage %.4f ms' % k[0]['avg_ms'] if k else ''))"
}
for rep in 1 2; do for W in 0 1 2; do EMF_TRACK_WIDE=$W run "EMF_TRACK_WIDE=$W"; done; done
for W in 1 2; do
echo "== EMF_TRACK_WIDE=$W"
EMF_TRACK_WIDE=$W bash scripts/quick_trace.sh --track --no-target 2>&1 | tail -1
python scripts/track_launch_stats.py
done
