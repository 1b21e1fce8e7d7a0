#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_tracking.py tests/test_gpu_tracking_pipeline.py tests/test_golden.py tests/test_abi_exports.py -x -q -m gpu 2>&1 | tail -15
for A in ${AHEADS:-0 2}; do for rep in 1 2; do
EMF_TRACK_AHEAD=$A python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --no-target --track 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = [x for x in d['kernels'] if x['kind'] == 'track'][0]
print('EMF_TRACK_AHEAD=$A rep $rep: %.1f frames/s  %.4f ms/frame  stage %.4f ms' % (d['value'], d['ms_per_step'], k['avg_ms']))"
done; done
for A in ${AHEADS:-0 2}; do
  echo "== EMF_TRACK_AHEAD=$A"
  EMF_TRACK_AHEAD=$A bash scripts/quick_trace.sh --track --no-target 2>&1 | tail -1
  python scripts/track_launch_stats.py
done
