#!/bin/bash
# GPU box: pixels per workgroup / partial-sum row of k_track_step (EMF_TRACK_BLOCK), tracked bench of configs[1]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for B in ${BLOCKS:-1024 640 768 512}; do
  touch emfusion_amd/csrc/tracking.hip
  make -s -C emfusion_amd/csrc -j8 EXTRA="-DEMF_TRACK_BLOCK=$B $TRACK_EXTRA" > /tmp/tb.log 2>&1 || { tail -5 /tmp/tb.log; exit 1; }
  for rep in 1 2; do
    python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --no-target --track 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = [x for x in d['kernels'] if x['kind'] == 'track'][0]
print('EMF_TRACK_BLOCK=$B rep $rep: %.1f frames/s  %.4f ms/frame  stage %.4f ms' % (d['value'], d['ms_per_step'], k['avg_ms']))"
  done
  if [ -n "$TRACK_TESTS" ]; then python -m pytest tests/test_gpu_tracking.py tests/test_gpu_tracking_pipeline.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -3; fi
done
touch emfusion_amd/csrc/tracking.hip
