#!/bin/bash
# A/B on one box: does the number of hardware queues HIP multiplexes the streams onto change the frame rate?
# (tests/dynamic_probe.py found two of this process's streams sharing a queue with the default of 4)
for q in default 2 4 8 16; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for rep in 1 2; do
    python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-stats-replay 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('GPU_MAX_HW_QUEUES=$q rep $rep: %.1f frames/s, raycast %.3f ms, integrate_bg %.3f ms' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['integrate_stream']['avg_launch_ms']))"
  done
done
