#!/bin/bash
# Does holding the background's sweep back behind the raycast's crowded phase shorten the frame?
for d in 0 60 120 180 240 300; do
  export EMF_BG_DELAY_US=$d
  for rep in 1 2; do
    python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-stats-replay 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('EMF_BG_DELAY_US=$d rep $rep: %.1f frames/s  %.4f ms/frame  raycast %.3f ms  integrate_bg %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['integrate_stream']['avg_launch_ms']))"
  done
done
