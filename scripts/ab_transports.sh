#!/bin/bash
# One rank, exchange path forced on: what five exchanges per frame cost through RCCL and through the direct
# peer-write transport (same kernels otherwise), against the frame without any exchange.
for mode in "" "--force-sharded --comm rccl" "--force-sharded --comm peer"; do
  for rep in 1 2; do
    python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay $mode 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-32s rep $rep: %.1f frames/s  %.4f ms/frame  (%s)' % ('$mode' or 'no exchange path', d['value'], d['ms_per_step'], d['config']['transport']))"
  done
done
