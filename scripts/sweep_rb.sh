#!/bin/bash
# Batched raycast variants: wave-level loop vs per-lane loop, 1 vs 4 waves per workgroup.
# Each variant is its own libemf_hip.so under /tmp/rb_*; run on the GPU box via gpurun.
cd /root/repo
for v in "W4:-DEMF_RB_WAVES=4" "W1:-DEMF_RB_WAVES=1" "P4:-DEMF_RB_WAVES=4 -DEMF_RB_PLAIN" "P1:-DEMF_RB_WAVES=1 -DEMF_RB_PLAIN"; do
  name=${v%%:*}; flags=${v#*:}
  touch emfusion_amd/csrc/batched.hip
  make -s -C emfusion_amd/csrc EXTRA="$flags" >/dev/null 2>&1 || { echo "$name build failed"; continue; }
  timeout 150 python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:x['avg_ms'] for x in d['kernels']}; print('$name', d['value'], 'fps', d['ms_per_step'], 'ms; raycast', k['raycast'], 'integrate', k['integrate'])"
done
