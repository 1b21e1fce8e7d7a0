#!/bin/bash
# Usage: sweep_variants.sh "NAME:FLAGS" ...   -- builds libemf_hip.so per variant (make EXTRA=FLAGS)
# and runs the bench (no CPU baseline) for each; run on the GPU box via gpurun.  Compare variants
# only within ONE call: boxes differ by up to 2x.
cd /root/repo
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  touch emfusion_amd/csrc/*.hip
  make -s -C emfusion_amd/csrc -j8 EXTRA="$flags" >/tmp/build_$name.log 2>&1 || { echo "$name build failed"; tail -5 /tmp/build_$name.log; continue; }
  timeout 150 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay $BENCH_ARGS 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:x['avg_ms'] for x in d['kernels']}; print('VARIANT $name', d['value'], 'fps', d['ms_per_step'], 'ms; raycast', k['raycast'], 'integrate', k.get('integrate'), 'integrate_bg', k.get('integrate_bg'), 'assoc', k.get('assoc'), 'track', k.get('track'))"
done
