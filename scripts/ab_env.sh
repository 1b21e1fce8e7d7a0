#!/bin/bash
# Usage: ab_env.sh "NAME:VAR=VALUE ..." ...  -- same build, bench under different environments.
cd /root/repo
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 150 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --no-target --no-strong --no-entry 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:x['avg_ms'] for x in d['kernels']}; print('ENV $name', d['value'], 'fps', d['ms_per_step'], 'ms; raycast', k['raycast'], 'integrate', k.get('integrate'), 'integrate_bg', k.get('integrate_bg'), 'assoc', k.get('assoc'))"
done
