#!/bin/bash
# Usage: sweep_env_variants.sh "ENV..." "NAME:FLAGS" ...  -- like sweep_variants.sh, with a fixed environment for
# every bench run (first argument, e.g. "EMF_BG_OVERLAP=0").
cd /root/repo
envs=$1; shift
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  touch emfusion_amd/csrc/*.hip
  make -s -C emfusion_amd/csrc -j8 EXTRA="$flags" >/tmp/build_$name.log 2>&1 || { echo "$name build failed"; tail -5 /tmp/build_$name.log; continue; }
  env $envs timeout 150 python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:x['avg_ms'] for x in d['kernels']}; print('VARIANT $name [$envs]', d['value'], 'fps', d['ms_per_step'], 'ms; raycast', k['raycast'], 'integrate', k.get('integrate'), 'integrate_bg', k.get('integrate_bg'))"
done
