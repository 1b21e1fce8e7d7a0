#!/bin/bash
# round 5: what pins the raycast at 0.39 ms alone in every march form?  no objects / far-bound scan of the objects; the full-size TUM test
cd /root/repo
mkdir -p gpurun_out/r05f
ab() { name=$1; shift; envs=$1; shift
  env $envs timeout 150 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --no-target "$@" 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['kind']:x['avg_ms'] for x in d['kernels']}; print('ENV $name', d['value'], 'fps', d['ms_per_step'], 'ms; raycast', k['raycast'], 'integrate', k.get('integrate'), 'integrate_bg', k.get('integrate_bg'))"; }
for rows in 1 2 4; do
ab "noobj-rows$rows-serial" "EMF_MARCH_ROWS=$rows EMF_BG_OVERLAP=0" --objects-per-gpu 0
ab "noobj-rows$rows" "EMF_MARCH_ROWS=$rows" --objects-per-gpu 0
done 2>&1 | tee gpurun_out/r05f/ab.log
for rows in 1 4; do
ab "farscan-rows$rows-serial" "EMF_MARCH_ROWS=$rows EMF_BG_OVERLAP=0 EMF_FAR_SCAN=1"
ab "farscan-rows$rows" "EMF_MARCH_ROWS=$rows EMF_FAR_SCAN=1"
done 2>&1 | tee -a gpurun_out/r05f/ab.log
timeout 1500 python -m pytest tests/test_gpu_tum_fullsize.py -q -m gpu -s > gpurun_out/r05f/tum.log 2>&1
echo "tum rc $?"; grep -v "^$" gpurun_out/r05f/tum.log | tail -30 | cut -c1-600
