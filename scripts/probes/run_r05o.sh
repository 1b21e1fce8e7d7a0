#!/bin/bash
# round 5: the raycast's registers: the hit's gradient one component at a time (86 -> 71 VGPRs: 5 -> 7 waves per SIMD), and caps
cd /root/repo
mkdir -p gpurun_out/r05o
sed -i 's/touch emfusion_amd\/csrc\/\*.hip/touch emfusion_amd\/csrc\/batched.hip emfusion_amd\/csrc\/raycast.hip/' scripts/sweep_variants.sh
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_golden.py -q -m gpu 2>&1 | tail -2
BENCH_ARGS="--no-target" bash scripts/sweep_variants.sh "full86:-DEMF_GRAD_FULL" "lean71:" "lean-wpe8:-DEMF_RAY_WPE=8" "lean-wpe6:-DEMF_RAY_WPE=6" "full86:-DEMF_GRAD_FULL" "lean71:" "lean-wpe8:-DEMF_RAY_WPE=8" 2>&1 | tee gpurun_out/r05o/ab.log
for v in "full86:-DEMF_GRAD_FULL" "lean71:" "lean-wpe8:-DEMF_RAY_WPE=8"; do
  name=${v%%:*}; flags=${v#*:}
  touch emfusion_amd/csrc/batched.hip; make -s -C emfusion_amd/csrc -j8 EXTRA="$flags" >/dev/null 2>&1
  EMF_BG_OVERLAP=0 timeout 150 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --no-target 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['kind']:x['avg_ms'] for x in d['kernels']}; print('SERIAL $name', d['value'], 'fps', d['ms_per_step'], 'ms; raycast', k['raycast'], 'integrate', k.get('integrate'))"
done 2>&1 | tee -a gpurun_out/r05o/ab.log
git checkout scripts/sweep_variants.sh 2>/dev/null
