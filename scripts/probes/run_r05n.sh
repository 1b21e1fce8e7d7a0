#!/bin/bash
# round 5: XCD-banded E-step grid against the plain one (all kernel events: `assoc` timed in every frame), then parity
cd /root/repo
mkdir -p gpurun_out/r05n
sed -i 's/touch emfusion_amd\/csrc\/\*.hip/touch emfusion_amd\/csrc\/batched.hip/' scripts/sweep_variants.sh
BENCH_ARGS="--no-target --all-kernel-events" bash scripts/sweep_variants.sh "plain:-DEMF_ESTEP_PLAIN_GRID" "banded:" "plain:-DEMF_ESTEP_PLAIN_GRID" "banded:" 2>&1 | tee gpurun_out/r05n/ab.log
BENCH_ARGS="--no-target" bash scripts/sweep_variants.sh "plain:-DEMF_ESTEP_PLAIN_GRID" "banded:" "plain:-DEMF_ESTEP_PLAIN_GRID" "banded:" 2>&1 | tee -a gpurun_out/r05n/ab.log
git checkout scripts/sweep_variants.sh 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_golden.py -q -m gpu 2>&1 | tail -3
