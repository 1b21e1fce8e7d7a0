#!/bin/bash
# round 5: wave priority of the march against the sweep's waves on the same SIMDs
cd /root/repo
mkdir -p gpurun_out/r05p
sed -i 's/touch emfusion_amd\/csrc\/\*.hip/touch emfusion_amd\/csrc\/batched.hip/' scripts/sweep_variants.sh
BENCH_ARGS="--no-target" bash scripts/sweep_variants.sh "prio0:" "prio1:-DEMF_RAY_PRIO=1" "prio2:-DEMF_RAY_PRIO=2" "prio3:-DEMF_RAY_PRIO=3" "prio0:" "prio1:-DEMF_RAY_PRIO=1" "prio3:-DEMF_RAY_PRIO=3" 2>&1 | tee gpurun_out/r05p/ab.log
