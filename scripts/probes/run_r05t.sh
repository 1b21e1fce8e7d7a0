#!/bin/bash
cd /root/repo
sed -i 's/touch emfusion_amd\/csrc\/\*.hip/touch emfusion_amd\/csrc\/batched.hip/' scripts/sweep_variants.sh
BENCH_ARGS="--no-target --all-kernel-events" bash scripts/sweep_variants.sh "cull:" "nocull:-DEMF_OBJ_NO_CULL" "cull:" "nocull:-DEMF_OBJ_NO_CULL" 2>&1 | grep VARIANT
BENCH_ARGS="--no-target --all-kernel-events --objects-per-gpu 8" bash scripts/sweep_variants.sh "cull8:" "nocull8:-DEMF_OBJ_NO_CULL" 2>&1 | grep VARIANT
