#!/bin/bash
# round 5: the objects' integration (k_integrate_batched, 62-72 us on the frame's critical stream for 4096 tiles): 1 / 2 / 4 / 8
# tiles per workgroup; all kernel events on, so that `integrate` is timed in every frame
cd /root/repo
mkdir -p gpurun_out/r05j
sed -i 's/touch emfusion_amd\/csrc\/\*.hip/touch emfusion_amd\/csrc\/batched.hip/' scripts/sweep_variants.sh
BENCH_ARGS="--no-target --all-kernel-events" bash scripts/sweep_variants.sh "per1:-DEMF_OBJ_TILES_PER_WG=1" "per2:-DEMF_OBJ_TILES_PER_WG=2" "per4:-DEMF_OBJ_TILES_PER_WG=4" "per8:-DEMF_OBJ_TILES_PER_WG=8" "per1:-DEMF_OBJ_TILES_PER_WG=1" "per4:-DEMF_OBJ_TILES_PER_WG=4" 2>&1 | tee gpurun_out/r05j/ab.log
BENCH_ARGS="--no-target" bash scripts/sweep_variants.sh "per1:-DEMF_OBJ_TILES_PER_WG=1" "per4:-DEMF_OBJ_TILES_PER_WG=4" "per1:-DEMF_OBJ_TILES_PER_WG=1" "per4:-DEMF_OBJ_TILES_PER_WG=4" 2>&1 | tee -a gpurun_out/r05j/ab.log
git checkout scripts/sweep_variants.sh 2>/dev/null
