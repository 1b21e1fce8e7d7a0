#!/bin/bash
# round 5: rows 1 / 2 / 4 for the background with and without the background's integration beside the raycast; the full-size TUM test
cd /root/repo
mkdir -p gpurun_out/r05e
bash scripts/ab_env.sh "rows1:EMF_MARCH_ROWS=1" "rows2:EMF_MARCH_ROWS=2" "rows4:EMF_MARCH_ROWS=4" "rows1-serial:EMF_MARCH_ROWS=1 EMF_BG_OVERLAP=0" "rows2-serial:EMF_MARCH_ROWS=2 EMF_BG_OVERLAP=0" "rows4-serial:EMF_MARCH_ROWS=4 EMF_BG_OVERLAP=0" "rows1:EMF_MARCH_ROWS=1" "rows4:EMF_MARCH_ROWS=4" "rows4-serial:EMF_MARCH_ROWS=4 EMF_BG_OVERLAP=0" 2>&1 | tee gpurun_out/r05e/ab.log
timeout 1500 python -m pytest tests/test_gpu_tum_fullsize.py -x -q -m gpu -s > gpurun_out/r05e/tum.log 2>&1
echo "tum rc $?"; tail -30 gpurun_out/r05e/tum.log | cut -c1-400
