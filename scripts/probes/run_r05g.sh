#!/bin/bash
# round 5: the two closed-loop tracking tests; counter profile of the tracked workload
cd /root/repo
mkdir -p gpurun_out/r05g
timeout 1500 python -m pytest tests/test_gpu_tracking_divergence.py tests/test_gpu_tum_fullsize.py -q -m gpu -s > gpurun_out/r05g/tests.log 2>&1
echo "tests rc $?"; grep -v "^$" gpurun_out/r05g/tests.log | tail -25 | cut -c1-900
STEPS=20 WARMUP=5 bash scripts/profile_round.sh r05_track --track > gpurun_out/r05g/profile.log 2>&1
tail -5 gpurun_out/r05g/profile.log
