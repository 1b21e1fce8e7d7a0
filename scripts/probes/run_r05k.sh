#!/bin/bash
# round 5: whole GPU suite on the pruned tree; counter profile of the headline workload (r05a) and of the 8-object target (r05_cfg3)
cd /root/repo
mkdir -p gpurun_out/r05k
timeout 3000 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/r05k/tests.log 2>&1
echo "tests rc $?"; grep -v "^$" gpurun_out/r05k/tests.log | tail -25 | cut -c1-300
bash scripts/profile_round.sh r05a > gpurun_out/r05k/profile_r05a.log 2>&1; tail -3 gpurun_out/r05k/profile_r05a.log
bash scripts/profile_round.sh r05_cfg3 --objects-per-gpu 8 > gpurun_out/r05k/profile_cfg3.log 2>&1; tail -3 gpurun_out/r05k/profile_cfg3.log
