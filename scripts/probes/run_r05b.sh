#!/bin/bash
# round 5: lanes-per-ray march with 1 / 2 / 4 rows, and the y-pair copy probe (two 16-byte gathers per sample)
cd /root/repo
mkdir -p gpurun_out/r05b
for rows in 2; do
EMF_MARCH_ROWS=$rows timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r05b/tests_rows$rows.log 2>&1
echo "rows $rows tests rc $?"; tail -2 gpurun_out/r05b/tests_rows$rows.log
done
EMF_PAIR_PROBE=1 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r05b/tests_pairs.log 2>&1
echo "pairs tests rc $?"; tail -2 gpurun_out/r05b/tests_pairs.log
bash scripts/ab_env.sh "rows1:EMF_MARCH_ROWS=1" "rows2:EMF_MARCH_ROWS=2" "rows4:EMF_MARCH_ROWS=4" "pairs:EMF_PAIR_PROBE=1" "rows1:EMF_MARCH_ROWS=1" "rows2:EMF_MARCH_ROWS=2" "pairs:EMF_PAIR_PROBE=1" 2>&1 | tee gpurun_out/r05b/ab.log
for rows in 1 2 4; do
EMF_MARCH_ROWS=$rows timeout 150 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-target 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ROWS $rows samples', r.get('march_samples_per_launch'), 'gathered', r.get('march_gathered_per_launch'))"
done 2>&1 | tee -a gpurun_out/r05b/ab.log
