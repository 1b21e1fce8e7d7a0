#!/bin/bash
# round 5, first measurement of the four-lanes-per-ray march: parity tests, the probe, the bench A/B
cd /root/repo
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity.py tests/test_gpu_fast_paths.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r05a/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r05a/tests.log
tail -5 gpurun_out/r05a/tests.log
timeout 300 ./build_tmp/march_probe > gpurun_out/r05a/probe.log 2>&1
grep -v "^VALU" gpurun_out/r05a/probe.log | grep "rcp\|quad"
bash scripts/ab_env.sh "lane:EMF_MARCH_ROWS=1" "quad:EMF_MARCH_ROWS=4" "lane:EMF_MARCH_ROWS=1" "quad:EMF_MARCH_ROWS=4" 2>&1 | tee gpurun_out/r05a/ab.log
