#!/bin/bash
cd /root/repo
for ts in 0.5 0.25; do
EMF_TEST_TIME_SCALE=$ts timeout 600 python -m pytest tests/test_gpu_tracking_divergence.py -q -m gpu -s 2>&1 | grep "^\[{\|passed\|failed" | cut -c1-3000
cp gpurun_out/tracking_divergence_report.json gpurun_out/tracking_divergence_ts$ts.json
done
