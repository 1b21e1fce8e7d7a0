#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_tracking.py tests/test_gpu_tracking_pipeline.py tests/test_golden.py tests/test_abi_exports.py -x -q -m gpu 2>&1 | tail -5
REPS="3" bash scripts/ab_track_base.sh
