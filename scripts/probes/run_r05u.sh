#!/bin/bash
# round 5: the raycast in front of the E-steps when both poses are supplied (EMF_EARLY_RAYCAST, debug-switch build of the host
# library for the A/B); parity suites with the new default
cd /root/repo
mkdir -p gpurun_out/r05u
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_long_sequence.py tests/test_gpu_switch_pairs.py tests/test_gpu_config_shares.py tests/test_gpu_dynamic_objects.py tests/test_gpu_lifecycle.py tests/test_gpu_cpp_app.py -q -m gpu 2>&1 | tail -4 | cut -c1-300
touch emfusion_amd/csrc/core/*.cpp; make -s -C emfusion_amd/csrc -j8 EXTRA_HOST=-DEMF_DEBUG_SWITCHES > /tmp/b.log 2>&1 || tail -5 /tmp/b.log
bash scripts/ab_env.sh "reference-order:EMF_EARLY_RAYCAST=0" "early-raycast:EMF_EARLY_RAYCAST=1" "reference-order:EMF_EARLY_RAYCAST=0" "early-raycast:EMF_EARLY_RAYCAST=1" 2>&1 | tee gpurun_out/r05u/ab.log
for e in 0 1; do
EMF_EARLY_RAYCAST=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('DRIVER-WINDOW early=$e', d['value'], d['ms_per_step'], 'target', d['target_config']['value'], 'steady', d['steady_state']['value'])"
done 2>&1 | tee -a gpurun_out/r05u/ab.log
EMF_EARLY_RAYCAST=1 bash scripts/quick_trace.sh --no-target > /dev/null 2>&1; python scripts/frame_timeline.py 2>&1 | tail -40 > gpurun_out/r05u/timeline_early.log
rm -rf gpurun_out/quick_trace/*.db gpurun_out/quick_trace/*/*.db
