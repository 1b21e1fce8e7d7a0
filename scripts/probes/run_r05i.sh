#!/bin/bash
# round 5: the LDS pixel-window sweep (integrate_tile<OUT, WIN>): parity with the window forced on, A/B on configs[1] and on the
# configs[4] share (libemf_hip.so built with -DEMF_DEBUG_SWITCHES: EMF_INT_WINDOW); frame timeline of the headline
cd /root/repo
mkdir -p gpurun_out/r05i
EMF_INT_WINDOW=1 timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r05i/tests_window.log 2>&1
echo "window tests rc $?"; tail -3 gpurun_out/r05i/tests_window.log
ab() { name=$1; shift; envs=$1; shift
  env $envs timeout 280 python bench.py --no-cpu-baseline --no-stats-replay --no-target "$@" 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['kind']:x['avg_ms'] for x in d['kernels']}; print('ENV $name', d['value'], 'fps', d['ms_per_step'], 'ms; raycast', k['raycast'], 'integrate', k.get('integrate'), 'integrate_bg', k.get('integrate_bg'))"; }
C5="--width 1280 --height 960 --bg-res 1024 --bg-voxel 0.005 --obj-res 256 --objects-per-gpu 2 --steps 40 --warmup 15"
for rep in 1 2; do
ab "cfg4-gather" "EMF_INT_WINDOW=0" $C5
ab "cfg4-window" "EMF_INT_WINDOW=1" $C5
ab "cfg4-gather-serial" "EMF_INT_WINDOW=0 EMF_BG_OVERLAP=0" $C5
ab "cfg4-window-serial" "EMF_INT_WINDOW=1 EMF_BG_OVERLAP=0" $C5
ab "cfg1-gather" "EMF_INT_WINDOW=0" --steps 100 --warmup 30
ab "cfg1-window" "EMF_INT_WINDOW=1" --steps 100 --warmup 30
done 2>&1 | tee gpurun_out/r05i/ab.log
bash scripts/quick_trace.sh > gpurun_out/r05i/quick_trace.log 2>&1; python scripts/frame_timeline.py > gpurun_out/r05i/frame_timeline.log 2>&1; tail -45 gpurun_out/r05i/frame_timeline.log
rm -rf gpurun_out/quick_trace/*.db gpurun_out/quick_trace/*/*.db
