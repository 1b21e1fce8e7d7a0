bash scripts/ab_transports.sh
CFG4="--width 1280 --height 960 --bg-res 1024 --bg-voxel 0.005 --obj-res 256 --objects-per-gpu 2"
for v in "A=1" "EMF_BG_OVERLAP=0" "A=1" "EMF_BG_OVERLAP=0"; do env $v python bench.py --steps 60 --warmup 30 --no-cpu-baseline --no-stats-replay --no-kernel-events --no-target $CFG4 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 share [$v]', d['value'], d['ms_per_step'])"; done
python -m pytest tests/test_gpu_exchange_latency.py tests/test_gpu_stream_history.py -x -q -m gpu -s 2>&1 | tail -25
touch emfusion_amd/csrc/tracking.hip; make -s -C emfusion_amd/csrc -j8 EXTRA=-DEMF_TRACK_TRACE=150 > /tmp/tt.log 2>&1 || tail -5 /tmp/tt.log
python scripts/track_step_trace.py 2>&1 | grep -v amdgpu.ids | tail -40
