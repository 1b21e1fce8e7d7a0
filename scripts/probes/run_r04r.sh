CFG4="--width 1280 --height 960 --bg-res 1024 --bg-voxel 0.005 --obj-res 256 --objects-per-gpu 2"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline $CFG4 > gpurun_out/r04r_cfg4_bench.json 2>/dev/null
for v in "A=1" "EMF_BG_OVERLAP=0" "A=1" "EMF_BG_OVERLAP=0"; do env $v python bench.py --steps 60 --warmup 30 --no-cpu-baseline --no-stats-replay --no-kernel-events $CFG4 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 share [$v]', d['value'], d['ms_per_step'])"; done
(time python -m pytest tests -x -q -m gpu --durations=6) > gpurun_out/r04r_gpu_tests.log 2>&1; tail -14 gpurun_out/r04r_gpu_tests.log
