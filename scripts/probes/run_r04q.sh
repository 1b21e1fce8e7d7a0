bash scripts/profile_round.sh r04a > gpurun_out/r04a_profile.log 2>&1; tail -5 gpurun_out/r04a_profile.log
PROFILE_TIMEOUT=500 bash scripts/profile_round.sh r04_cfg4 --width 1280 --height 960 --bg-res 1024 --bg-voxel 0.005 --obj-res 256 --objects-per-gpu 2 > gpurun_out/r04_cfg4_profile.log 2>&1; tail -5 gpurun_out/r04_cfg4_profile.log
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r04q_bench.json 2> gpurun_out/r04q_bench.err; tail -3 gpurun_out/r04q_bench.err; head -c 3000 gpurun_out/r04q_bench.json
for i in 1 2; do python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --no-target --force-sharded --comm peer 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('peer (fenced wait)', d['value'], d['ms_per_step'])"; done
