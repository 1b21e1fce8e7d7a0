#!/bin/bash
# round 5 final checks: smoke(), the driver's bench command with its wall time, counter profiles of the final kernels
cd /root/repo
mkdir -p gpurun_out/r05q
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -5
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05q/bench_driver.json 2> gpurun_out/r05q/bench_driver.err ) 2>&1 | tail -4
python -c "
import json; d=json.loads(open('gpurun_out/r05q/bench_driver.json').read().strip().splitlines()[-1]); r=d['roofline']
print('DRIVER', d['value'], d['ms_per_step'], 'roofline', r['bound'], r['frac'], r['achieved'], r['unit'], 'traffic', r['traffic'], 'from', r['counters_from'][:40])
print('target', d['target_config']['value'], 'steady', d['steady_state']['value'], 'cpu', d['cpu_baseline'])"
bash scripts/profile_round.sh r05b > gpurun_out/r05q/profile.log 2>&1; tail -2 gpurun_out/r05q/profile.log
