#!/bin/bash
# round 5: the whole GPU suite after the switchboard was pruned; tracked and headline bench lines
cd /root/repo
mkdir -p gpurun_out/r05h
timeout 3000 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r05h/tests.log 2>&1
echo "tests rc $?"; grep -v "^$" gpurun_out/r05h/tests.log | tail -30 | cut -c1-300
python bench.py --steps 100 --warmup 30 --no-cpu-baseline --track 2>/dev/null | tail -1 > gpurun_out/r05h/bench_track.json
python -c "
import json; d=json.load(open('gpurun_out/r05h/bench_track.json')); r=d['roofline']; print('TRACK', d['value'], d['ms_per_step'], r.get('bound'), r.get('frac'), r.get('achieved'), r.get('unit'))"
python bench.py 2>/dev/null | tail -1 > gpurun_out/r05h/bench_default.json
python -c "
import json; d=json.load(open('gpurun_out/r05h/bench_default.json')); print('HEADLINE', d['value'], d['ms_per_step'], 'target', d['target_config']['value'], 'steady', d['steady_state'])"
