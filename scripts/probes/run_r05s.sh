#!/bin/bash
# round 5: the driver's window (frames 5-24) -- is its raycast slower by itself or because of the heavier sweep beside it?
cd /root/repo
for envs in "A=1" "EMF_BG_OVERLAP=0"; do
env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target --all-kernel-events 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:(x['avg_ms'],x['launches']) for x in d['kernels']}; r=d['roofline']; print('WINDOW 5-24 $envs', d['value'], d['ms_per_step'], k, 'samples', r.get('march_samples_per_launch'))"
env $envs python bench.py --steps 40 --warmup 70 --no-cpu-baseline --no-target --all-kernel-events 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:(x['avg_ms'],x['launches']) for x in d['kernels']}; r=d['roofline']; print('WINDOW 70-110 $envs', d['value'], d['ms_per_step'], k, 'samples', r.get('march_samples_per_launch'))"
done
