#!/bin/bash
# round 5: why is the driver's first bench of a fresh box 12 % below the same command run second?
cd /root/repo
for i in 1 2 3; do
BENCH_PER_STEP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target 2> /tmp/err_$i.txt | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:(x['avg_ms'],x['launches']) for x in d['kernels']}; print('RUN $i', d['value'], d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'], k)"
grep PER_STEP /tmp/err_$i.txt | cut -c1-300
done
