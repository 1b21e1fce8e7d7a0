#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=emfusion_amd
t0=$(date +%s)
run() {
  python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --no-target --track 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = [x for x in d['kernels'] if x['kind'] == 'track']
print('$1: %.1f frames/s  %.4f ms/frame%s' % (d['value'], d['ms_per_step'], '  stage %.4f ms' % k[0]['avg_ms'] if k else ''), ' t=%d s' % ($(date +%s) - $t0))"
}
mkdir -p /tmp/ab_new /tmp/ab_v1
cp $P/libemf_hip.so $P/libemf_fusion.so /tmp/ab_new/
touch $P/csrc/tracking.hip; make -s -C $P/csrc -j8 EXTRA="-DEMF_TRACK_NO_AHEAD_BODY" > /tmp/tb.log 2>&1 || { tail /tmp/tb.log; exit 1; }
cp $P/libemf_hip.so $P/libemf_fusion.so /tmp/ab_v1/
echo "built t=$(( $(date +%s) - t0 )) s"
for rep in 1 2; do
  cp build_tmp/base/$P/*.so $P/; run "base            "
  cp /tmp/ab_v1/*.so $P/; run "new, one body   "
  cp /tmp/ab_new/*.so $P/; EMF_TRACK_AHEAD=0 run "new, two, ahead=0"
done
cp /tmp/ab_new/*.so $P/
