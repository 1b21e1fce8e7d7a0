#!/bin/bash
# GPU box: tracked bench of configs[1] with the libraries of build_tmp/base (a build of another commit: `git archive
# <commit> emfusion_amd include | tar -x -C build_tmp/base`, make there) against the tree's, interleaved in one
# process sequence on one box -- frame times of different boxes differ by 3 %.
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=emfusion_amd
mkdir -p /tmp/ab_new && cp $P/libemf_hip.so $P/libemf_fusion.so /tmp/ab_new/
run() {
  python bench.py --steps ${STEPS:-100} --warmup 30 --no-cpu-baseline --no-stats-replay --no-target ${BENCH_ARGS:---track} 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = [x for x in d['kernels'] if x['kind'] == 'track']
print('$1: %.1f frames/s  %.4f ms/frame%s' % (d['value'], d['ms_per_step'], '  stage %.4f ms' % k[0]['avg_ms'] if k else ''))"
}
for rep in 1 2 ${REPS}; do
  cp build_tmp/base/$P/*.so $P/; run "base       "
  cp /tmp/ab_new/*.so $P/; run "tree       "
done
cp /tmp/ab_new/*.so $P/
