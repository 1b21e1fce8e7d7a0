#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
P=emfusion_amd
mkdir -p /tmp/ab_new && cp $P/libemf_hip.so $P/libemf_fusion.so /tmp/ab_new/
for V in base tree; do
  if [ $V = base ]; then cp build_tmp/base/$P/*.so $P/; else cp /tmp/ab_new/*.so $P/; fi
  echo "== $V"
  bash scripts/quick_trace.sh --track --no-target 2>&1 | tail -1
  python scripts/track_launch_stats.py
done
cp /tmp/ab_new/*.so $P/
