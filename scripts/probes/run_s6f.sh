#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for A in 0 2; do
  echo "== EMF_TRACK_AHEAD=$A"
  EMF_TRACK_AHEAD=$A bash scripts/quick_trace.sh --track --no-target 2>&1 | tail -12
  python scripts/track_launch_stats.py
done
