#!/bin/bash
# round 5: where the one-rank sharded frame's extra time is (timelines of the plain, the RCCL and the peer-write frame)
cd /root/repo
mkdir -p gpurun_out/r05m
bash scripts/ab_transports.sh 2>&1 | tee gpurun_out/r05m/ab.log
for mode in "" "--force-sharded --comm peer" "--force-sharded --comm rccl"; do
  echo "== $mode"
  bash scripts/quick_trace.sh --no-target $mode > gpurun_out/r05m/qt.log 2>&1; tail -1 gpurun_out/r05m/qt.log
  python scripts/frame_timeline.py 2>&1 | tail -42
  rm -rf gpurun_out/quick_trace/*.db gpurun_out/quick_trace/*/*.db
done > gpurun_out/r05m/timelines.log 2>&1
