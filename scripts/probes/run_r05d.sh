#!/bin/bash
# round 5: lanes-per-ray march after pinning the row bases in SGPRs (one round trip per iteration): A/B and wave timelines
cd /root/repo
mkdir -p gpurun_out/r05d
bash scripts/ab_env.sh "rows1:EMF_MARCH_ROWS=1" "rows2:EMF_MARCH_ROWS=2" "rows4:EMF_MARCH_ROWS=4" "rows1:EMF_MARCH_ROWS=1" "rows2:EMF_MARCH_ROWS=2" "rows4:EMF_MARCH_ROWS=4" 2>&1 | tee gpurun_out/r05d/ab.log
cp emfusion_amd/libemf_hip.so /tmp/libemf_hip.keep
for rows in 4 2; do
echo "== timeline rows $rows"
EMF_MARCH_ROWS=$rows bash scripts/run_trace.sh 2>&1 | tee -a gpurun_out/r05d/timeline_rows$rows.log
done
cp /tmp/libemf_hip.keep emfusion_amd/libemf_hip.so
