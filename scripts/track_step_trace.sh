#!/bin/bash
# GPU box: phase stamps of k_track_step for several workgroups, 640x480; MODELS="1 1 1 1": an object stage of four
cd ${GRAFT_REPO_ROOT:-/root/repo}
for WG in ${WGS:-0 100 200}; do
  touch emfusion_amd/csrc/tracking.hip
  make -s -C emfusion_amd/csrc -j8 EXTRA="-DEMF_TRACK_TRACE=$WG $TRACK_EXTRA" > /tmp/tb.log 2>&1 || { tail -5 /tmp/tb.log; exit 1; }
  echo "== EMF_TRACK_TRACE=$WG models ${MODELS:-0}"
  python scripts/track_step_trace.py ${MODELS:-0} 2>&1 | tail -${TAIL:-16}
done
touch emfusion_amd/csrc/tracking.hip
make -s -C emfusion_amd/csrc -j8 > /tmp/tb.log 2>&1
