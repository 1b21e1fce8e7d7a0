#!/bin/bash
cd /root/repo
touch emfusion_amd/csrc/batched.hip
make -s -C emfusion_amd/csrc -j8 EXTRA="-DEMF_RAY_TRACE $1" > /tmp/trace_build.log 2>&1 || { tail -5 /tmp/trace_build.log; exit 1; }
timeout 300 python scripts/raycast_timeline.py ${FRAMES:-40} 2>&1 | grep -v amdgpu.ids
