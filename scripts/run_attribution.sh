#!/bin/bash
# GPU box: build the attribution variant of the raycast in place and run scripts/raycast_attribution.py
cd ${GRAFT_REPO_ROOT:-/root/repo}
touch emfusion_amd/csrc/batched.hip
make -s -C emfusion_amd/csrc -j8 EXTRA="-DEMF_RAY_TRACE -DEMF_MARCH_STAMP" > /tmp/attr_build.log 2>&1 || { tail -5 /tmp/attr_build.log; exit 1; }
timeout 300 python scripts/raycast_attribution.py ${FRAMES:-40} 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/raycast_attribution.py ${FRAMES:-40} --no-bg-overlap 2>&1 | grep -v amdgpu.ids
