#!/bin/bash
cd $GRAFT_REPO_ROOT/emfusion_amd/csrc
touch raycast.hip batched.hip
make -s CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-parameter -I../../include -I. -DEMF_COOP_MAX_RAYS=12 -DEMF_COOP_DEBUG_CALLS" > /dev/null 2>&1
cd ../.. && timeout 60 python scripts/raycast_probe.py 60 2>&1 | grep -E "stats|raycast alone" | head -12
cd emfusion_amd/csrc; touch raycast.hip batched.hip; make -s > /dev/null 2>&1
