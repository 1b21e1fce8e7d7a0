#!/bin/bash
# diagnostic: sweep the cooperative-mode threshold of the wave-scheduled march
cd $GRAFT_REPO_ROOT/emfusion_amd/csrc
for T in 0 4 12 24 48; do
  touch raycast.hip batched.hip
  make -s CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter -I../../include -I. -DEMF_COOP_MAX_RAYS=$T" > /dev/null 2>&1
  t=$(cd ../.. && timeout 60 python scripts/raycast_probe.py 60 2>&1 | grep -E "bg raycast alone:|obj [13] raycast" | sed 's/samples.*//; s/, with flags.*//' | tr '\n' ';')
  echo "T=$T | $t"
done
touch raycast.hip batched.hip; make -s > /dev/null 2>&1
