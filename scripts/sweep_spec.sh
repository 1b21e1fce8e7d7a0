#!/bin/bash
# diagnostic: sweep the speculative batch size / register cap of the level-1 raycast kernel
cd $GRAFT_REPO_ROOT/emfusion_amd/csrc
cp raycast.hip /tmp/raycast.hip.orig
for K in 1 2 3 4; do for W in 0 5; do
  cp /tmp/raycast.hip.orig raycast.hip
  sed -i "s/constexpr int kSpecBatch = 4;/constexpr int kSpecBatch = $K;/" raycast.hip
  if [ $W -ne 0 ]; then sed -i "s/__global__ __launch_bounds__(64) void k_raycast(/__global__ __launch_bounds__(64, $W) void k_raycast(/" raycast.hip; fi
  make -s > /dev/null 2>&1
  v=$(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -c raycast.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "k_raycastILb1" | grep -E "VGPRs:|ScratchSize" | sed -E 's/.*(VGPRs: [0-9]+|ScratchSize \[bytes\/lane\]: [0-9]+).*/\1/' | tr '\n' ' ')
  t=$(cd ../.. && timeout 60 python scripts/raycast_probe.py 60 2>&1 | grep "bg raycast alone:" | sed 's/samples.*//')
  echo "K=$K bound=$W  $v | $t"
done; done
cp /tmp/raycast.hip.orig raycast.hip; make -s > /dev/null 2>&1
