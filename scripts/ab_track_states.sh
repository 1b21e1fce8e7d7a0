cd /root/repo
for v in "old:-DEMF_TRACK_XPOSE=0" "new:-DEMF_TRACK_XPOSE=1"; do
  name=${v%%:*}; flags=${v#*:}
  touch emfusion_amd/csrc/tracking.hip
  make -s -C emfusion_amd/csrc -j8 EXTRA="$flags" >/tmp/build_$name.log 2>&1 || { echo "$name build failed"; tail -5 /tmp/build_$name.log; }
  python scripts/dump_track_states.py $name 2>&1 | tail -2
done
python -c "
import numpy as np
a=np.load('gpurun_out/old.npy'); b=np.load('gpurun_out/new.npy'); print('states identical:', np.array_equal(a,b), (a!=b).sum())"
