#!/bin/bash
# Round 6: when does the out-of-place background sweep (second copy of the volume, integration on `aux` beside the raycast)
# pay?  frames/s with the overlap on / off and with the 1/lambda table on / off over background sizes and image sizes
# (the volume always spans 5.12 m; objects as in the configuration the size belongs to).  DESIGN.md 5.1b.
cd /root/repo
mkdir -p gpurun_out/r06_rule
run() { name=$1; envs=$2; shift 2
  env $envs timeout 200 python bench.py --no-cpu-baseline --no-stats-replay --no-target --no-strong --no-entry "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:x['avg_ms'] for x in d.get('kernels',[])}; print('RULE $name [$envs]', d['value'], 'fps', d['ms_per_step'], 'ms', k)"; }
for size in "640 480 4 128" "1280 960 2 256"; do
  set -- $size; W=$1; H=$2; NOBJ=$3; ORES=$4
  for BG in 512 768 1024; do
    VOX=$(python -c "print(5.12/$BG)")
    for envs in "EMF_X=0" "EMF_BG_OVERLAP=0" "EMF_LAMBDA_TABLE=0" "EMF_BG_OVERLAP=0 EMF_LAMBDA_TABLE=0"; do
      run "${W}x${H}_bg${BG}" "$envs" --width $W --height $H --bg-res $BG --bg-voxel $VOX --obj-res $ORES --objects-per-gpu $NOBJ --steps 40 --warmup 20
    done
  done
done 2>&1 | tee gpurun_out/r06_rule/rule.log
