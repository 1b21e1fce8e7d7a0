cd /root/repo
run() { echo -n "$1: "; env $2 python bench.py --no-cpu-baseline --no-stats-replay $3 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:x['avg_ms'] for x in d['kernels']}; print(d['value'], 'fps', d['ms_per_step'], k)"; }
C5="--width 1280 --height 960 --bg-res 1024 --bg-voxel 0.005 --obj-res 256 --objects-per-gpu 2 --steps 40 --warmup 15"
run c4_scan EMF_FAR_SCAN=1 "--objects-per-gpu 8"
run c4_noscan EMF_FAR_SCAN=0 "--objects-per-gpu 8"
run c5_scan EMF_FAR_SCAN=1 "$C5"
run c5_noscan EMF_FAR_SCAN=0 "$C5"
run off20_scan EMF_FAR_SCAN=1 "--steps 20 --warmup 5"
run off20_noscan EMF_FAR_SCAN=0 "--steps 20 --warmup 5"
run trk_scan EMF_FAR_SCAN=1 "--track --steps 60 --warmup 20"
run trk_noscan EMF_FAR_SCAN=0 "--track --steps 60 --warmup 20"
