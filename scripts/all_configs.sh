cd /root/repo
run() { name=$1; shift; timeout 280 python bench.py --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:x['avg_ms'] for x in d.get('kernels',[])}; print('CFG $name', d['value'], 'fps', d['ms_per_step'], 'ms', k)"; }
run c0 --objects-per-gpu 0 --bg-res 256 --bg-voxel 0.02
run c1
run c1track --track --steps 60 --warmup 20
run c4share --objects-per-gpu 8
run c5share --width 1280 --height 960 --bg-res 1024 --bg-voxel 0.005 --obj-res 256 --objects-per-gpu 2 --steps 40 --warmup 15
