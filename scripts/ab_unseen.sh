cd /root/repo
run() { echo -n "$1: "; env $2 python bench.py --no-cpu-baseline $3 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kind']:x['avg_ms'] for x in d['kernels']}; print(d['value'], 'fps', d['ms_per_step'], k)"; }
C5="--width 1280 --height 960 --bg-res 1024 --bg-voxel 0.005 --obj-res 256 --objects-per-gpu 2 --steps 40 --warmup 15"
run c5_off EMF_UNSEEN_TILES=0 "$C5"
run c5_on EMF_UNSEEN_TILES=1 "$C5"
run off20 EMF_UNSEEN_TILES=0 "--steps 20 --warmup 5"
run on20 EMF_UNSEEN_TILES=1 "--steps 20 --warmup 5"
run off20 EMF_UNSEEN_TILES=0 "--steps 20 --warmup 5"
run on20 EMF_UNSEEN_TILES=1 "--steps 20 --warmup 5"
run c0_off EMF_UNSEEN_TILES=0 "--objects-per-gpu 0 --bg-res 256 --bg-voxel 0.02"
run c0_on EMF_UNSEEN_TILES=1 "--objects-per-gpu 0 --bg-res 256 --bg-voxel 0.02"
run trk_off EMF_UNSEEN_TILES=0 "--track --steps 60 --warmup 20"
run trk_on EMF_UNSEEN_TILES=1 "--track --steps 60 --warmup 20"
