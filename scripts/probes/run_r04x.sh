for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver protocol', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-target 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steady', d['value'], d['ms_per_step'])"
python -m pytest tests/test_gpu_dynamic_objects.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3
