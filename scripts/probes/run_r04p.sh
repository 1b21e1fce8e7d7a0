bash scripts/quick_trace.sh --force-sharded --comm peer 2>&1 | grep -E "k_peer|k_composite|k_pack|k_vis|k_estep|fps"
for i in 1 2; do python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --force-sharded --comm peer 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('peer fused', d['value'], d['ms_per_step'])"; done
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_peer_exchange.py -x -q -m gpu 2>&1 | tail -3
