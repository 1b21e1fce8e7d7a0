python scripts/stream_history_probe.py --matrix2 2>&1 | grep -v amdgpu.ids
for sc in foreign2 foreign5 foreign6; do PROBE_KEEP_GC=1 python scripts/stream_history_probe.py $sc 2>&1 | grep PROBE_RESULT | sed 's/^/gc-on /'; done
bash scripts/ab_transports.sh
EMF_PEER_FUSED=0 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-stats-replay --force-sharded --comm peer 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('peer unfused', d['value'], d['ms_per_step'])"
