b() { echo "== bench [$*]"; env $ENVV python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-stats-replay "$@" 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
ENVV="A=1" b
ENVV="A=1" b --no-kernel-events
ENVV="A=1" b --event-stride 1
ENVV="EMF_BG_OVERLAP=0" b
ENVV="GPU_MAX_HW_QUEUES=8" b
ENVV="A=1" b --steps 20 --warmup 5
PROBE_TIMERS=1 python scripts/stream_history_probe.py fresh
PROBE_TIMERS=1 PROBE_TORCH_FIRST=1 python scripts/stream_history_probe.py fresh
python scripts/stream_history_probe.py fresh
