b() { echo "== bench [$*]"; env $ENVV python bench.py --no-cpu-baseline --no-stats-replay "$@" 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'issue', d['host_issue_ms_per_step'])"; }
for i in 1 2 3 4; do ENVV="A=1" b --steps 60 --warmup 20; done
for i in 1 2 3; do ENVV="A=1" b --steps 20 --warmup 5; done
for i in 1 2; do ENVV="A=1" b --steps 100 --warmup 30; done
