for i in 1 2 3; do BENCH_PER_STEP=1 python bench.py --no-cpu-baseline --no-stats-replay --steps 60 --warmup 20 2>&1 | grep -E "PER_STEP|^\{" | cut -c1-700; done
echo "--- top during a run"; (BENCH_PER_STEP=1 python bench.py --no-cpu-baseline --no-stats-replay --steps 60 --warmup 20 > /dev/null 2>&1 &) ; sleep 6; top -b -n 1 | head -15; wait; ps aux --sort=-%cpu | head -8
