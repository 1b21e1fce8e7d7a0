for i in 1 2 3 4; do BENCH_PER_STEP=1 python bench.py --no-cpu-baseline --no-stats-replay --steps 60 --warmup 20 2>&1 | grep -E "PER_STEP|^\{" | cut -c1-120; done
