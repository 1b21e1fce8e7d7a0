r() { env $1 BENCH_PER_STEP=1 python bench.py --no-cpu-baseline --no-stats-replay --steps 60 --warmup 20 2>&1 | grep -E "PER_STEP" | cut -c1-60; }
echo warm; r A=1
echo notorch; r BENCH_NO_TORCH_SYNC=1; r BENCH_NO_TORCH_SYNC=1; r BENCH_NO_TORCH_SYNC=1
echo gcoff; r BENCH_GC_OFF=1; r BENCH_GC_OFF=1
echo plain; r A=1; r A=1
