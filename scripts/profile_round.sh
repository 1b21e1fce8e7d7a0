#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + two PMC passes of bench.py.
# Outputs land under gpurun_out/prof_$1/ ; summarise with scripts/summarize_profile.py.
# Every profiler call has its own timeout (a hung PMC pass must not eat the GPU budget).
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 60 --warmup 30 --no-cpu-baseline"
PMCBENCH="python bench.py --steps 20 --warmup 30 --no-cpu-baseline --no-kernel-events --no-stats-replay"
timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $BENCH > $out/trace.log 2>&1; echo "trace rc=$?"
# HBM-side read requests by size (4 TCC slots), then write requests: separate passes, no trace domains
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B --kernel-trace -d $out/pmc_rd -o p -- $PMCBENCH > $out/pmc_rd.log 2>&1; echo "pmc_rd rc=$?"
timeout 200 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_HIT TCC_MISS --kernel-trace -d $out/pmc_wr -o p -- $PMCBENCH > $out/pmc_wr.log 2>&1; echo "pmc_wr rc=$?"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o p -- $PMCBENCH > $out/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o p -- $PMCBENCH > $out/pmc_write.log 2>&1; echo "pmc_write rc=$?"
grep -h '"metric"' $out/trace.log | tail -1 > $out/bench_under_trace.json
mkdir -p gpurun_out/summary
python scripts/summarize_profile.py $tag gpurun_out/summary > $out/summary.log 2>&1; tail -3 $out/summary.log
# the raw databases are tens of MB (per-dispatch, per-XCC rows): keep only the kernel trace
rm -rf $out/pmc_rd $out/pmc_wr $out/pmc_fetch $out/pmc_write
ls -la gpurun_out/summary
