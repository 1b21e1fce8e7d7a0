#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + two PMC passes of bench.py.
# Outputs land under gpurun_out/prof_$1/ ; summarise with scripts/summarize_profile.py.
# Every profiler call has its own timeout (a hung PMC pass must not eat the GPU budget).
# Usage: profile_round.sh TAG [bench.py workload arguments, e.g. --width 1280 --height 960 --bg-res 1024 ...]
tag=${1:-r01}
shift
ARGS="$*"
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the driver's protocol (python bench.py --steps 20 --warmup 5): the summaries average over the TIMED launches only
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-5}
export EMF_PROFILE_STEPS=$STEPS EMF_PROFILE_WARMUP=$WARMUP
BENCH="python bench.py --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-target --no-strong --no-entry $ARGS"
PMCBENCH="python bench.py --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-kernel-events --no-stats-replay --no-target --no-strong --no-entry $ARGS"
export EMF_PROFILE_ARGS="$ARGS"
T=${PROFILE_TIMEOUT:-200}
timeout $T rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $BENCH > $out/trace.log 2>&1; echo "trace rc=$?"
# HBM-side read requests by size (4 TCC slots), then write requests: separate passes, no trace domains
timeout $T rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B --kernel-trace -d $out/pmc_rd -o p -- $PMCBENCH > $out/pmc_rd.log 2>&1; echo "pmc_rd rc=$?"
timeout $T rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_HIT TCC_MISS --kernel-trace -d $out/pmc_wr -o p -- $PMCBENCH > $out/pmc_wr.log 2>&1; echo "pmc_wr rc=$?"
timeout $T rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o p -- $PMCBENCH > $out/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
timeout $T rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o p -- $PMCBENCH > $out/pmc_write.log 2>&1; echo "pmc_write rc=$?"
# what binds the kernels (bench.py roofline): issue, L1, L2 request counters -- own passes, no trace domains, no TA_* (they hang)
timeout $T rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace -d $out/pmc_sq -o p -- $PMCBENCH > $out/pmc_sq.log 2>&1; echo "pmc_sq rc=$?"
timeout $T rocprofv3 --pmc SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_sq2 -o p -- $PMCBENCH > $out/pmc_sq2.log 2>&1; echo "pmc_sq2 rc=$?"
timeout $T rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --kernel-trace -d $out/pmc_tcp -o p -- $PMCBENCH > $out/pmc_tcp.log 2>&1; echo "pmc_tcp rc=$?"
timeout $T rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_BUSY_sum --kernel-trace -d $out/pmc_tcc -o p -- $PMCBENCH > $out/pmc_tcc.log 2>&1; echo "pmc_tcc rc=$?"
grep -h '"metric"' $out/trace.log | tail -1 > $out/bench_under_trace.json
mkdir -p gpurun_out/summary
python scripts/summarize_profile.py $tag gpurun_out/summary > $out/summary.log 2>&1; tail -3 $out/summary.log
# the raw databases are tens of MB (per-dispatch, per-XCC rows): keep only the kernel trace
rm -rf $out/pmc_rd $out/pmc_wr $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_sq2 $out/pmc_tcp $out/pmc_tcc
ls -la gpurun_out/summary
