mkdir -p gpurun_out
python -m pytest tests/test_gpu_config3_rehearsal.py -x -q -m gpu > gpurun_out/r04c_cfg3.log 2>&1; tail -15 gpurun_out/r04c_cfg3.log
python scripts/stream_history_probe.py --matrix2 > gpurun_out/r04c_matrix2.log 2>&1; cat gpurun_out/r04c_matrix2.log
for v in "" "EMF_PRIO_MAIN=high" "EMF_PRIO_MAIN=high EMF_PRIO_LISTS=high" "EMF_PRIO_LISTS=low" "EMF_PRIO_AUX=normal"; do
  echo "== bench plain [$v]"; env $v python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-stats-replay 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
PROBE_TORCH_FIRST=1 python scripts/stream_history_probe.py fresh; PROBE_TORCH_FIRST=1 python scripts/stream_history_probe.py foreign1; PROBE_TORCH_FIRST=1 python scripts/stream_history_probe.py foreign2
