bash scripts/quick_trace.sh --force-sharded --comm peer > gpurun_out/r04n_trace_peer.log 2>&1; python scripts/frame_timeline.py >> gpurun_out/r04n_trace_peer.log 2>&1; cat gpurun_out/r04n_trace_peer.log | tail -60
python -m pytest tests/test_gpu_stream_history.py -x -q -m gpu 2>&1 | tail -5
