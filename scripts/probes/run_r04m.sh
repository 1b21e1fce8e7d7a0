bash scripts/ab_transports.sh
python scripts/stream_history_probe.py --matrix3 2>&1 | grep -v amdgpu.ids
