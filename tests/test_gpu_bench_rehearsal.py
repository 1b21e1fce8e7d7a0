"""bench.py as the driver launches it for N > 1 -- one process per rank under torch.distributed.run --
rehearsed with 2 ranks on ONE GPU: `--comm gloo` swaps RCCL for the host-staged communicator, everything
else (rank / world handling, object ownership, depth broadcast, band split, barrier + max-over-ranks
timing, ONE JSON line from rank 0) is the code of the real multi-GPU run."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_bench_runs_with_several_ranks(dev, world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(ROOT / "bench.py"),
           "--gpus", str(world), "--steps", "6", "--warmup", "3", "--bg-res", "128", "--bg-voxel", "0.04",
           "--obj-res", "32", "--objects-per-gpu", "2", "--width", "320", "--height", "240",
           "--no-cpu-baseline", "--comm", "gloo"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 6 and d["warmup"] == 3 and d["value"] > 0
    assert d["config"]["objects_total"] == 2 * world and d["config"]["background"] == "replicated"
    assert "broadcast(depth)" in d["config"]["collectives_per_frame"]
    assert d["scaling"] == "weak" and "REHEARSAL" in d["data"]
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None or d["n_gpus"] == 1
