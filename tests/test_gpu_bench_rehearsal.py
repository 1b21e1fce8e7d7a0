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


# objects of the strong-scaling sub-run: with 2 ranks 64 (configs[3]'s count: 32 objects + the background = 33 table slots per
# rank, i.e. the SHARDED path in two chunks of the model table), with 3 ranks 12
STRONG = {2: 64, 3: 12}


@pytest.mark.parametrize("world", [2, 3])
def test_bench_runs_with_several_ranks(dev, world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(ROOT / "bench.py"),
           "--gpus", str(world), "--steps", "6", "--warmup", "3", "--bg-res", "128", "--bg-voxel", "0.04",
           "--obj-res", "32", "--objects-per-gpu", "2", "--width", "320", "--height", "240",
           "--no-cpu-baseline", "--comm", "gloo", "--strong-objects", str(STRONG[world])]
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
    # ---- the line validates itself (round 6): what the transport saw ...
    rc = d["rccl"]
    assert rc["ranks"] == world and rc["ranks_agree"] and len(rc["devices"]) == world
    assert [e["rank"] for e in rc["devices"]] == list(range(world))
    assert rc["distinct_devices"] == 1 and not rc["one_device_per_rank"]  # the rehearsal's ranks share the box's one GPU
    assert all(e["pci_bus_id"] for e in rc["devices"])
    # ... replicas and joint images against a single-rank re-run of the same frames ...
    par = d["sharded_parity"]
    assert par["ok"], par
    assert par["replicas_and_joint_images_identical_on_all_ranks"] and par["vs_single_rank"]["visible_sets_equal"]
    assert par["vs_single_rank"]["labels_in_segmentation"] >= 2
    # ... every rank's kernels priced (no committed profile for this toy share: bound null, durations there) ...
    pr = d["per_rank_roofline"]
    assert [e["rank"] for e in pr] == list(range(world)) and all(e["raycast_ms"] > 0 for e in pr)
    assert all(e["objects"] == 2 for e in pr)
    # ... and the strong-scaling sub-run: a FIXED scene split over the ranks
    st = d["strong_scaling"]
    assert st["scaling"] == "strong" and st["objects_total"] == STRONG[world] and st["n_gpus"] == world and st["value"] > 0
    assert sum(st["objects_per_gpu"]) == STRONG[world] and max(st["objects_per_gpu"]) - min(st["objects_per_gpu"]) <= 1
    assert st["launches_per_stage"] == (2 if world == 2 else 1)
    assert d["config"]["path"] == "batched"


def test_rccl_failure_exits_non_zero_instead_of_falling_back(dev):
    """Two ranks on ONE GPU: RCCL refuses (two ranks on one device), and bench.py must refuse to measure something else."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(ROOT / "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--bg-res", "64", "--bg-voxel", "0.08",
           "--obj-res", "32", "--objects-per-gpu", "1", "--width", "160", "--height", "120", "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode != 0, p.stdout[-2000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert "RCCL communicator could NOT be created" in p.stderr + p.stdout
