"""bench.py's `roofline` object is computed from the committed counter summaries under profiles/ and the launch
durations of the run: the arithmetic is host logic and is checked here, on CPU, against the committed profile --
every fraction a utilisation in [0, 1], the bound the largest of them, the byte model kept apart, the provenance
named.  (The numbers themselves are measured on the GPU box; this pins how they are turned into the line.)"""
import importlib.util
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_profile_is_found_and_complete(bench):
    prof = bench.committed_profile()
    assert prof["tag"] and prof["tag"][:3] in ("r03", "r04", "r05", "r06"), prof["tag"]
    assert "--steps 20 --warmup 5" in prof["protocol"]  # the driver's protocol, timed launches only
    for kind in ("raycast", "integrate_bg", "integrate", "assoc", "stream_copy"):
        c = prof["counters"][kind]
        for name in ("SQ_INSTS_VALU", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_READ_sum", "TCC_WRITE_sum", "SQ_WAVE_CYCLES"):
            assert c[name]["per_launch"] > 0, (kind, name)
    # the copy kernel calibrates the request sizes at the L2: 1 GiB read in 128-byte, written in 64-byte requests
    cal = prof["counters"]["stream_copy"]
    assert abs(2 ** 30 / cal["TCC_READ_sum"]["per_launch"] - 128) < 1 and abs(2 ** 30 / cal["TCC_WRITE_sum"]["per_launch"] - 64) < 1
    assert prof["traffic"]["raycast"]["hbm_bytes_per_launch"] > 1e7


def test_resource_fractions_are_utilisations(bench):
    prof = bench.committed_profile()
    for kind, ms in (("raycast", 0.49), ("integrate_bg", 0.45), ("integrate", 0.046), ("assoc", 0.0134)):
        res = bench.resource_fractions(kind, ms, prof)
        assert set(res) >= {"valu", "l1", "l2", "hbm", "mean_waves_per_simd"}
        for r in ("valu", "l1", "l2", "hbm"):
            assert 0.0 < res[r]["frac"] <= 1.0, (kind, r, res[r])
            assert abs(res[r]["frac"] - res[r]["achieved"] / res[r]["peak"]) < 1e-3
        assert 0 < res["mean_waves_per_simd"] <= 8
    assert bench.resource_fractions("raycast", 0.0, prof) is None and bench.resource_fractions("nonesuch", 1.0, prof) is None
    assert bench.PEAKS["valu"][1] == pytest.approx(1228.8) and bench.PEAKS["l1"][1] == pytest.approx(614.4)
    assert bench.PEAKS["hbm"][1] == 8000.0 and bench.PEAKS["l2"][1] == 34500.0


def test_roofline_object_of_a_bench_line(bench):
    kern = {"raycast": {"units": 307200 * 5 * 20, "launches": 20, "total_ms": 20 * 0.4876},
            "integrate_bg": {"units": 134217728 * 20, "launches": 20, "total_ms": 20 * 0.452}, "_dropped": 0}
    roof, rows = bench.roofline(kern, (70e6 * 20, 280000 * 20, 0, 0), 307200, 5900.0, True)
    assert roof["kernel"] == "k_raycast" and roof["bound"] in ("valu", "l1", "l2", "hbm")
    fr = {k: v["frac"] for k, v in roof["resources"].items() if isinstance(v, dict)}
    assert roof["frac"] == max(fr.values()) and roof["bound"] == max(fr, key=fr.get) and roof["frac"] <= 1.0
    assert roof["achieved"] / roof["peak"] == pytest.approx(roof["frac"], abs=1e-3)
    assert roof["traffic"] > 1e7 and "profiles/r0" in roof["counters_from"]
    assert roof["model_GBs"] > 8000 and "not a roofline fraction" in roof["model_note"]  # the byte model: a rate, kept apart
    integ = roof["integrate_stream"]
    assert integ["concurrent_with_raycast"] and 0 < integ["frac"] <= 1.0
    both = roof["chip_while_raycast_runs"]
    assert all(both[k] >= fr[k] for k in both)
    json.dumps(roof)  # the line must serialise
    # another workload than the profiled one: nothing to price the kernel against, and the line says so
    roof2, _ = bench.roofline(kern, None, 307200, None, False)
    assert roof2["bound"] is None and roof2["frac"] is None and roof2["traffic"] is None and "none committed" in roof2["counters_from"]


def test_profiles_are_selected_by_workload(bench):
    """A run is priced by counters taken on ITS workload: the headline's key finds the configs[1] profile, the
    configs[4] share its own (profiles/r04_cfg4_*), an unprofiled workload nothing."""
    assert bench.committed_profile(bench.HEADLINE_KEY)["tag"]
    assert bench.committed_profile("640x480_bg64_obj1x32")["tag"] is None
    cfg4 = bench.committed_profile(bench.workload_key(1280, 960, 1024, 256, 2))
    assert cfg4["tag"] and "cfg4" in cfg4["tag"] and cfg4["counters"]["integrate_bg"]["SQ_INSTS_VALU"]["per_launch"] > 0
    assert cfg4["traffic"]["integrate_bg"]["hbm_bytes_per_launch"] > 1e8


def test_the_committed_counters_are_not_older_than_the_kernels_they_price(bench):
    """roofline divides THIS run's launch durations into counters of a committed profiler run: a change to the frame's
    kernels without a new profile would price new time with old counts.  The newest configs[1] profile must have been
    committed no earlier than the last commit that touched the sources of the frame's kernels."""
    import subprocess
    if not (ROOT / ".git").exists():
        pytest.skip("no git history here (the GPU box gets a snapshot of the tree)")

    def last_commit_time(paths):
        out = subprocess.run(["git", "log", "-1", "--format=%ct", "--"] + paths, cwd=ROOT, capture_output=True, text=True)
        return int(out.stdout.strip() or 0)
    prof = bench.committed_profile()
    kernels = [f"emfusion_amd/csrc/{f}" for f in ("batched.hip", "march_wave.hpp", "device_core.hpp", "volume_sweep.hip",
                                                  "raycast.hip", "pixel_ops.hip", "common.hpp")]
    t_prof, t_src = last_commit_time([f"profiles/{prof['file']}"]), last_commit_time(kernels)
    assert t_prof > 0, f"profiles/{prof['file']} is not committed"
    assert t_prof >= t_src, (f"profiles/{prof['file']} was committed before the last change to the frame's kernels: "
                             "run scripts/profile_round.sh and commit the new summaries")


def test_host_thread_count_respects_affinity_and_quota(monkeypatch, tmp_path):
    import os
    from oracle import binding
    n = binding.host_threads()
    assert 1 <= n <= len(os.sched_getaffinity(0))
