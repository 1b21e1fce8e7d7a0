"""The demoted A/B switches (core/types.hpp debugEnv: EMF_FUSE_POINTS, EMF_FUSE_VISIBILITY, EMF_EARLY_FAR_BOUNDS,
EMF_OBJ_CULL, EMF_FAR_SCAN, EMF_RAY_FOOTPRINTS, EMF_BRICK_FLAGS, the tracking driver's EMF_TRACK_WINDOW / EMF_TRACK_CHUNK)
are no-ops in the product build, but the paths behind them stay compiled in (several are functional fall-backs: the
un-fused composite serves the per-volume path, the chunked tracking loop serves hosts without device-visible memory).
libemf_fusion_dbg.so is the same source with -DEMF_DEBUG_SWITCHES (make -C emfusion_amd/csrc dbg); here the switch-pair
test and the tracking tests run against it with the switches live -- and the product build says so when one of them is
set."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _pytest(args, env):
    return subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu"] + args, cwd=ROOT,
                          env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)


def test_demoted_switches_keep_the_bytes_alone_and_in_pairs(dev):
    assert (ROOT / "emfusion_amd" / "libemf_fusion_dbg.so").exists(), "built by __graft_entry__.build() (make dbg)"
    p = _pytest(["tests/test_gpu_switch_pairs.py"], {"EMF_FUSION_VARIANT": "_dbg"})
    assert p.returncode == 0 and "1 passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


@pytest.mark.parametrize("env", [{"EMF_TRACK_WINDOW": "0"}, {"EMF_TRACK_WINDOW": "0", "EMF_TRACK_CHUNK": "0"},
                                 {"EMF_TRACK_WINDOW": "2"}], ids=["chunked_polls", "one_chunk", "window_2"])
def test_the_tracking_drivers_host_loops_agree_with_the_oracle(dev, env):
    """the closed-loop tracking tests (HIP classes against the frame-level oracle) with the other host loops of trackModels"""
    p = _pytest(["tests/test_gpu_tracking_pipeline.py"], dict(env, EMF_FUSION_VARIANT="_dbg"))
    assert p.returncode == 0 and "passed" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


def test_the_product_build_says_when_a_demoted_switch_is_set(dev):
    code = ("import numpy as np\nfrom emfusion_amd import pipeline, devmem\ndevmem.set_device(0)\n"
            "f = pipeline.Fusion(pipeline.make_params(160, 120, 64, 0.04, 32), None)\nf.close()\n")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, EMF_FUSE_POINTS="0", EMF_OBJ_CULL="1"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stderr.count("EMF_FUSE_POINTS is set, but this build ignores it") == 1
    assert p.stderr.count("EMF_OBJ_CULL is set, but this build ignores it") == 1
