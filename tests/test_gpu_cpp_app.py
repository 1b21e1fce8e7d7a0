"""The class-level drop-in from C++: apps/emfusion_synth is the reference's main loop (apps/EM-Fusion.cpp)
on emf::EMFusion, linked against libemf_fusion.so -- no Python, no C handle API in between."""
import re
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
APP = ROOT / "apps" / "emfusion_synth"
SMALL = ["--frames", "12", "--objects", "2", "--bg-res", "128", "--obj-res", "32", "--width", "320", "--height", "240"]


def run(*args):
    if not APP.exists():
        pytest.fail("apps/emfusion_synth is not built (python -c 'import __graft_entry__ as g; g.build()')")
    p = subprocess.run([str(APP), *SMALL, *args], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    return p.stdout


def test_supplied_poses_loop_and_result_files(dev, tmp_path):
    out = run("--out", str(tmp_path))
    assert "batched launches: yes" in out and re.search(r"12 frames, 2 objects, bg 128\^3", out)
    m = re.search(r"background mesh (\d+) vertices, (\d+) triangles", out)
    assert m and int(m.group(1)) > 1000 and int(m.group(2)) > 300
    for name in ("poses-cam.txt", "poses-1.txt", "poses-1-corrected.txt", "mesh_bg.ply", "mesh_1.ply", "mesh_2.ply",
                 "tsdfs/bg_tsdf.bin", "tsdfs/tsdf_1.bin", "tsdfs/fgProbs_2.bin"):
        assert (tmp_path / name).stat().st_size > 0, name
    head = (tmp_path / "mesh_bg.ply").read_text().splitlines()[:3]
    assert head[0] == "ply" and head[2] == "element vertex " + m.group(1)
    assert len((tmp_path / "poses-cam.txt").read_text().splitlines()) == 12


def test_autonomous_loop(dev):
    out = run("--autonomous")
    m = re.search(r"autonomous: (\d+) objects spawned from masks; camera position error after 11 tracked frames: ([0-9.]+) mm", out)
    assert m, out
    assert int(m.group(1)) == 2 and float(m.group(2)) < 25.0
