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


def test_sequence_mode_equals_the_python_driver(dev, tmp_path):
    """apps/emfusion_synth --sequence -- TUMRGBDReader, EMFusion::usePreprocMasks, processFrame(frame), getLastMasks,
    writeResults, all in C++ (reference apps/EM-Fusion.cpp:100-204) -- on a staged TUM sequence: the pose files must equal
    those of apps/run_tum.py (the Python readers, the C handle API) byte for byte, and the volume dumps too."""
    import sys
    from tests import tum_staging as T
    seq, masks, truth = T.stage(tmp_path)
    out_cpp, out_py = tmp_path / "out_cpp", tmp_path / "out_py"
    p = subprocess.run([str(APP), "--sequence", seq, "--masks", masks, "--out", str(out_cpp), "--volumes", *T.SMALL],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert "2 instances in the last mask frame" in p.stdout
    q = subprocess.run([sys.executable, str(ROOT / "apps" / "run_tum.py"), seq, "--masks", masks, "--out", str(out_py), "--volumes",
                        *T.SMALL], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stdout[-1500:] + q.stderr[-1500:]
    names = sorted(f.name for f in out_py.glob("poses-*.txt"))
    assert "poses-cam.txt" in names and "poses-1.txt" in names and names == sorted(f.name for f in out_cpp.glob("poses-*.txt"))
    for name in names:
        assert (out_cpp / name).read_bytes() == (out_py / name).read_bytes(), name
    assert len((out_cpp / "poses-cam.txt").read_text().splitlines()) == T.N
    for name in ("bg_tsdf.bin", "tsdf_1.bin", "fgProbs_1.bin"):
        assert (out_cpp / "tsdfs" / name).read_bytes() == (out_py / "tsdfs" / name).read_bytes(), name


def test_dir_mode_reads_a_cofusion_layout_like_the_tum_one(dev, tmp_path):
    """apps/emfusion_synth --dir -- emf::ImageReader + readExr (reference src/utils/ImageReader.cpp, apps/EM-Fusion.cpp:
    118-126) -- on the staged sequence re-written as ColorNNNN.png + DepthNNNN.exr (ZIP, float; starting at index 3, as
    the Co-Fusion sequences do not start at 0): the same depth values as the PNGs hold, so every result file must equal
    the --sequence run's byte for byte."""
    import numpy as np
    from emfusion_amd import readers
    from tests import tum_staging as T
    from tests.test_readers import write_exr
    seq, masks, _ = T.stage(tmp_path)
    base = tmp_path / "cofusion"
    (base / "colour").mkdir(parents=True)
    (base / "depth").mkdir()
    for f in range(T.N):
        d = readers.read_png_gray(tmp_path / "seq" / "depth" / f"{f:04d}.png").astype(np.float32) * np.float32(1 / 5000.0)
        write_exr(base / "depth" / f"Depth{f + 3:04d}.exr", {"Z": d}, 3 if f % 2 else 1, "f")
        readers.write_png_gray16(base / "colour" / f"Color{f + 3:04d}.png", np.zeros((T.H, T.W), np.uint16))
    # the mask files are numbered by the frame count, not by the file index (EMFusion.cpp:383-389)
    outs = {}
    for name, args in (("tum", ["--sequence", seq]), ("dir", ["--dir", str(base) + "/"])):
        outs[name] = tmp_path / ("out_" + name)
        p = subprocess.run([str(APP), *args, "--masks", masks, "--out", str(outs[name]), "--volumes", *T.SMALL],
                           cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    names = sorted(f.name for f in outs["tum"].glob("poses-*.txt"))
    assert "poses-cam.txt" in names and names == sorted(f.name for f in outs["dir"].glob("poses-*.txt"))
    for name in names:
        assert (outs["dir"] / name).read_bytes() == (outs["tum"] / name).read_bytes(), name
    for name in ("bg_tsdf.bin", "tsdf_1.bin"):
        assert (outs["dir"] / "tsdfs" / name).read_bytes() == (outs["tum"] / "tsdfs" / name).read_bytes(), name
    # and the debug images of --save-output (setupOutput is on in this mode): one per tracked frame
    assert len(list((outs["dir"] / "huber_weights" / "bg").glob("*.png"))) == T.N - 1
    assert len(list((outs["dir"] / "assoc_weights" / "bg" / "postTrack").glob("*.png"))) == T.N - 1


def test_configfile_drives_the_sequence_mode(dev, tmp_path):
    """--configfile: the reference's configuration syntax (core/Config.cpp; apps/EM-Fusion.cpp:268-371) instead of the
    sizing options -- a file that says what tests/tum_staging.SMALL says must give the same result files."""
    from tests import tum_staging as T
    seq, masks, _ = T.stage(tmp_path)
    f = 525.0 * T.W / 640
    (tmp_path / "small.cfg").write_text(f"""\
[Params]
frameSize = {T.W} {T.H}
[Params.intr]
fx = {f}
fy = {f}
cx = {T.W // 2 - 0.5}
cy = {T.H // 2 - 0.5}
[Params]
globalVolumeDims = 64 64 64
globalVoxelSize = 0.04
volumePose = 0 0 1.28
objVolumeDims = 32 32 32
visibilityThresh = 100
boundary = {round(20 * T.W / 640)}
maskRCNNFrames = {T.MASK_EVERY}
""")
    outs = {}
    for name, args in (("cli", T.SMALL), ("cfg", ["--configfile", str(tmp_path / "small.cfg")])):
        outs[name] = tmp_path / ("out_" + name)
        p = subprocess.run([str(APP), "--sequence", seq, "--masks", masks, "--out", str(outs[name]), "--volumes", *args],
                           cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    names = sorted(f.name for f in outs["cli"].glob("poses-*.txt"))
    assert len(names) >= 2 and names == sorted(f.name for f in outs["cfg"].glob("poses-*.txt"))
    for name in names:
        assert (outs["cfg"] / name).read_bytes() == (outs["cli"] / name).read_bytes(), name
    assert (outs["cfg"] / "tsdfs" / "bg_tsdf.bin").read_bytes() == (outs["cli"] / "tsdfs" / "bg_tsdf.bin").read_bytes()
    (tmp_path / "wrong.cfg").write_text("[Params]\nframeSize = 640 480\n")
    p = subprocess.run([str(APP), "--sequence", seq, "--out", str(tmp_path / "o"), "--configfile", str(tmp_path / "wrong.cfg")],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "the images are" in (p.stdout + p.stderr)
