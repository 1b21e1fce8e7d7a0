"""The reference's on-disk result formats (SURVEY f-4, mesh-free part): raw volume dumps
(EMFusion::writeVolume, EMFusion.cpp:1302-1313) and TUM-style pose files (writePoseFile,
EMFusion.cpp:1238-1254).  Host-only code of libemf_fusion.so: runs without a GPU."""
import struct

import numpy as np
import pytest


def test_volume_dump_layout(tmp_path):
    from emfusion_amd import pipeline
    vol = np.arange(4 * 3 * 2, dtype=np.float32).reshape(4, 3, 2) * 0.5  # (Nz, Ny, Nx)
    f = tmp_path / "bg_tsdf.bin"
    pipeline.write_volume(f, vol, 0.0125)
    raw = f.read_bytes()
    nx, ny, nz = struct.unpack_from("<3i", raw, 0)
    (elem,) = struct.unpack_from("<Q", raw, 12)  # size_t of a 64-bit build
    (voxel,) = struct.unpack_from("<f", raw, 20)
    assert (nx, ny, nz, elem) == (2, 3, 4, 4) and voxel == np.float32(0.0125)
    assert len(raw) == 24 + vol.size * 4
    assert np.array_equal(np.frombuffer(raw, np.float32, offset=24).reshape(4, 3, 2), vol)  # x fastest


def test_pose_file_is_tum_format(tmp_path):
    from emfusion_amd import pipeline
    c, s = np.cos(0.3), np.sin(0.3)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
    Rx180 = np.diag([1, -1, -1]).astype(np.float32)  # trace < 0: the other branch of the conversion
    poses = {7: (Rz, [1.5, -2.25, 0.125]), 2: (np.eye(3), [0, 0, 0]), 11: (Rx180, [0.1, 0.2, 0.3])}
    f = tmp_path / "poses-cam.txt"
    pipeline.write_pose_file(f, poses)
    rows = [line.split() for line in f.read_text().strip().splitlines()]
    assert [int(r[0]) for r in rows] == [2, 7, 11]  # ascending frame index (std::map order)
    assert all(len(r) == 8 for r in rows)
    vals = {int(r[0]): np.array(r[1:], np.float64) for r in rows}
    assert np.allclose(vals[2], [0, 0, 0, 0, 0, 0, 1])
    assert np.allclose(vals[7][:3], [1.5, -2.25, 0.125])
    assert np.allclose(vals[7][3:], [0, 0, np.sin(0.15), np.cos(0.15)], atol=1e-6)  # (x, y, z, w)
    assert np.allclose(np.abs(vals[11][3:]), [1, 0, 0, 0], atol=1e-6)
    assert rows[1][1] == "1.5" and rows[1][2] == "-2.25"  # default ostream formatting


@pytest.mark.gpu
def test_write_results_of_a_run(tmp_path, dev):
    from emfusion_amd import pipeline
    from emfusion_amd.ops import image_view
    from tests.parity_util import to_dev
    W, H = 160, 120
    prm = pipeline.make_params(W, H, 64, 0.04, 32, visibility_thresh=100, boundary=5)
    synth = pipeline.SyntheticStream(W, H, np.array(prm.K, np.float32), 1, seed=0xE3F5)
    fus = pipeline.Fusion(prm, None)
    c, r, vs = synth.sphere(0, 0)
    oid = fus.add_object(c, vs)
    fus.enable_pose_log(True)
    truth = {}
    for f in range(3):
        depth, sid = synth.render(f)
        R, t = synth.camera_pose(f)
        truth[f] = t
        d, m = to_dev(depth), to_dev((sid == 1).astype(np.uint8))
        fus.process_frame(image_view(d), R, t, {oid: (np.eye(3, dtype=np.float32).reshape(-1), synth.sphere(0, f)[0])},
                          {oid: image_view(m)}, f == 0)
        fus.synchronize()
    fus.write_results(tmp_path, volumes=True)
    rows = [line.split() for line in (tmp_path / "poses-cam.txt").read_text().strip().splitlines()]
    assert [int(r[0]) for r in rows] == [0, 1, 2]
    assert np.allclose([float(v) for v in rows[2][1:4]], truth[2], atol=1e-5)
    assert (tmp_path / f"poses-{oid}.txt").exists()
    raw = (tmp_path / "tsdfs" / "bg_tsdf.bin").read_bytes()
    assert struct.unpack_from("<3i", raw, 0) == (64, 64, 64)
    got = np.frombuffer(raw, np.float32, offset=24).reshape(64, 64, 64)
    assert np.array_equal(got, fus.volume("tsdf", 0))
    for name in (f"tsdf_{oid}", f"weights_{oid}", f"fgProbs_{oid}"):
        assert (tmp_path / "tsdfs" / f"{name}.bin").stat().st_size == 24 + 32 ** 3 * 4
    fus.close()
    synth.close()
