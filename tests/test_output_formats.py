"""The reference's on-disk result formats (SURVEY f-4): PLY meshes (writeMesh, EMFusion.cpp:1263-1300), raw volume dumps
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


def read_ply(path):
    lines = path.read_text().splitlines()
    end = lines.index("end_header")
    nv = int([l for l in lines[:end] if l.startswith("element vertex")][0].split()[-1])
    nf = int([l for l in lines[:end] if l.startswith("element face")][0].split()[-1])
    body = lines[end + 1:]
    assert len(body) == nv + nf
    v = np.array([l.split() for l in body[:nv]], np.float64).reshape(nv, 6)
    f = np.array([l.split() for l in body[nv:]], np.int64).reshape(nf, 4)
    return lines[:end + 1], v, f


def test_ply_is_the_reference_layout(tmp_path):
    from emfusion_amd import pipeline
    v = np.array([[0, 0.5, -1.25], [1e-7, 2, 3], [1, 1, 1]], np.float32)
    n = np.array([[0, 0, 1], [0.25, 0.5, 0.125], [-1, 0, 0]], np.float32)
    t = np.array([[3, 0, 2, 1]], np.int32)
    f = tmp_path / "mesh_bg.ply"
    pipeline.write_mesh(f, v, n, t)
    text = f.read_text().splitlines()
    assert text[:12] == ["ply", "format ascii 1.0", "element vertex 3", "property float x",
                         "property float y", "property float z", "property float nx", "property float ny",
                         "property float nz", "element face 1", "property list uchar int vertex_index",
                         "end_header"]
    assert text[12] == "0.000000 0.500000 -1.250000 0.000000 0.000000 1.000000"  # %f
    assert text[13].startswith("0.000000 2.000000 3.000000 0.250000")
    assert text[15] == "3 0 2 1" and len(text) == 16
    pipeline.write_mesh(tmp_path / "empty.ply", np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 4)))
    hdr, vv, ff = read_ply(tmp_path / "empty.ply")
    assert len(vv) == 0 and len(ff) == 0


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
    # meshes: the files hold what getMesh returns, which is what the oracle's marching cubes gives
    from oracle import binding as orc
    for mid, name, vox in ((0, "mesh_bg.ply", 0.04), (oid, f"mesh_{oid}.ply", None)):
        v, n, t = fus.mesh(mid)
        _, fv, ff = read_ply(tmp_path / name)
        assert len(v) > 50 and fv.shape == (len(v), 6) and np.array_equal(ff, t)
        assert np.allclose(fv[:, :3], v, atol=1e-6) and np.allclose(fv[:, 3:], n, atol=1e-6)
        if vox is not None:
            want = orc.marching_cubes(fus.volume("tsdf", 0), fus.volume("weights", 0), vox)
            assert all(a.tobytes() == b.tobytes() for a, b in zip((v, n, t), want))
    # the object's mesh only covers foreground voxels (ObjTSDF::getMesh)
    vs_obj = np.float32(vs) / np.float32(32)
    want = orc.marching_cubes(fus.volume("tsdf", oid), fus.volume("weights", oid), float(vs_obj),
                              fg=fus.volume("fgmask", oid))
    assert all(a.tobytes() == b.tobytes() for a, b in zip(fus.mesh(oid), want))
    fus.close()
    synth.close()
