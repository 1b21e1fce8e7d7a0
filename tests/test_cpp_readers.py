"""The C++ dataset readers behind apps/emfusion_synth --sequence and EMFusion::usePreprocMasks (core/Readers.cpp;
reference src/utils/TUMRGBDReader.cpp, src/core/MaskRCNN.cpp:250-282 + apps/maskrcnn.in.py:188-268), through the C API,
against files this test writes and against the Python readers (emfusion_amd/readers.py).  No dataset exists here."""
import pickle

import numpy as np
import pytest

from emfusion_amd import pipeline, readers


def test_png_decoder_handles_every_filter_type_and_both_depths(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 65535, (48, 64)).astype(np.uint16)
    for k, filters in enumerate((None, rng.integers(0, 5, 48), np.full(48, 4), np.full(48, 3))):
        readers.write_png_gray16(tmp_path / f"a{k}.png", img, filters=filters)
        got = pipeline.read_depth_png(tmp_path / f"a{k}.png", 1.0)
        assert got.dtype == np.float32 and np.array_equal(got, img.astype(np.float32))
        assert np.array_equal(readers.read_png_gray(tmp_path / f"a{k}.png"), img)
    tum = pipeline.read_depth_png(tmp_path / "a1.png")  # TUM scale: raw / 5000, the same float as the Python reader's
    assert np.array_equal(tum, img.astype(np.float32) * np.float32(1 / 5000.0))
    (tmp_path / "bad.png").write_bytes(b"not a png")
    with pytest.raises(pipeline.FusionError, match="not a PNG"):
        pipeline.read_depth_png(tmp_path / "bad.png")


def test_associations_are_parsed_like_the_reference_reader(tmp_path):
    (tmp_path / "associations.txt").write_text("# comment line\n0.100 rgb/0.png 0.110 depth/0.png\n0.200 rgb/1.png 0.210 depth/1.png\n"
                                               "short line\n")
    assert pipeline.tum_associations(tmp_path / "associations.txt") == [("depth/0.png", 0.1), ("depth/1.png", 0.2)]
    # depth first: decided by the first line (TUMRGBDReader::readFileAssociations)
    (tmp_path / "b.txt").write_text("0.1 depth/a.png 0.1 rgb/a.png\n0.2 depth/b.png 0.2 rgb/b.png\n")
    assert [n for n, _ in pipeline.tum_associations(tmp_path / "b.txt")] == ["depth/a.png", "depth/b.png"]
    py = readers.read_associations(tmp_path / "associations.txt")
    assert py[1] == ["depth/0.png", "depth/1.png"]


@pytest.mark.parametrize("protocol", [0, 1, 2, 3, 4, 5])
def test_mask_pickles_of_every_protocol(tmp_path, protocol):
    """generate_result() pickles three parallel lists: boxes (lists of numbers), masks (2-D slices of an (H, W, N)
    bool array: non-contiguous views) and the 81 class scores (lists of floats); HIGHEST_PROTOCOL of the writing
    interpreter -- 2 for Python 2, 4 or 5 for Python 3."""
    rng = np.random.default_rng(protocol)
    seg = np.zeros((48, 64, 2), bool)
    seg[5:9, 7:9, 0] = True
    seg[10:20, 30:40, 1] = True
    boxes, scores = [[1, 2, 3, 4], [5, 6, 7, 8]], rng.random((2, 81))
    with open(tmp_path / "Mask0000.plk", "wb") as f:
        pickle.dump((boxes, [seg[:, :, 0], seg[:, :, 1]], scores.tolist()), f, protocol=protocol)
    b, m, s = pipeline.load_preproc_masks(tmp_path / "Mask0000.plk")
    assert np.array_equal(b, np.array(boxes, float)) and np.array_equal(s, scores)
    assert m.dtype == np.uint8 and np.array_equal(m[0], seg[:, :, 0]) and np.array_equal(m[1], seg[:, :, 1])
    pb, pm, ps = readers.load_preprocessed_masks(tmp_path / "Mask0000.plk")  # the Python reader agrees
    assert np.array_equal(pb, b) and np.array_equal(np.stack(pm), m) and np.array_equal(ps, s)


def test_mask_pickles_in_other_shapes(tmp_path):
    m0 = np.zeros((48, 64), np.uint8)
    m0[1:3, 2:5] = 1
    boxes, scores = np.array([[1, 2, 3, 4], [5, 6, 7, 8]]), np.random.default_rng(1).random((2, 81))
    with open(tmp_path / "a.plk", "wb") as f:  # arrays instead of lists: int64 boxes, an (N, H, W) stack, float64 scores
        pickle.dump((boxes, np.stack([m0, m0.astype(bool)]), scores), f, protocol=2)
    b, m, s = pipeline.load_preproc_masks(tmp_path / "a.plk")
    assert m.sum() == 12 and np.array_equal(b, boxes.astype(float)) and np.array_equal(s, scores)
    with open(tmp_path / "b.plk", "wb") as f:  # a frame without detections
        pickle.dump(([], [], []), f, protocol=4)
    assert len(pipeline.load_preproc_masks(tmp_path / "b.plk")[1]) == 0
    with open(tmp_path / "c.plk", "wb") as f:  # Fortran-ordered masks, float32 values
        pickle.dump(([[0, 0, 1, 1]], [np.asfortranarray(m0.astype(np.float32))], [[0.0] * 81]), f, protocol=4)
    assert np.array_equal(pipeline.load_preproc_masks(tmp_path / "c.plk")[1][0], m0)
    with open(tmp_path / "d.plk", "wb") as f:
        pickle.dump({"not": "a tuple"}, f, protocol=2)
    with pytest.raises(pipeline.FusionError, match="tuple"):
        pipeline.load_preproc_masks(tmp_path / "d.plk")


def test_malformed_mask_pickles_are_rejected_not_read_out_of_bounds(tmp_path):
    """ragged class-score rows used to be copied `len(row 0)` doubles each; a negative extent in an array's shape
    passed the size test as a product of two negatives."""
    m0 = np.zeros((8, 8), np.uint8)
    with open(tmp_path / "ragged.plk", "wb") as f:
        pickle.dump(([[0, 0, 1, 1]] * 2, [m0, m0], [[0.5] * 81, [0.5] * 3]), f, protocol=2)
    with pytest.raises(pipeline.FusionError, match="different lengths"):
        pipeline.load_preproc_masks(tmp_path / "ragged.plk")
    with open(tmp_path / "ok.plk", "wb") as f:
        pickle.dump(([[0, 0, 1, 1]], [m0], [[0.0] * 81]), f, protocol=2)
    raw = bytearray((tmp_path / "ok.plk").read_bytes())
    at = raw.index(b"K\x08K\x08")  # the shape tuple (8, 8) as two BININT1
    # shape (-8, -8): BININT (4-byte signed) twice instead of BININT1 twice
    raw[at:at + 4] = b""
    raw[at:at] = b"J\xf8\xff\xff\xffJ\xf8\xff\xff\xff"
    (tmp_path / "neg.plk").write_bytes(raw)
    with pytest.raises(pipeline.FusionError, match="shape"):
        pipeline.load_preproc_masks(tmp_path / "neg.plk")


def test_exr_block_offsets_cannot_wrap(tmp_path):
    """an offset-table entry near 2^64 passed `off + 8 > size` by wrapping and was then dereferenced."""
    from tests.test_readers import write_exr
    write_exr(tmp_path / "a.exr", {"Z": np.ones((4, 4), np.float32)}, 0, "f")
    raw = bytearray((tmp_path / "a.exr").read_bytes())
    # the offset table follows the header's terminating zero byte: find it as the first 8 bytes that point inside the file
    import struct
    for at in range(8, len(raw) - 8):
        v = struct.unpack_from("<Q", raw, at)[0]
        if at + 8 * 4 <= v < len(raw) and struct.unpack_from("<Q", raw, at + 8)[0] > v:
            break
    struct.pack_into("<Q", raw, at, 2 ** 64 - 4)
    (tmp_path / "wrap.exr").write_bytes(raw)
    with pytest.raises(pipeline.FusionError, match="offset beyond the file"):
        pipeline.read_exr(tmp_path / "wrap.exr")


def test_exr_data_window_is_held_against_the_file_per_compression(tmp_path):
    """a small uncompressed (NONE) file whose header claims a huge data window must not get its w * h floats
    allocated: uncompressed blocks cannot shrink at all, RLE at best 1 : 64, ZIP ~1 : 1030."""
    import struct
    from tests.test_readers import write_exr
    for comp in (0, 1):
        write_exr(tmp_path / "a.exr", {"Z": np.ones((8, 8), np.float32)}, comp, "f")
        raw = bytearray((tmp_path / "a.exr").read_bytes())
        at = raw.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
        assert struct.unpack_from("<4i", raw, at) == (0, 0, 7, 7)
        struct.pack_into("<4i", raw, at, 0, 0, 7, (2000 if comp == 0 else 60000) - 1)  # 8 x 2000 / 8 x 60000 pixels claimed
        (tmp_path / "big.exr").write_bytes(raw)
        with pytest.raises(pipeline.FusionError, match="larger than the file can hold|truncated"):
            pipeline.read_exr(tmp_path / "big.exr")


# ---- OpenEXR depth files and the Co-Fusion directory layout in C++ (reference src/utils/ImageReader.cpp) --------------

def test_cpp_exr_reader_on_a_real_file():
    """tests/golden/python_logo.exr was written by the OpenEXR library (CPython's test data): the C++ decoder must give
    what the Python one gives, channel by channel, and the committed pixels."""
    from pathlib import Path
    gold = Path(__file__).parent / "golden"
    want = np.load(gold / "python_logo_rgba.npy")
    for k, c in enumerate("RGBA"):
        got = pipeline.read_exr(gold / "python_logo.exr", c)
        assert got.dtype == np.float32 and np.array_equal(got, readers.read_exr(gold / "python_logo.exr", c))
        assert np.abs(got - want[..., k].astype(np.float32) / 255).max() < 2e-3
    assert np.array_equal(pipeline.read_exr(gold / "python_logo.exr"), pipeline.read_exr(gold / "python_logo.exr", "R"))
    with pytest.raises(pipeline.FusionError, match="no channel"):
        pipeline.read_exr(gold / "python_logo.exr", "Z")
    with pytest.raises(pipeline.FusionError, match="not an OpenEXR"):
        pipeline.read_exr(gold / "python_logo_rgba.npy")


@pytest.mark.parametrize("compression", [0, 1, 2, 3], ids=["none", "rle", "zips", "zip"])
@pytest.mark.parametrize("pixel", ["f", "h", "u"])
def test_cpp_exr_reader_against_written_files(tmp_path, compression, pixel):
    from tests.test_readers import write_exr
    rng = np.random.default_rng(10 * compression + ord(pixel))
    if pixel == "u":
        z = rng.integers(0, 70000, (37, 53)).astype(np.float32)
    else:
        z = rng.uniform(0.3, 6.0, (37, 53)).astype(np.float16 if pixel == "h" else np.float32).astype(np.float32)
    z[5:20, 7:30] = z[5, 7]  # runs, so that RLE / ZIP blocks really shrink
    other = np.full_like(z, 3.0)
    f = tmp_path / "Depth0000.exr"
    write_exr(f, {"Z": z}, compression, pixel)
    assert np.array_equal(pipeline.read_exr(f), z) and np.array_equal(readers.read_exr(f), z)
    write_exr(f, {"A": other, "Z": z, "R": other * 2}, compression, pixel)  # stored alphabetically: A, R, Z
    assert np.array_equal(pipeline.read_exr(f), z) and np.array_equal(pipeline.read_exr(f, "A"), other)
    raw = bytearray(f.read_bytes())
    (tmp_path / "cut.exr").write_bytes(raw[:len(raw) - 40])
    with pytest.raises(pipeline.FusionError):
        pipeline.read_exr(tmp_path / "cut.exr")


def test_cpp_exr_reader_rejects_what_it_cannot_decode(tmp_path):
    from tests.test_readers import write_exr
    z = np.ones((16, 16), np.float32)
    write_exr(tmp_path / "a.exr", {"Z": z}, 0, "f")
    raw = bytearray((tmp_path / "a.exr").read_bytes())
    at = raw.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    raw[at] = 4  # PIZ
    (tmp_path / "piz.exr").write_bytes(raw)
    with pytest.raises(pipeline.FusionError, match="compression 4 is not supported"):
        pipeline.read_exr(tmp_path / "piz.exr")
    raw[at] = 0
    raw[5] |= 0x02  # tiled
    (tmp_path / "tiled.exr").write_bytes(raw)
    with pytest.raises(pipeline.FusionError, match="single-part scan-line"):
        pipeline.read_exr(tmp_path / "tiled.exr")


def test_cpp_image_reader_directory_layout(tmp_path):
    from tests.test_readers import write_exr
    (tmp_path / "colour").mkdir()
    (tmp_path / "depth").mkdir()
    for i in range(3, 7):  # the Co-Fusion sequences do not start at 0
        readers.write_png_gray16(tmp_path / "colour" / ("Color%04d.png" % i), np.zeros((4, 4), np.uint16))
        write_exr(tmp_path / "depth" / ("Depth%04d.exr" % i), {"Z": np.full((4, 4), float(i), np.float32)}, 3, "f")
    assert pipeline.image_reader(tmp_path) == (4, 3)
    py = readers.ImageReader(tmp_path)
    assert (len(py), py.first) == (4, 3)
    (tmp_path / "depth" / "Depth0006.exr").unlink()
    with pytest.raises(pipeline.FusionError, match="Different number of rgb and depth files"):
        pipeline.image_reader(tmp_path)
    with pytest.raises(pipeline.FusionError, match="Could not read color or depth dir"):
        pipeline.image_reader(tmp_path / "nowhere")
