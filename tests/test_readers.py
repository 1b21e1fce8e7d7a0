"""Dataset readers (SURVEY f-5): TUM associations + 16-bit depth PNGs (/5000 -> metres) and the
preprocessed Mask R-CNN pickles, against files written by the test itself (no dataset here)."""
import pickle
import struct
import zlib

import numpy as np
import pytest

from pathlib import Path

from emfusion_amd import readers

ROOT = Path(__file__).resolve().parents[1]


def _png_with_filters(path, img16):
    """16-bit grayscale PNG whose rows use all five filter types."""
    h, w = img16.shape
    raw = np.zeros((h, 2 * w), np.uint8)
    raw[:, 0::2], raw[:, 1::2] = (img16 >> 8).astype(np.uint8), (img16 & 255).astype(np.uint8)
    rows, prev = [], np.zeros(2 * w, np.int32)
    for y in range(h):
        f, cur = y % 5, raw[y].astype(np.int32)
        a = np.concatenate([[0, 0], cur[:-2]])
        c = np.concatenate([[0, 0], prev[:-2]])
        if f == 0:
            pred = 0
        elif f == 1:
            pred = a
        elif f == 2:
            pred = prev
        elif f == 3:
            pred = (a + prev) >> 1
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        rows.append(bytes([f]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))
    data = zlib.compress(b"".join(rows))
    path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)) +
                     chunk(b"IDAT", data[:100]) + chunk(b"IDAT", data[100:]) + chunk(b"IEND", b""))


def test_png16_roundtrip_all_filters(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 65536, (23, 31), dtype=np.uint16)
    img[5:9, 3:20] = 0
    _png_with_filters(tmp_path / "a.png", img)
    assert np.array_equal(readers.read_png_gray(tmp_path / "a.png"), img)
    readers.write_png_gray16(tmp_path / "b.png", img)
    assert np.array_equal(readers.read_png_gray(tmp_path / "b.png"), img)
    (tmp_path / "bad.png").write_bytes(b"not a png")
    with pytest.raises(ValueError):
        readers.read_png_gray(tmp_path / "bad.png")


def test_png16_full_frame_with_adaptive_filters_is_fast_and_both_decoders_agree(tmp_path, monkeypatch):
    """A 640 x 480 TUM-sized depth image whose rows use Sub / Average / Paeth (what libpng writes):
    the C unfilter of the host library decodes it in milliseconds; the numpy / Python path used
    when the library is missing gives the same pixels."""
    import time
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:480, 0:640]
    img = (5000 * (1.5 + 0.001 * xx + 0.0005 * yy) + rng.integers(0, 30, (480, 640))).astype(np.uint16)
    img[rng.uniform(size=img.shape) < 0.02] = 0
    readers.write_png_gray16(tmp_path / "d.png", img, filters=rng.choice([1, 3, 4, 2, 0], 480))
    t0 = time.time()
    got = readers.read_png_gray(tmp_path / "d.png")
    dt = time.time() - t0
    assert np.array_equal(got, img)
    assert dt < 0.5, dt
    small = img[:12, :40]
    readers.write_png_gray16(tmp_path / "s.png", small, filters=[1, 3, 4, 2, 0, 4, 3, 1, 4, 4, 3, 3])
    from emfusion_amd import pipeline

    def missing():
        raise RuntimeError("no library")
    monkeypatch.setattr(pipeline, "load", missing)
    assert np.array_equal(readers.read_png_gray(tmp_path / "s.png"), small)


@pytest.mark.parametrize("rgb_first", [True, False])
def test_tum_sequence(tmp_path, rgb_first):
    (tmp_path / "depth").mkdir()
    depth_m = np.array([[0.0, 1.0, 2.5], [0.4, 13.1, 0.0002]], np.float32)
    lines = []
    for i in range(3):
        raw = np.round(depth_m * 5000 + i).astype(np.uint16)
        readers.write_png_gray16(tmp_path / "depth" / f"{i}.png", raw)
        t = 100.0 + 0.5 * i
        lines.append(f"{t} rgb/{i}.png {t + 0.01} depth/{i}.png" if rgb_first else
                     f"{t} depth/{i}.png {t + 0.01} rgb/{i}.png")
    (tmp_path / "associations.txt").write_text("\n".join(lines) + "\n# trailing comment line\n")
    r = readers.TUMReader(tmp_path)
    assert len(r) == 3 and r.depth_names == [f"depth/{i}.png" for i in range(3)]
    assert r.rgb_names[0] == "rgb/0.png" and r.frame_rate == pytest.approx(3.0)
    frames = list(r)
    assert [i for i, _ in frames] == [0, 1, 2]
    want = np.round(depth_m * 5000 + 2).astype(np.uint16).astype(np.float32) * np.float32(1 / 5000.0)
    assert frames[2][1].dtype == np.float32 and np.array_equal(frames[2][1], want)
    assert frames[0][1][0, 0] == 0  # invalid stays exactly 0


def test_preprocessed_masks(tmp_path):
    """The layout generate_result() of the reference's maskrcnn.in.py pickles: three parallel lists."""
    m0, m1 = np.zeros((4, 5), bool), np.zeros((4, 5), np.uint8)
    m0[1:3, 1:4] = True
    m1[0, :] = 1
    boxes = [[1, 1, 3, 4], [0, 0, 1, 5]]
    scores = np.random.default_rng(1).random((2, 81))
    with open(tmp_path / "Mask0000.plk", "wb") as f:
        pickle.dump((boxes, [m0, m1], scores.tolist()), f, protocol=pickle.HIGHEST_PROTOCOL)
    b, m, s = readers.load_preprocessed_masks(tmp_path / "Mask0000.plk")
    assert len(m) == 2 and m[0].dtype == np.uint8 and m[0].shape == (4, 5) and m[0].sum() == 6 and m[1].sum() == 5
    assert np.array_equal(b, np.array(boxes)) and np.allclose(s, scores) and s.shape == (2, 81)
    with open(tmp_path / "Mask0001.plk", "wb") as f:  # a frame without detections
        pickle.dump(([], [], []), f, protocol=2)
    b, m, s = readers.load_preprocessed_masks(tmp_path / "Mask0001.plk")
    assert b.shape == (0, 4) and m == [] and s.shape[0] == 0
    with open(tmp_path / "Mask0002.plk", "wb") as f:  # (N, H, W) array: same items
        pickle.dump((np.array(boxes), np.stack([m0, m1.astype(bool)]), scores), f, protocol=2)
    assert [x.sum() for x in readers.load_preprocessed_masks(tmp_path / "Mask0002.plk")[1]] == [6, 5]


# ---- OpenEXR depth files (Co-Fusion datasets, reference ImageReader.cpp) ---------------------------

def write_exr(path, channels, compression=3, pixel="f"):
    """Minimal scan-line OpenEXR writer for the tests (inverse of readers.read_exr, from the same
    published layout): channels = {name: (H, W) array}; compression 0 NONE, 1 RLE, 2 ZIPS, 3 ZIP."""
    import struct
    import zlib
    names = sorted(channels)
    h, w = channels[names[0]].shape
    dt = {"f": np.dtype("<f4"), "h": np.dtype("<f2"), "u": np.dtype("<u4")}[pixel]
    ptype = {"u": 0, "h": 1, "f": 2}[pixel]

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iIii", ptype, 0, 1, 1) for n in names) + b"\0"
    head = b"\x76\x2f\x31\x01" + struct.pack("<i", 2)
    head += attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression]))
    head += attr("dataWindow", "box2i", struct.pack("<4i", 0, 0, w - 1, h - 1))
    head += attr("displayWindow", "box2i", struct.pack("<4i", 0, 0, w - 1, h - 1))
    head += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1))
    head += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0))
    head += attr("screenWindowWidth", "float", struct.pack("<f", 1)) + b"\0"
    per = {0: 1, 1: 1, 2: 1, 3: 16}[compression]

    def rle(b):
        out, i = bytearray(), 0
        while i < len(b):
            run = 1
            while i + run < len(b) and b[i + run] == b[i] and run < 128:
                run += 1
            if run >= 3:
                out += bytes([run - 1, b[i]])
                i += run
            else:
                j = i
                while j < len(b) and j - i < 127 and not (j + 2 < len(b) and b[j] == b[j + 1] == b[j + 2]):
                    j += 1
                out += bytes([(i - j) & 0xFF]) + b[i:j]
                i = j
        return bytes(out)
    blocks = []
    for y0 in range(0, h, per):
        rawb = b"".join(np.ascontiguousarray(channels[n][y], dt).tobytes()
                        for y in range(y0, min(y0 + per, h)) for n in names)
        data = rawb
        if compression:
            t = np.frombuffer(rawb, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]]).astype(np.int64)
            p = t.copy()
            p[1:] = (t[1:] - t[:-1] + 128 + 256) & 0xFF
            pb = p.astype(np.uint8).tobytes()
            cand = rle(pb) if compression == 1 else zlib.compress(pb)
            if len(cand) < len(rawb):
                data = cand
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    table_at = len(head)
    offs, pos = [], table_at + 8 * len(blocks)
    for b in blocks:
        offs.append(pos)
        pos += len(b)
    with open(path, "wb") as f:
        f.write(head + struct.pack("<%dQ" % len(offs), *offs) + b"".join(blocks))


def test_exr_reader_on_a_real_file():
    """python_logo.exr was written by the OpenEXR library (tests/golden/make_exr_golden.py): 4 HALF
    channels, uncompressed; the same image as an 8-bit PNG gives the expected values."""
    from emfusion_amd.readers import read_exr
    gold = ROOT / "tests" / "golden"
    want = np.load(gold / "python_logo_rgba.npy").astype(np.float32) / np.float32(255)
    for k, c in enumerate("RGBA"):
        got = read_exr(gold / "python_logo.exr", c)
        assert got.shape == (16, 16) and got.dtype == np.float32
        assert np.abs(got - want[..., k]).max() < 5e-4
    assert np.array_equal(read_exr(gold / "python_logo.exr"), read_exr(gold / "python_logo.exr", "R"))
    with pytest.raises(ValueError):
        read_exr(gold / "python_logo.exr", "Z")
    with pytest.raises(ValueError):
        read_exr(gold / "python_logo_rgba.npy")


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
@pytest.mark.parametrize("pixel", ["f", "h"])
def test_exr_roundtrip(tmp_path, compression, pixel):
    from emfusion_amd.readers import read_exr
    rng = np.random.default_rng(compression * 7 + ord(pixel))
    h, w = 37, 53  # not a multiple of the 16-line ZIP block
    z = (rng.uniform(0.4, 6.0, (h, w)) * (rng.uniform(size=(h, w)) > 0.1)).astype(np.float32)
    z[5:9] = 2.5  # constant rows: runs for RLE
    if pixel == "h":
        z = z.astype(np.float16).astype(np.float32)
    other = rng.standard_normal((h, w)).astype(np.float32 if pixel == "f" else np.float16).astype(np.float32)
    f = tmp_path / "Depth0000.exr"
    write_exr(f, {"Z": z}, compression, pixel)
    assert np.array_equal(read_exr(f), z)
    write_exr(f, {"A": other, "Z": z, "R": other * 2}, compression, pixel)  # stored alphabetically: A, R, Z
    assert np.array_equal(read_exr(f), z) and np.array_equal(read_exr(f, "A"), other)


def test_image_reader_directory_layout(tmp_path):
    from emfusion_amd.readers import ImageReader, write_png_gray16
    (tmp_path / "colour").mkdir()
    (tmp_path / "depth").mkdir()
    rng = np.random.default_rng(2)
    frames = {}
    for i in range(3, 7):  # the sequence starts at index 3
        d = rng.uniform(0.5, 4.0, (24, 32)).astype(np.float32)
        d[0, :5] = 1e4      # "infinitely far" background of the synthetic datasets
        d[1, 1] = 100.0     # exactly 100 stays
        frames[i] = d
        write_exr(tmp_path / "depth" / ("Depth%04d.exr" % i), {"Z": d}, 3, "f")
        write_png_gray16(tmp_path / "colour" / ("Color%04d.png" % i), np.zeros((24, 32), np.uint16))
    r = ImageReader(tmp_path)
    assert len(r) == 4 and r.first == 3
    got = dict(r)
    assert sorted(got) == [3, 4, 5, 6]
    for i, d in frames.items():
        want = d.copy()
        want[want > 100] = 0
        assert np.array_equal(got[i], want)
    assert got[3][0, 0] == 0 and got[3][1, 1] == 100.0
    (tmp_path / "depth" / "Depth0006.exr").unlink()
    with pytest.raises(RuntimeError):
        ImageReader(tmp_path)  # different number of colour and depth files
    with pytest.raises(RuntimeError):
        ImageReader(tmp_path / "nowhere")
