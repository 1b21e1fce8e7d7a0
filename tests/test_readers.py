"""Dataset readers (SURVEY f-5): TUM associations + 16-bit depth PNGs (/5000 -> metres) and the
preprocessed Mask R-CNN pickles, against files written by the test itself (no dataset here)."""
import pickle
import struct
import zlib

import numpy as np
import pytest

from emfusion_amd import readers


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
    masks = np.zeros((4, 5, 2), bool)
    masks[1:3, 1:4, 0] = True
    masks[0, :, 1] = True
    boxes = np.array([[1, 1, 3, 4], [0, 0, 1, 5]])
    scores = np.random.default_rng(1).random((2, 81))
    with open(tmp_path / "Mask0000.plk", "wb") as f:
        pickle.dump((boxes, masks, scores), f, protocol=2)
    b, m, s = readers.load_preprocessed_masks(tmp_path / "Mask0000.plk")
    assert len(m) == 2 and m[0].dtype == np.uint8 and m[0].sum() == 6 and m[1].sum() == 5
    assert np.array_equal(b, boxes) and np.allclose(s, scores)
