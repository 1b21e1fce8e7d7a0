"""Dataset readers: what the reference's TUMRGBDReader
(src/utils/TUMRGBDReader.cpp) and MaskRCNN::loadPreprocessed (src/core/MaskRCNN.cpp:250-282) deliver
to the fusion loop.  Pure Python / numpy with the standard library only (there is no OpenCV here): a
16-bit grayscale PNG decoder is ~60 lines with zlib.

    reader = TUMReader("/data/rgbd_dataset_freiburg3_walking_xyz/")
    for index, depth in reader:          # float32 (H, W), metres, 0 = invalid
        ...
"""
from __future__ import annotations

import pickle
import struct
import zlib
from pathlib import Path
from typing import Iterator, List, Tuple

import numpy as np

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def read_png_gray(path) -> np.ndarray:
    """Decode a non-interlaced 8- or 16-bit grayscale PNG to uint8 / uint16 (H, W)."""
    raw = Path(path).read_bytes()
    if raw[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(raw):
        (n,), kind = struct.unpack(">I", raw[pos:pos + 4]), raw[pos + 4:pos + 8]
        body = raw[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    if hdr is None:
        raise ValueError(f"{path}: no IHDR chunk")
    w, h, depth, color, _, _, interlace = hdr
    if color != 0 or depth not in (8, 16) or interlace != 0:
        raise ValueError(f"{path}: only non-interlaced 8/16-bit grayscale PNGs are supported "
                         f"(color type {color}, depth {depth}, interlace {interlace})")
    bpp = depth // 8
    stride = w * bpp
    data = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        f, line = int(data[y, 0]), data[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:  # Up
            cur = (line + prev) & 255
        elif f in (1, 3, 4):  # Sub, Average, Paeth: sequential in x
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 255
        else:
            raise ValueError(f"{path}: bad filter type {f}")
        out[y] = cur
        prev = cur
    if bpp == 1:
        return out
    return out.reshape(h, w, 2).astype(np.uint16) @ np.array([256, 1], np.uint16)  # big endian


def write_png_gray16(path, image: np.ndarray) -> None:
    """Minimal writer (filter 0) -- used by the tests and to stage synthetic sequences."""
    img = np.ascontiguousarray(image, np.uint16)
    h, w = img.shape
    rows = np.zeros((h, 1 + 2 * w), np.uint8)
    rows[:, 1::2], rows[:, 2::2] = (img >> 8).astype(np.uint8), (img & 255).astype(np.uint8)

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))
    Path(path).write_bytes(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)) + chunk(b"IEND", b""))


def read_associations(filename) -> Tuple[List[str], List[str], List[float]]:
    """TUM associations.txt -> (rgb names, depth names, timestamps): lines of four blank-separated
    fields `t1 file1 t2 file2`; which of the two is the colour image is decided by the first line
    (starts with "rgb/"), as TUMRGBDReader::readFileAssociations does."""
    rgb, depth, stamps = [], [], []
    rgb_first = None
    for line in Path(filename).read_text().splitlines():
        parts = line.replace("\t", " ").split(" ")
        if len(parts) != 4 or line.lstrip().startswith("#"):  # (comment lines: tolerated here)
            continue
        if rgb_first is None:
            rgb_first = parts[1].startswith("rgb/")
        rgb.append(parts[1] if rgb_first else parts[3])
        depth.append(parts[3] if rgb_first else parts[1])
        stamps.append(float(parts[0]))
    return rgb, depth, stamps


class TUMReader:
    """Depth frames of a TUM RGB-D sequence in metres (raw / 5000, TUMRGBDReader.cpp readFrame)."""

    def __init__(self, path):
        self.path = Path(path)
        self.rgb_names, self.depth_names, self.stamps = read_associations(self.path / "associations.txt")
        n = len(self.depth_names)
        self.frame_rate = n / (self.stamps[-1] - self.stamps[0]) if n > 1 else 0.0

    def __len__(self):
        return len(self.depth_names)

    def depth(self, index: int) -> np.ndarray:
        raw = read_png_gray(self.path / self.depth_names[index])
        return raw.astype(np.float32) * np.float32(1 / 5000.0)

    def __iter__(self) -> Iterator[Tuple[int, np.ndarray]]:
        for i in range(len(self)):
            yield i, self.depth(i)


def load_preprocessed_masks(filename):
    """A Mask R-CNN result file of the reference's preprocessing script (maskrcnn.in.py:258-268,
    read by MaskRCNN::loadPreprocessed): a pickle of (boxes (N, 4), masks (H, W, N) bool / uint8,
    scores (N, 81)).  Returns (boxes, [uint8 (H, W) 0/1 mask per instance], scores)."""
    with open(filename, "rb") as f:
        boxes, masks, scores = pickle.load(f, encoding="latin1")
    masks = np.asarray(masks)
    per_instance = [np.ascontiguousarray(masks[:, :, i] != 0, np.uint8) for i in range(masks.shape[2])] \
        if masks.ndim == 3 else []
    return np.asarray(boxes), per_instance, np.asarray(scores)
