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
    out = _unfilter(np.ascontiguousarray(data), h, stride, bpp, path)
    if bpp == 1:
        return out
    return out.reshape(h, w, 2).astype(np.uint16) @ np.array([256, 1], np.uint16)  # big endian


def _unfilter(data: np.ndarray, h: int, stride: int, bpp: int, path) -> np.ndarray:
    """Undo the scan-line filters.  libpng writes adaptive filters (Sub / Average / Paeth on most rows
    of a TUM depth image), whose recurrence runs along x: the host library does it in C
    (emf_io_png_unfilter); without the library, numpy handles None / Sub / Up per row and only
    Average / Paeth rows fall back to a Python loop."""
    if (data[:, 0] > 4).any():
        raise ValueError(f"{path}: bad filter type {int(data[:, 0].max())}")
    out = np.zeros((h, stride), np.uint8)
    try:
        from . import pipeline
        lib = pipeline.load()
    except (RuntimeError, OSError, AttributeError):  # no library, or a stale one without this entry
        lib = None
    if lib is not None and hasattr(lib, "emf_io_png_unfilter"):
        rc = lib.emf_io_png_unfilter(data.ctypes.data, h, stride, bpp, out.ctypes.data)
        if rc != 0:
            raise ValueError(f"{path}: emf_io_png_unfilter failed ({rc})")
        return out
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        f, line = int(data[y, 0]), data[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 1:  # Sub: running sum mod 256 per byte lane
            cur = np.empty(stride, np.int32)
            for c in range(bpp):
                cur[c::bpp] = np.cumsum(line[c::bpp]) & 255
        elif f == 2:  # Up
            cur = (line + prev) & 255
        else:  # Average, Paeth: sequential in x
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if f == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 255
        out[y] = cur
        prev = cur
    return out


def write_png_gray16(path, image: np.ndarray, filters=None) -> None:
    """Minimal writer -- used by the tests and to stage synthetic sequences.  filters: None (type 0
    everywhere) or one filter type 0..4 per row (adaptive filtering as libpng writes it)."""
    img = np.ascontiguousarray(image, np.uint16)
    h, w = img.shape
    rows = np.zeros((h, 1 + 2 * w), np.uint8)
    rows[:, 1::2], rows[:, 2::2] = (img >> 8).astype(np.uint8), (img & 255).astype(np.uint8)
    if filters is not None:
        raw = rows[:, 1:].astype(np.int32)
        enc = np.zeros_like(raw)
        zero = np.zeros(2 * w, np.int32)
        for y in range(h):
            f = int(filters[y])
            a = np.concatenate([zero[:2], raw[y, :-2]])
            b = raw[y - 1] if y else zero
            c = np.concatenate([zero[:2], b[:-2]])
            if f == 0:
                pred = zero
            elif f == 1:
                pred = a
            elif f == 2:
                pred = b
            elif f == 3:
                pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
            enc[y] = (raw[y] - pred) & 255
            rows[y, 0] = f
        rows[:, 1:] = enc.astype(np.uint8)

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


# ---- OpenEXR scan-line files (the depth images of the Co-Fusion datasets) ---------------------------

_EXR_LINES = {0: 1, 1: 1, 2: 1, 3: 16}  # scan lines per block: NONE, RLE, ZIPS, ZIP
_EXR_PIXEL = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}


def _exr_unrle(data: bytes, size: int) -> bytes:
    out, i = bytearray(), 0
    while i < len(data):
        n = data[i] - 256 if data[i] > 127 else data[i]
        i += 1
        if n < 0:  # -n literal bytes
            out += data[i:i - n]
            i -= n
        else:      # the next byte n + 1 times
            out += data[i:i + 1] * (n + 1)
            i += 1
    if len(out) != size:
        raise ValueError("EXR: RLE block decodes to %d bytes, expected %d" % (len(out), size))
    return bytes(out)


def _exr_unpredict(buf: bytes) -> bytes:
    """Undo the byte predictor and the even / odd split OpenEXR applies before RLE / ZIP."""
    t = np.frombuffer(buf, np.uint8).astype(np.int64)
    t[1:] -= 128
    t = (np.cumsum(t) & 0xFF).astype(np.uint8)
    half = (len(t) + 1) // 2
    out = np.empty(len(t), np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def read_exr(path, channel: str | None = None) -> np.ndarray:
    """One channel of a single-part scan-line OpenEXR file as float32 (H, W): what
    cv::imread(IMREAD_UNCHANGED) gives the reference for the Co-Fusion depth files
    (ImageReader.cpp:105-110).  Compression NONE, RLE, ZIPS and ZIP; HALF, FLOAT and UINT pixels.
    channel: name to read; default the only channel, else the first of Z, Y, R that exists.
    Written from the published file-format description (openexr.com, "OpenEXR File Layout")."""
    raw = Path(path).read_bytes()
    if len(raw) < 8 or raw[:4] != b"\x76\x2f\x31\x01":
        raise ValueError("%s is not an OpenEXR file" % path)
    version = int.from_bytes(raw[4:8], "little")
    if version & 0xFF != 2 or version & 0x1A00:  # tiled, deep, multi-part
        raise ValueError("%s: only single-part scan-line EXR files are supported" % path)
    pos, attrs = 8, {}
    while raw[pos] != 0:
        end = raw.index(b"\0", pos)
        name = raw[pos:end].decode()
        tend = raw.index(b"\0", end + 1)
        size = int.from_bytes(raw[tend + 1:tend + 5], "little")
        attrs[name] = raw[tend + 5:tend + 5 + size]
        pos = tend + 5 + size
    pos += 1
    chans, c = [], attrs["channels"]
    i = 0
    while c[i] != 0:
        end = c.index(b"\0", i)
        ptype, _, xs, ys = struct.unpack_from("<iIii", c, end + 1)
        chans.append((c[i:end].decode(), ptype, xs, ys))
        i = end + 17
    comp = attrs["compression"][0]
    if comp not in _EXR_LINES:
        raise ValueError("%s: EXR compression %d is not supported (NONE, RLE, ZIPS, ZIP are)" % (path, comp))
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"])
    w, h = xmax - xmin + 1, ymax - ymin + 1
    if any(xs != 1 or ys != 1 for _, _, xs, ys in chans):
        raise ValueError("%s: sub-sampled EXR channels are not supported" % path)
    names = [n for n, _, _, _ in chans]
    if channel is None:
        channel = names[0] if len(names) == 1 else next((n for n in ("Z", "Y", "R") if n in names), None)
    if channel not in names:
        raise ValueError("%s: no channel %r among %s" % (path, channel, names))
    line_bytes = sum(_EXR_PIXEL[t].itemsize * w for _, t, _, _ in chans)
    offset_in_line, dtype = 0, None
    for n, t, _, _ in chans:  # channels are stored in the order of the list (alphabetical)
        if n == channel:
            dtype = _EXR_PIXEL[t]
            break
        offset_in_line += _EXR_PIXEL[t].itemsize * w
    per_block = _EXR_LINES[comp]
    nblocks = (h + per_block - 1) // per_block
    offsets = struct.unpack_from("<%dQ" % nblocks, raw, pos)
    out = np.zeros((h, w), np.float32)
    for off in offsets:
        y, size = struct.unpack_from("<ii", raw, off)
        lines = min(per_block, ymax - y + 1)
        want = lines * line_bytes
        data = raw[off + 8:off + 8 + size]
        if comp != 0 and size < want:  # a block that does not shrink is stored as it is
            data = _exr_unpredict(zlib.decompress(data) if comp in (2, 3) else _exr_unrle(data, want))
        if len(data) != want:
            raise ValueError("%s: EXR block at line %d has %d bytes, expected %d" % (path, y, len(data), want))
        for l in range(lines):
            start = l * line_bytes + offset_in_line
            out[y - ymin + l] = np.frombuffer(data, dtype, w, start).astype(np.float32)
    return out


class ImageReader:
    """Depth frames of a Co-Fusion style dataset (reference ImageReader.cpp): <base>/<colordir>/
    ColorNNNN.png and <base>/<depthdir>/DepthNNNN.exr, depth in metres, values above 100 set to 0.
    Frames start at the first index for which both files exist."""

    def __init__(self, base, colordir="colour", depthdir="depth"):
        self.color, self.depth_dir = Path(base) / colordir, Path(base) / depthdir
        if not (self.color.is_dir() and self.depth_dir.is_dir()):
            raise RuntimeError("Could not read color or depth dir!")
        rgbs = sum(1 for f in self.color.iterdir() if f.suffix == ".png")
        depths = sum(1 for f in self.depth_dir.iterdir() if f.suffix == ".exr")
        if rgbs != depths:
            raise RuntimeError("Different number of rgb and depth files!")
        self.num_frames, self.first = rgbs, 0
        while not (self._color(self.first).exists() and self._depth(self.first).exists()):
            self.first += 1
            if self.first >= rgbs:
                raise RuntimeError("Could not find starting index!")

    def _color(self, i):
        return self.color / ("Color%04d.png" % i)

    def _depth(self, i):
        return self.depth_dir / ("Depth%04d.exr" % i)

    def __len__(self):
        return self.num_frames

    def depth(self, index: int) -> np.ndarray:
        d = read_exr(self._depth(index))
        d[d > 100] = 0
        return d

    def __iter__(self) -> Iterator[Tuple[int, np.ndarray]]:
        for i in range(self.first, self.first + self.num_frames):
            if self._depth(i).exists():
                yield i, self.depth(i)


def load_preprocessed_masks(filename):
    """A Mask R-CNN result file of the reference's preprocessing script: `generate_result`
    (apps/maskrcnn.in.py:188-206) pickles three parallel SEQUENCES over the kept instances -- boxes
    (4 numbers each), masks (one (H, W) array each) and the 81 class scores -- already filtered by
    FILTER_CLASSES / STATIC_OBJECTS; MaskRCNN::loadPreprocessed (MaskRCNN.cpp:250-282) reads them item by
    item (getSegmentation, MaskRCNN.cpp:152-172).  Returns (boxes (N, 4), [uint8 (H, W) 0/1 mask per
    instance], scores (N, 81))."""
    with open(filename, "rb") as f:
        boxes, masks, scores = pickle.load(f, encoding="latin1")
    per_instance = []
    for m in masks:  # a list of arrays, or an (N, H, W) array: one item per instance either way
        m = np.asarray(m)
        if m.ndim != 2:
            raise ValueError("%s: instance masks must be 2-D, got shape %s" % (filename, m.shape))
        per_instance.append(np.ascontiguousarray(m != 0, np.uint8))
    boxes = np.asarray(boxes, np.float64).reshape(-1, 4)
    scores = np.asarray(scores, np.float64).reshape(len(per_instance), -1) if len(per_instance) else np.zeros((0, 81))
    if len(boxes) != len(per_instance):
        raise ValueError("%s: %d boxes for %d masks" % (filename, len(boxes), len(per_instance)))
    return boxes, per_instance, scores
