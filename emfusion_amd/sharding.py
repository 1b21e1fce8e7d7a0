"""Host-side bookkeeping of the object-sharded multi-GPU layout.

Mirrors emfusion_amd/csrc/core/Communicator.hpp (ownerOf) and the key format of
emfusion_amd/csrc/multigpu.hip so that harness code and CPU tests can reason about who owns what
and what the all-reduced composite keys mean.  Pure numpy, no device code.
"""
from __future__ import annotations

import numpy as np

NO_HIT = np.uint64(0xFFFFFFFFFFFFFFFF)


def owner_of(object_id: int, world: int) -> int:
    """Rank that holds an object volume: round-robin by 1-based object id."""
    return (object_id - 1) % world


def bg_band(rank: int, world: int, height: int, tile: int = 16):
    """(first row, rows) of the background-raycast band of `rank` (Communicator.hpp bgBandRows):
    whole 16-row tiles, ceil(tiles / world) per rank; trailing ranks may get a short or empty band."""
    tiles = (height + tile - 1) // tile
    rows = ((tiles + world - 1) // world) * tile
    r0 = min(rank * rows, tiles * tile)
    return r0, max(0, min(rows, height - r0))


def local_objects(ids, rank: int, world: int):
    return [i for i in ids if owner_of(i, world) == rank]


def pack_hit_keys(raylengths, hit_masks, list_positions) -> np.ndarray:
    """key = (float_bits(raylength) << 32) | list position, minimum over the given objects;
    all ones where none of them hit (same rule as k_pack_keys)."""
    h, w = raylengths[0].shape if len(raylengths) else (0, 0)
    keys = np.full((h, w), NO_HIT, np.uint64)
    for ray, hit, pos in zip(raylengths, hit_masks, list_positions):
        ray = np.ascontiguousarray(ray, np.float32)
        bits = np.where(ray > 0, ray.view(np.uint32), np.uint32(0xFFFFFFFE)).astype(np.uint64)
        cand = (bits << np.uint64(32)) | np.uint64(pos)
        cand = np.where(np.asarray(hit) != 0, cand, NO_HIT)
        keys = np.minimum(keys, cand)
    return keys


def unpack_hit_keys(keys):
    """(raylength f32, list position int32 or -1) per pixel."""
    keys = np.asarray(keys, np.uint64)
    none = keys == NO_HIT
    pos = np.where(none, -1, (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)).astype(np.int32)
    bits = (keys >> np.uint64(32)).astype(np.uint32)
    ray = np.where(none, np.float32(0), bits.view(np.float32))
    return ray.astype(np.float32), pos
