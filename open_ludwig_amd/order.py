"""Launch-order builders for the stream-collide kernel (performance only; results never depend on the order).

A work item is (block0 << 3) | z0: one 256-thread workgroup steps z-planes z0..z0+3 of block0. MI355X deals
consecutive workgroup ids round-robin to its 8 XCDs (workgroup g -> XCD g % 8, each XCD with a private 4 MiB L2;
MI355X_MICROARCH.md "Workgroup dispatch"), and ids start roughly in order. The pull reads the one-cell face layer of
up to 26 neighbour blocks, i.e. cache lines that the neighbour block's own workgroup also reads, so the order decides
whether that second read is an L2 hit, an Infinity-Cache hit or a second HBM fetch.

Item -1 is a no-op workgroup (used to pad per-XCD sequences to equal length).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

N_XCD = 8


def _items(blocks: np.ndarray) -> np.ndarray:
    """both z-halves of every block, adjacent"""
    b = np.asarray(blocks, dtype=np.int64)
    return np.stack([b << 3, (b << 3) | 4], axis=1).reshape(-1)


def _per_xcd(seqs: Sequence[np.ndarray]) -> np.ndarray:
    """slot g = 8*j + x holds the j-th item of XCD x's sequence; shorter sequences are padded with -1"""
    n = max(len(s) for s in seqs)
    grid = np.full((n, N_XCD), -1, dtype=np.int64)
    for x, s in enumerate(seqs):
        grid[: len(s), x] = s
    return grid.reshape(-1)


def natural(coords: np.ndarray) -> np.ndarray:
    """reference block order (bx slowest, bz fastest), workgroups round-robin over XCDs"""
    return _items(np.arange(len(coords))).astype(np.int32)


def sorted_blocks(coords: np.ndarray, fastest: str = "x") -> np.ndarray:
    c = np.asarray(coords)
    keys = {"x": (c[:, 0], c[:, 1], c[:, 2]), "z": (c[:, 2], c[:, 1], c[:, 0]), "y": (c[:, 1], c[:, 0], c[:, 2])}[fastest]
    return np.lexsort(keys)


def xcd_chunks(coords: np.ndarray, fastest: str = "z") -> np.ndarray:
    """every XCD sweeps one contiguous eighth of the block list (sorted with `fastest` varying fastest)"""
    order = sorted_blocks(coords, fastest)
    it = _items(order)
    per = -(-len(it) // N_XCD)
    per += per % 2   # keep both halves of a block on one XCD
    return _per_xcd([it[x * per:(x + 1) * per] for x in range(N_XCD)]).astype(np.int32)


def xcd_rows(coords: np.ndarray, axis: str = "x", group: int = 1) -> np.ndarray:
    """Rows of blocks along `axis` are dealt to the XCDs round-robin in groups of `group` rows; each XCD sweeps its
    rows one after the other. All 8 XCDs therefore work in the same neighbourhood at the same time (faces shared
    between rows meet in the Infinity Cache) while the faces along the row stay inside one L2."""
    c = np.asarray(coords)
    ax = "xyz".index(axis)
    o1, o2 = [a for a in range(3) if a != ax]
    # row id = (slow other axis, fast other axis)
    row_keys = c[:, o1].astype(np.int64) * (c[:, o2].max() + 1) + c[:, o2]
    uniq, row_of = np.unique(row_keys, return_inverse=True)
    order = np.lexsort((c[:, ax], row_of))            # by row, then along the axis
    row_sorted = row_of[order]
    seqs = [[] for _ in range(N_XCD)]
    starts = np.flatnonzero(np.r_[True, row_sorted[1:] != row_sorted[:-1], True])
    for r in range(len(starts) - 1):
        x = (r // group) % N_XCD
        seqs[x].append(_items(order[starts[r]:starts[r + 1]]))
    seqs = [np.concatenate(s) if s else np.zeros(0, np.int64) for s in seqs]
    return _per_xcd(seqs).astype(np.int32)


def xcd_tiles(coords: np.ndarray, tile=(4, 4, 4)) -> np.ndarray:
    """Blocks grouped into tiles of `tile` blocks; tiles dealt round-robin to XCDs; inside a tile x varies fastest."""
    c = np.asarray(coords).astype(np.int64) - 1
    t = c // np.array(tile)
    tdim = t.max(axis=0) + 1
    tid = (t[:, 2] * tdim[1] + t[:, 1]) * tdim[0] + t[:, 0]
    inner = c % np.array(tile)
    order = np.lexsort((inner[:, 0], inner[:, 1], inner[:, 2], tid))
    tid_sorted = tid[order]
    starts = np.flatnonzero(np.r_[True, tid_sorted[1:] != tid_sorted[:-1], True])
    seqs = [[] for _ in range(N_XCD)]
    for r in range(len(starts) - 1):
        seqs[r % N_XCD].append(_items(order[starts[r]:starts[r + 1]]))
    seqs = [np.concatenate(s) if s else np.zeros(0, np.int64) for s in seqs]
    return _per_xcd(seqs).astype(np.int32)


BUILDERS = {
    "natural": natural,
    "chunks_z": lambda c: xcd_chunks(c, "z"),
    "chunks_x": lambda c: xcd_chunks(c, "x"),
    "rows_x": lambda c: xcd_rows(c, "x", 1),
    "rows_z": lambda c: xcd_rows(c, "z", 1),
    "rows_x4": lambda c: xcd_rows(c, "x", 4),
    "tiles444": lambda c: xcd_tiles(c, (4, 4, 4)),
    "tiles844": lambda c: xcd_tiles(c, (8, 4, 4)),
    "tiles882": lambda c: xcd_tiles(c, (8, 8, 2)),
}


def build(name: str, coords) -> np.ndarray:
    return BUILDERS[name](np.asarray(coords))
