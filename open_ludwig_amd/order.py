"""Launch-order builders for the stream-collide kernel (performance only; results never depend on the order).

One work item per WAVE: (block0 << 3) | z = the 8x8 z-plane `z` of block `block0`; -1 = idle wave. Four consecutive
items form a 256-thread workgroup. MI355X deals consecutive workgroup ids round-robin to its 8 XCDs (workgroup g ->
XCD g % 8, each XCD with a private 4 MiB L2; MI355X_MICROARCH.md "Workgroup dispatch"), ids start roughly in order.

Who shares cache lines: the 128-B lines of population k, plane p of a block are read by the waves working on plane
p + cz(k) of that block and of its x/y neighbours (face / edge cells) - never by another plane index. So the
productive grouping is "same plane, x/y-adjacent blocks": inside one workgroup (L1), else on one XCD close in time (L2).
"""
from __future__ import annotations

import numpy as np

N_XCD = 8
WAVES = 4


def _pad4(a):
    a = list(a)
    while len(a) % WAVES:
        a.append(-1)
    return a


def _per_xcd(seqs) -> np.ndarray:
    """seqs[x] = list of workgroups (each 4 items) for XCD x; slot g = 8*j + x; short sequences padded with idle WGs"""
    n = max(len(s) for s in seqs)
    grid = np.full((n, N_XCD, WAVES), -1, dtype=np.int64)
    for x, s in enumerate(seqs):
        if len(s):
            grid[: len(s), x, :] = np.asarray(s, dtype=np.int64).reshape(-1, WAVES)
    return grid.reshape(-1).astype(np.int32)


def block_planes(coords) -> np.ndarray:
    """round-1 baseline: a workgroup = 4 consecutive planes of one block, reference block order, round-robin XCDs"""
    b = np.arange(len(coords), dtype=np.int64)
    z = np.arange(8, dtype=np.int64)
    return ((b[:, None] << 3) | z[None, :]).reshape(-1).astype(np.int32)


def _patches(coords, px: int, py: int, sweep: str):
    """group blocks into px x py patches in (bx,by) at equal bz; returns list of arrays of block ids, in sweep order"""
    c = np.asarray(coords).astype(np.int64) - 1
    kx, ky, kz = c[:, 0] // px, c[:, 1] // py, c[:, 2]
    key = {"xyz": (kx, ky, kz), "xzy": (kx, kz, ky), "zxy": (kz, kx, ky), "yxz": (ky, kx, kz), "yzx": (ky, kz, kx),
           "zyx": (kz, ky, kx)}[sweep]   # fastest first
    order = np.lexsort((c[:, 0], c[:, 1]) + key)
    ks = np.stack([kx, ky, kz], 1)[order]
    cut = np.flatnonzero(np.r_[True, (ks[1:] != ks[:-1]).any(1), True])
    return [order[cut[i]:cut[i + 1]] for i in range(len(cut) - 1)]


def plane_per_xcd(coords, px: int = 2, py: int = 2, sweep: str = "xyz") -> np.ndarray:
    """XCD z handles in-block plane z of every block; a workgroup = that plane of a px x py patch of blocks (4 waves
    per workgroup, so larger patches become several consecutive workgroups on the same XCD)."""
    seqs = [[] for _ in range(N_XCD)]
    for blocks in _patches(coords, px, py, sweep):
        for z in range(8):
            items = _pad4((int(b) << 3) | z for b in blocks)
            for i in range(0, len(items), WAVES):
                seqs[z].append(items[i:i + WAVES])
    return _per_xcd(seqs)


def plane_per_xcd_rot(coords, px: int = 4, py: int = 1, sweep: str = "zxy", rot: str = "bz") -> np.ndarray:
    """as plane_per_xcd, but plane z of a patch at block-z bz goes to XCD (z + bz) % 8: x/y neighbours (same bz, same
    plane) still meet on one XCD, while every XCD now sees every plane index, i.e. all address bits 8..10"""
    c = np.asarray(coords).astype(np.int64) - 1
    seqs = [[] for _ in range(N_XCD)]
    for blocks in _patches(coords, px, py, sweep):
        r = int(c[blocks[0], 2]) if rot == "bz" else int(c[blocks[0], 2] + c[blocks[0], 1])
        for z in range(8):
            items = _pad4((int(b) << 3) | z for b in blocks)
            for i in range(0, len(items), WAVES):
                seqs[(z + r) % N_XCD].append(items[i:i + WAVES])
    return _per_xcd(seqs)


def plane_per_xcd_rot_w8(coords, px: int = 4, py: int = 2, sweep: str = "yxz") -> np.ndarray:
    """Written for the 8-wave workgroups of rounds 1-2 (removed in round 3: slower; the library now groups any list by 4, so these
    stay valid orders for tools/order_sweep.py): a workgroup = one plane of a px x py patch of blocks (px * py = 8), rows
    of x-consecutive blocks one after the other, so x neighbours sit in neighbouring waves (face column through LDS) and the
    y neighbour's plane is loaded by a wave of the same workgroup (its lines are in the CU's L1 when the face row is read).
    Plane z of a patch at block-z bz goes to XCD (z + bz) % 8 as in plane_per_xcd_rot."""
    assert px * py == 8
    c = np.asarray(coords).astype(np.int64) - 1
    seqs = [[] for _ in range(N_XCD)]
    for blocks in _patches(coords, px, py, sweep):
        r = int(c[blocks[0], 2])
        for z in range(8):
            items = [(int(b) << 3) | z for b in blocks]
            while len(items) % 8:
                items.append(-1)
            for i in range(0, len(items), 8):
                seqs[(z + r) % N_XCD].append(items[i:i + 8])
    n = max(len(s2) for s2 in seqs)
    grid = np.full((n, N_XCD, 8), -1, dtype=np.int64)
    for x, s2 in enumerate(seqs):
        if len(s2):
            grid[: len(s2), x, :] = np.asarray(s2, dtype=np.int64)
    return grid.reshape(-1).astype(np.int32)


def brick(coords, px: int, py: int, pz: int, rotate: bool = True) -> np.ndarray:
    """Workgroup = px x-adjacent blocks x py y-adjacent blocks x pz consecutive planes (px * py * pz = 4 waves for the library to keep the group), waves
    ordered x fastest (same-plane x neighbours in neighbouring waves -> LDS column exchange); blocks visited in MEMORY order (bz
    fastest, then by, then bx) so that every population stream is read and written sequentially; the 8 / pz plane groups of a
    brick are consecutive workgroups, rotated with bz. Full boxes whose extents divide by px, py only (bench / tools)."""
    c = np.asarray(coords).astype(np.int64) - 1
    nb = c.max(axis=0) + 1
    lut = np.full(tuple(nb), -1, dtype=np.int64)
    lut[c[:, 0], c[:, 1], c[:, 2]] = np.arange(len(c))
    assert (lut >= 0).all() and nb[0] % px == 0 and nb[1] % py == 0
    ng = 8 // pz
    out = []
    for bx0 in range(0, nb[0], px):
        for by0 in range(0, nb[1], py):
            for bz in range(nb[2]):
                for gi in range(ng):
                    g = (gi + bz) % ng if rotate else gi
                    for z in range(g * pz, g * pz + pz):
                        for wy in range(py):
                            for wx in range(px):
                                out.append((int(lut[bx0 + wx, by0 + wy, bz]) << 3) | z)
    return np.asarray(out, dtype=np.int32)


def run_per_xcd(coords, rotate: bool = True) -> np.ndarray:
    """x-runs of 4 in the sweep x, y, z, but the 8 planes of a run all on ONE XCD (run g -> XCD g % 8), start plane rotated by the XCD:
    workgroup 64 s + 8 j + x = plane (j + x) % 8 of run 8 s + x. Full boxes with nbx % 4 == 0 only (tools)."""
    c = np.asarray(coords).astype(np.int64) - 1
    nb = c.max(axis=0) + 1
    lut = np.full(tuple(nb), -1, dtype=np.int64)
    lut[c[:, 0], c[:, 1], c[:, 2]] = np.arange(len(c))
    runs = [[int(lut[bx0 + w, by, bz]) for w in range(4)] for bz in range(nb[2]) for by in range(nb[1]) for bx0 in range(0, nb[0], 4)]
    out = []
    for s0 in range(0, len(runs), 8):
        for j in range(8):
            for x in range(8):
                z = (j + x) % 8 if rotate else j
                out += [(b << 3) | z for b in runs[s0 + x]] if s0 + x < len(runs) else [-1] * 4     # idle workgroup: keeps the XCD slots
    return np.asarray(out, dtype=np.int32)


def plane_round_robin(coords, px: int = 2, py: int = 2, sweep: str = "xyz") -> np.ndarray:
    """same workgroups as plane_per_xcd but without aiming planes at XCDs: patch after patch, all 8 planes in turn,
    for each workgroup of a large patch -> isolates the effect of workgroup grouping from XCD placement"""
    out = []
    for blocks in _patches(coords, px, py, sweep):
        groups = [blocks[i:i + WAVES] for i in range(0, len(blocks), WAVES)]
        for grp in groups:
            for z in range(8):
                out += _pad4((int(b) << 3) | z for b in grp)
    return np.asarray(out, dtype=np.int32)


def columns(coords, tx: int = 1, ty: int = 0) -> np.ndarray:
    """workgroup = 4 consecutive planes of one block; blocks swept bz-fastest (adjacent memory); the (bx,by) columns
    are visited in tx x ty tiles (ty = 0: whole y extent, i.e. the reference order when tx = 1)"""
    c = np.asarray(coords).astype(np.int64) - 1
    ty_ = ty if ty > 0 else int(c[:, 1].max()) + 1
    key = (c[:, 2], c[:, 0] % tx + 0 * c[:, 0], c[:, 1] % ty_, c[:, 0] // tx, c[:, 1] // ty_)
    # fastest first: bz, then x inside tile, y inside tile, then tile x, tile y
    order = np.lexsort((c[:, 2], c[:, 0] % tx, c[:, 1] % ty_, c[:, 0] // tx, c[:, 1] // ty_))
    z = np.arange(8, dtype=np.int64)
    return ((order[:, None] << 3) | z[None, :]).reshape(-1).astype(np.int32)


def region_per_xcd(coords, regions=(4, 2), run: int = 4, sweep: str = "yxz", planes_inner: bool = True) -> np.ndarray:
    """XCD <-> region of (bx,by) block columns: the cross-section is cut into regions[0] x regions[1] = 8 rectangles,
    XCD i sweeps region i through all bz. A workgroup = one plane of an x-run of `run` blocks; the 8 planes of a run are
    consecutive workgroups of the same XCD, so x/y face lines AND the z+-1 velocity planes are re-read on the XCD that
    fetched them; only the region perimeter crosses XCDs."""
    c = np.asarray(coords).astype(np.int64) - 1
    nbx, nby = int(c[:, 0].max()) + 1, int(c[:, 1].max()) + 1
    rx = np.minimum(c[:, 0] * regions[0] // nbx, regions[0] - 1)
    ry = np.minimum(c[:, 1] * regions[1] // nby, regions[1] - 1)
    xcd = rx * regions[1] + ry
    seqs = [[] for _ in range(N_XCD)]
    for r in range(N_XCD):
        ids = np.flatnonzero(xcd == r)
        if ids.size == 0:
            continue
        cr = c[ids]
        x0 = cr[:, 0].min()
        kx = (cr[:, 0] - x0) // run
        keys = {"yxz": (cr[:, 1], kx, cr[:, 2]), "zyx": (cr[:, 2], cr[:, 1], kx), "yzx": (cr[:, 1], cr[:, 2], kx), "xyz": (kx, cr[:, 1], cr[:, 2])}[sweep]
        order = np.lexsort((cr[:, 0],) + keys)         # fastest first: x inside the run, then the sweep keys
        gkey = np.stack([kx, cr[:, 1], cr[:, 2]], 1)[order]
        cut = np.flatnonzero(np.r_[True, (gkey[1:] != gkey[:-1]).any(1), True])
        for i in range(len(cut) - 1):
            blocks = ids[order[cut[i]:cut[i + 1]]]
            for z in range(8):
                items = _pad4((int(b) << 3) | z for b in blocks)
                for j in range(0, len(items), WAVES):
                    seqs[r].append(items[j:j + WAVES])
    return _per_xcd(seqs)


BUILDERS = {
    "block_planes": block_planes,
    "pxcd_2x2_xyz": lambda c: plane_per_xcd(c, 2, 2, "xyz"),
    "pxcd_2x2_zxy": lambda c: plane_per_xcd(c, 2, 2, "zxy"),
    "pxcd_4x1_xyz": lambda c: plane_per_xcd(c, 4, 1, "xyz"),
    "pxcd_1x4_yxz": lambda c: plane_per_xcd(c, 1, 4, "yxz"),
    "pxcd_4x4_xyz": lambda c: plane_per_xcd(c, 4, 4, "xyz"),
    "pxcd_8x8_xyz": lambda c: plane_per_xcd(c, 8, 8, "xyz"),
    "pxcd_4x4_zxy": lambda c: plane_per_xcd(c, 4, 4, "zxy"),
    "prr_2x2_xyz": lambda c: plane_round_robin(c, 2, 2, "xyz"),
    "prr_4x1_xyz": lambda c: plane_round_robin(c, 4, 1, "xyz"),
    "pxcdrot_4x1_zxy": lambda c: plane_per_xcd_rot(c, 4, 1, "zxy", "bz"),
    "pxcdrot_4x1_xzy": lambda c: plane_per_xcd_rot(c, 4, 1, "xzy", "bz"),
    "pxcdrot_4x1_xyz": lambda c: plane_per_xcd_rot(c, 4, 1, "xyz", "bz"),
    "pxcdrot2_4x1_zxy": lambda c: plane_per_xcd_rot(c, 4, 1, "zxy", "bzy"),
    "pxcd_4x1_yxz": lambda c: plane_per_xcd(c, 4, 1, "yxz"),
    "reg42_yxz": lambda c: region_per_xcd(c, (4, 2), 4, "yxz"),
    "reg24_yxz": lambda c: region_per_xcd(c, (2, 4), 4, "yxz"),
    "reg81_yxz": lambda c: region_per_xcd(c, (8, 1), 4, "yxz"),
    "reg18_yxz": lambda c: region_per_xcd(c, (1, 8), 4, "yxz"),
    "reg42_zyx": lambda c: region_per_xcd(c, (4, 2), 4, "zyx"),
    "reg42_yzx": lambda c: region_per_xcd(c, (4, 2), 4, "yzx"),
    "reg24_zyx": lambda c: region_per_xcd(c, (2, 4), 4, "zyx"),
    "pxcd_4x1_yzx": lambda c: plane_per_xcd(c, 4, 1, "yzx"),
    "pxcd_4x1_zyx": lambda c: plane_per_xcd(c, 4, 1, "zyx"),
    "pxcdrot_4x1_yxz": lambda c: plane_per_xcd_rot(c, 4, 1, "yxz", "bz"),
    "pxcdrot_4x1_yzx": lambda c: plane_per_xcd_rot(c, 4, 1, "yzx", "bz"),
    "pxcdrot_4x1_zyx": lambda c: plane_per_xcd_rot(c, 4, 1, "zyx", "bz"),
    "prr_4x1_zxy": lambda c: plane_round_robin(c, 4, 1, "zxy"),
    "pxcd_4x1_zxy": lambda c: plane_per_xcd(c, 4, 1, "zxy"),
    "pxcd_4x1_xzy": lambda c: plane_per_xcd(c, 4, 1, "xzy"),
    "brick_2x1x2": lambda c: brick(c, 2, 1, 2),
    "brick_4x1x1": lambda c: brick(c, 4, 1, 1),
    "brick_2x1x4": lambda c: brick(c, 2, 1, 4),
    "brick_4x1x2": lambda c: brick(c, 4, 1, 2),
    "brick_2x2x2": lambda c: brick(c, 2, 2, 2),
    "run_per_xcd": lambda c: run_per_xcd(c, True),
    "run_per_xcd_norot": lambda c: run_per_xcd(c, False),
    "w8_8x1_xyz": lambda c: plane_per_xcd_rot_w8(c, 8, 1, "xyz"),
    "w8_4x2_xyz": lambda c: plane_per_xcd_rot_w8(c, 4, 2, "xyz"),
    "w8_4x2_yxz": lambda c: plane_per_xcd_rot_w8(c, 4, 2, "yxz"),
    "w8_8x1_yxz": lambda c: plane_per_xcd_rot_w8(c, 8, 1, "yxz"),
    "w8_2x4_yxz": lambda c: plane_per_xcd_rot_w8(c, 2, 4, "yxz"),
    "cols_xinner": lambda c: columns(c, 1 << 20, 1),
    "cols_t44": lambda c: columns(c, 4, 4),
    "cols_t22": lambda c: columns(c, 2, 2),
    "cols_t88": lambda c: columns(c, 8, 8),
}


def build(name: str, coords) -> np.ndarray:
    return BUILDERS[name](np.asarray(coords))
