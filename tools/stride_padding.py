"""Is the 256^3 box (32768 blocks: the stride between two populations is exactly 64 MiB) slowed by its power-of-two strides?
The same box with k extra blocks appended that nothing references (n_owned = 32768, n_blocks = 32768 + k) - they only lengthen the
stride between populations. usage: stride_padding.py [k ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from open_ludwig_amd import _lib, adapt, cases
from open_ludwig_amd.blocks import BlockLevel, build_neighbor_table
from open_ludwig_amd.physics import stream_collide
nb = 32
pads = [int(v) for v in sys.argv[1:]] or [0, 1, 8, 33, 64, 512, 1000, 4097, 6536, 0, 33]
coords = cases.full_box_coords(nb, nb, nb)
table0 = build_neighbor_table(coords, nb, nb, nb, (True, True, True))
_, params = cases.periodic_box((1, 1, 1))
import dataclasses
params = dataclasses.replace(params, domain_nx=8 * nb, domain_ny=8 * nb, domain_nz=8 * nb)
for k in pads:
    extra = [(nb + 2 + i % 16, 1 + (i // 16) % 64, 1 + i // 1024) for i in range(k)]
    table = np.zeros((len(coords) + k, 27), dtype=np.int32, order="F")
    table[: len(coords)] = table0
    lvl = BlockLevel(1, coords + extra, table, 1.0, 1.0, 0.5006, enable_temporal_interpolation=False)
    lvl.n_owned = len(coords)
    cases.init_taylor_green(lvl, (8 * nb,) * 3, 0.03, share_ab_buffers=True)
    d = adapt(lvl, 0)
    t = 1
    for _ in range(40):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(150):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 150 * 1e3
    print(f"{k:5d} extra blocks (stride {4 * 512 * (len(coords) + k) / 2**20:.3f} MiB): {ms:.4f} ms per step, {16777216 / ms / 1e3:.0f} MLUPS", flush=True)
    d.close(); del lvl, d
