"""Per-cell stepping rate of the REAL kernel against the block count of the level. With the reference's population-major arrays the
distance between two populations is n_blocks x 2 KiB, and the step lost 5-30 % at some distances (64.5, 65.75, 68, 76-77 ... MiB:
profiles/r02_population_stride_sweep.txt, profiles/r03_stride_sweep_emulation.txt). The device arrays are block-major since round 3:
the rate must no longer depend on n_blocks. usage: stride_padding.py [NBXxNBYxNBZ ...]   (periodic boxes, blocks per axis)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from open_ludwig_amd import adapt, cases
from open_ludwig_amd.physics import stream_collide

# 32768 = the headline box; 33024 = 64.5 MiB, 33792 = 66, 34816 = 68, 39424 = 77, 42496 = 83, 46080 = 90 MiB (the old bad distances)
boxes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or \
        [(32, 32, 32), (32, 24, 43), (32, 32, 33), (32, 32, 34), (32, 28, 44), (32, 32, 32), (32, 32, 41), (32, 36, 40), (32, 32, 32)]
for nb in boxes:
    grids, params = cases.periodic_box(nb, upload_only=True)
    cells = grids[0].n_blocks * 512
    d = adapt(grids[0], 0)
    del grids
    t = 1
    for _ in range(40):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(150):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 150 * 1e3
    print(f"{nb[0]}x{nb[1]}x{nb[2]} = {cells // 512} blocks ({cells // 512 / 512:.3f} MiB per population in the reference layout): {ms:.4f} ms per step, "
          f"{cells / ms / 1e3:.0f} MLUPS", flush=True)
    d.close(); del d
