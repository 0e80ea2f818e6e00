"""Small driver for rocprofv3: N stream-collide launches on a size^3 periodic box (no CPU leg, no extra kernels in the loop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open_ludwig_amd import adapt, cases, order as order_mod
from open_ludwig_amd.physics import stream_collide
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
order = sys.argv[3] if len(sys.argv) > 3 else None
nb = size // 8
grids, params = cases.periodic_box((nb, nb, nb))
coords = np.asarray(grids[0].active_block_coords)
level = adapt(grids[0], 0)
if order:
    level.set_order(order_mod.build(order, coords))
for t in range(1, steps + 1):
    stream_collide(level, None, 0.5, 0.0, params, t)
level.synchronize()
print("done", np.isfinite(level.download("rho")).all())
