"""Time the stream-collide kernel under different launch orders in ONE process (interleaved rounds, guide rule 24)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from open_ludwig_amd import adapt, cases, order as order_mod
from open_ludwig_amd.physics import stream_collide

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(order_mod.BUILDERS)
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = 20
nb = size // 8
grids, params = cases.periodic_box((nb, nb, nb))
coords = np.asarray(grids[0].active_block_coords)
level = adapt(grids[0], 0)
del grids
stream = torch.cuda.current_stream()
level.set_stream(stream.cuda_stream)
t = 1
res = {n: [] for n in names}
orders = {n: order_mod.build(n, coords) for n in names}
for r in range(rounds):
    for n in names:
        level.set_order(orders[n])
        for _ in range(3):
            stream_collide(level, None, 0.5, 0.0, params, t); t += 1
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            stream_collide(level, None, 0.5, 0.0, params, t); t += 1
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        res[n].append(ms)
cells = size ** 3
for n in names:
    ms = np.median(res[n])
    print(f"{n:12s} ms/step {ms:8.4f}  MLUPS {cells / ms / 1e3:9.1f}  algoGB/s {216 * cells / ms / 1e6:8.1f}  frac {216 * cells / ms / 1e6 / 8000:.3f}  all {['%.3f' % v for v in res[n]]}", flush=True)
rho = level.download("rho")
print("finite", np.isfinite(rho).all(), "rho std", rho.std())
