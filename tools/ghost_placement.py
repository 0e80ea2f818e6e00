"""Does it matter WHERE the ghost blocks of a brick sit in device memory? The library stores owned blocks first (sorted z, y, x) and the
ghosts behind them. Stand-in for "ghosts at their natural place": a periodic (nb+2)^3 box, all owned, whose outer shell is marked
as the boundary part - stepping the INTERIOR part alone then steps an nb^3 brick whose face neighbours lie where the sweep expects them.
Compared with one launch over rank 0's brick view of a 2x2x2 decomposition (ghosts behind) and over the plain nb^3 periodic box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from open_ludwig_amd import _lib, adapt, cases
from open_ludwig_amd.physics import stream_collide
import _rccl_loopback_worker as lw
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32

def timeit(d, params, part, n=100):
    t = 1
    for _ in range(30):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t, part=part); t += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t, part=part); t += 1
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for rep in range(2):
    grids, params = cases.periodic_box((nb, nb, nb), upload_only=True)
    d = adapt(grids[0], 0); a = timeit(d, params, _lib.PART_ALL); d.close()
    view, plan, vparams = lw.symmetric_brick_plan((2, 2, 2), nb)
    view.level.comm_boundary[:] = 0
    d = adapt(view.level, 0); b = timeit(d, vparams, _lib.PART_ALL); d.close()
    grids, params = cases.periodic_box((nb + 2, nb + 2, nb + 2), upload_only=True)
    g = grids[0]
    c = np.asarray(g.active_block_coords)
    g.comm_boundary = ((c == 1) | (c == nb + 2)).any(axis=1).astype(np.uint8)
    d = adapt(g, 0); e = timeit(d, params, _lib.PART_INTERIOR); d.close()
    print(f"{nb}^3 blocks per step: plain periodic box {a:.4f} ms | brick view, ghosts behind the owned blocks {b:.4f} ms | "
          f"brick inside a {nb + 2}^3 box (ghosts in place) {e:.4f} ms", flush=True)
