"""GPU check of the x-run kernel: bit-exact vs oracle on a periodic box under x-run orders, then timing at 256^3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open_ludwig_amd import adapt, cases, order as order_mod, execute_timestep_batch
from oracle import oracle
for nb in ((8, 4, 4), (4, 4, 4), (12, 2, 3)):
    grids, params = cases.periodic_box(nb)
    coords = np.asarray(grids[0].active_block_coords)
    for oname in ("pxcd_4x1_xyz", "prr_4x1_xyz", "block_planes"):
        g2, _ = cases.periodic_box(nb)
        d = adapt(g2[0], 0)
        d.set_order(order_mod.build(oname, coords))
        execute_timestep_batch([d], 1, 5, np.float32(0.0), params)
        g3, _ = cases.periodic_box(nb)
        oracle.execute_timestep_batch(g3, 1, 5, np.float32(0.0), params)
        ok = all(np.array_equal(d.download(n), getattr(g3[0], n)) for n in ("f", "vel", "rho"))
        print(nb, oname, "bit-exact" if ok else "MISMATCH", flush=True)
        if not ok:
            a, b = d.download("f"), g3[0].f
            bad = np.argwhere(a != b)
            print("  first bad", bad[:5], a[tuple(bad[0])], b[tuple(bad[0])], "n bad", len(bad))
        d.close()
