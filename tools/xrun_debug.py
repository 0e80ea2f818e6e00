import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open_ludwig_amd import adapt, cases, order as order_mod, execute_timestep_batch
from oracle import oracle
for nb in ((4, 4, 4), (12, 2, 3), (8, 4, 4), (4, 1, 1), (8, 1, 1)):
    grids, params = cases.periodic_box(nb)
    cases.init_perturbed(grids[0], 3)
    coords = np.asarray(grids[0].active_block_coords)
    import copy
    g3 = copy.deepcopy(grids)
    d = adapt(grids[0], 0)
    d.set_order(order_mod.build("prr_4x1_xyz", coords))
    execute_timestep_batch([d], 1, 1, np.float32(0.0), params)
    oracle.execute_timestep_batch(g3, 1, 1, np.float32(0.0), params)
    for n in ("rho", "vel", "f"):
        a, b = d.download(n), getattr(g3[0], n)
        bad = np.argwhere(a != b)
        print(nb, n, "bad", len(bad), "of", a.size, end=" | ")
        if len(bad):
            print("x", np.bincount(bad[:, 0], minlength=8), "y", np.bincount(bad[:, 1], minlength=8), "z", np.bincount(bad[:, 2], minlength=8),
                  "bx", np.bincount(coords[bad[:, 3], 0] - 1, minlength=nb[0]), "maxrel", float(np.abs(a - b).max() / np.abs(b).max()))
        else:
            print()
    d.close()
