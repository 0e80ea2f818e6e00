"""Regenerates the committed golden vectors:  python tests/golden/make_golden.py

Provenance: these vectors are produced by THIS repository's CPU oracle (oracle/ludwig_oracle.c), not by the
reference - the reference is Julia, which is absent from the build image, and ships no unit-level fixtures
(SURVEY.md 8c). They pin the oracle against regressions and give the GPU tests fixed, host-independent targets.
The external anchors the reference does ship (run-log series) are kept as data in sphere_re266k_log.csv; reaching
them needs the host pre-processing and force rows N1/N2, which are not built yet.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from open_ludwig_amd import cases  # noqa: E402
from oracle import oracle  # noqa: E402


def tgv16():
    grids, params = cases.periodic_box((2, 2, 2))
    steps = 10
    oracle.execute_timestep_batch(grids, 1, steps, np.float32(0.0), params)
    L = grids[0]
    fn, vn = oracle.newest_buffers(0, steps)
    return dict(steps=steps, rho=L.rho.copy(), vel=getattr(L, vn).copy(), f_sum=np.float64(getattr(L, fn).astype(np.float64).sum()))


def tunnel(levels, wall_model):
    grids, params = cases.tunnel_with_sphere((4, 3, 3), levels=levels, wall_model=wall_model, tau=0.5003 if wall_model else 0.5006, seed=11)
    steps = 3
    oracle.execute_timestep_batch(grids, 1, steps, np.float32(0.05), params)
    out = dict(steps=steps)
    for i, g in enumerate(grids):
        fn, vn = oracle.newest_buffers(i, steps)
        f = getattr(g, fn)
        out[f"l{i + 1}_rho"] = g.rho.copy()
        out[f"l{i + 1}_vel"] = getattr(g, vn).copy()
        out[f"l{i + 1}_f_sample"] = f.reshape(-1, order="F")[::997].copy()     # every 997th value of f
        out[f"l{i + 1}_f_sum"] = np.float64(f.astype(np.float64).sum())
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "tgv16_10steps.npz"), **tgv16())
    np.savez_compressed(os.path.join(HERE, "tunnel_1level_3steps.npz"), **tunnel(1, False))
    np.savez_compressed(os.path.join(HERE, "tunnel_2level_wall_3steps.npz"), **tunnel(2, True))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
