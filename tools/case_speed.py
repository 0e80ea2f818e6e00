"""Stepping rate of a ball1m variant / the real wing (3 levels) without diagnostics. usage: case_speed.py re266k|re10m|wing [coarse steps] [nowall]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from open_ludwig_amd import preprocess as pp, case
which = sys.argv[1] if len(sys.argv) > 1 else "re266k"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
G = os.path.join(ROOT, "tests", "golden")
ov = {"re266k": {"basic": {"surface_resolution": 25, "num_levels": 3, "flow": {"velocity": 4.0}}}, "re10m": {"basic": {"num_levels": 4}},
      "wing": {"basic": {"surface_resolution": 200, "num_levels": 3}}}[which]
name = "wing5deg" if which == "wing" else "ball1m"
cfg = pp.load_case_configuration(os.path.join(G, name + "_config.yaml"), ov)
grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, os.path.join(G, "wing5deg_model.stl" if which == "wing" else "ball1m.stl"))
sp = pp.solver_params(cfg, params)
if "nowall" in sys.argv[3:]:      # experiment: what the wall-model force costs (results differ, speed only)
    import dataclasses
    sp = dataclasses.replace(sp, wall_model_active=False)
st = case.HipStepper(grids)
for i, d in enumerate(st.dev):
    inf = d.info()
    print(f"level {i + 1}: {inf.n_blocks} blocks, x-run {inf.n_xrun_blocks}, other all-neighbour {inf.n_fast_blocks - inf.n_xrun_blocks}, general {inf.n_general_blocks}, Bouzidi cells {inf.n_boundary_cells}")
st.batch(1, 32, np.float32(0.02), sp)
for rep_i in range(2):
    t = time.time(); st.batch(33, steps, np.float32(0.02), sp); dt = time.time() - t
    upd = sum(g.n_blocks * 512 * 2 ** i for i, g in enumerate(grids)) * steps
    cells = sum(g.n_blocks * 512 for g in grids)
    print(f"{steps} coarse steps in {dt:.3f} s: {dt / steps * 1e3:.3f} ms/coarse step, {upd / dt / 1e6:.0f} MLUPS true count, {cells * steps / dt / 1e6:.0f} MLUPS as the reference prints it", flush=True)
