"""ball1m (Re 266k settings) stepping rate without diagnostics: coarse steps / s and true cell updates / s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from open_ludwig_amd import preprocess as pp, case
from open_ludwig_amd.solver_control import execute_timestep_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
G = os.path.join(ROOT, "tests", "golden")
cfg = pp.load_case_configuration(os.path.join(G, "ball1m_config.yaml"), {"basic": {"surface_resolution": 25, "flow": {"velocity": 4.0}, "simulation": {"steps": 6000, "output_freq": 1000}}})
grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, os.path.join(G, "ball1m.stl"))
sp = pp.solver_params(cfg, params)
st = case.HipStepper(grids)
st.batch(1, 64, np.float32(0.02), sp)
for rep_i in range(3):
    t = time.time(); st.batch(65, steps, np.float32(0.02), sp); dt = time.time() - t
    upd = sum(g.n_blocks * 512 * 2 ** i for i, g in enumerate(grids)) * steps
    cells = sum(g.n_blocks * 512 for g in grids)
    print(f"{steps} coarse steps in {dt:.3f} s: {dt / steps * 1e3:.3f} ms/coarse step, {upd / dt / 1e6:.0f} MLUPS true count, {cells * steps / dt / 1e6:.0f} MLUPS as the reference prints it", flush=True)
