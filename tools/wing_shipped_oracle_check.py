"""One-off parity check at the size the case ships at: Wing_5_deg with no overrides (5 levels, resolution 1100: 151 020 blocks, 77.3 M cells,
2.1 M Bouzidi cells, the WIDE instantiations on the finest level) stepped N coarse steps by the HIP library and by the CPU oracle from the
same rest state, every level's newest f, velocity and rho compared bit for bit. The inlet speed is the shipped u_lattice at once (no ramp):
after 2 coarse steps the ramp's own value would be 1e-8 and the comparison next to nothing. Too slow for the GPU suite (the oracle needs
about half a minute per coarse step on 16 cores), so it lives here and its output is kept under profiles/.
usage: wing_shipped_oracle_check.py [coarse steps = 2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from open_ludwig_amd import case, preprocess as pp
from oracle import oracle                      # test infrastructure: this tool is a checker, not the product path

G = os.path.join(ROOT, "tests", "golden")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"))
t0 = time.time()
grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, os.path.join(G, "wing5deg_model.stl"))
sp = pp.solver_params(cfg, params)
print(f"set-up {time.time() - t0:.1f} s: blocks {rep.level_blocks}, Bouzidi cells {rep.bouzidi_cells}", flush=True)
u = np.float32(cfg.u_lattice)
hip = case.HipStepper(grids)
t0 = time.time()
hip.batch(1, steps, u, sp)
for d in hip.dev:
    d.synchronize()
print(f"HIP: {steps} coarse steps in {time.time() - t0:.2f} s", flush=True)
oracle.build()
for g in grids:
    oracle.init_equilibrium(g)
t0 = time.time()
oracle.execute_timestep_batch(grids, 1, steps, u, sp)
print(f"oracle: {steps} coarse steps in {time.time() - t0:.1f} s", flush=True)
bad = 0
for i, g in enumerate(grids):
    fn, vn = oracle.newest_buffers(i, steps)
    for name in (fn, vn, "rho"):
        a, b = hip.field(i, name), getattr(g, name)
        n = int((a != b).sum())
        bad += n
        print(f"level {i + 1} {name:8s}: {a.size:12d} values, {n} differ, max |oracle| {float(np.abs(b).max()):.6g}, "
              f"max |oracle - rest| {float(np.abs(b - (1.0 if name == 'rho' else 0.0)).max()) if name != fn else float('nan'):.3e}", flush=True)
hip.close()
print("IDENTICAL" if bad == 0 else f"{bad} VALUES DIFFER")
sys.exit(0 if bad == 0 else 1)
