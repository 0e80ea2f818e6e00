"""ball1m as shipped (Re 9.87 M, N = 55, 4 levels): HIP path vs the reference's own CASES/ball1m/RESULTS/forces.csv."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from open_ludwig_amd import preprocess as pp, case
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
G = os.path.join(ROOT, "tests", "golden")
cfg = pp.load_case_configuration(os.path.join(G, "ball1m_config.yaml"), {"basic": {"num_levels": 4}})
cfg.diag_freq = 200
t = time.time(); setup = pp.setup_multilevel_domain(cfg, os.path.join(G, "ball1m.stl")); print("setup s", round(time.time() - t, 1), setup[3], flush=True)
ref = {int(l.split(",")[0]): [float(v) for v in l.split(",")[1:]] for l in open(os.path.join(G, "sphere_re10m_forces.csv")) if l[0].isdigit()}
import tempfile
out = tempfile.mkdtemp(prefix="re10m_")      # the flow VTU of 3.9 M cells is too big for gpurun_out
t = time.time(); rows, _, params = case.run_case(cfg, case.HipStepper, steps=steps, setup=setup, out_dir=out); dt = time.time() - t
upd = sum(g.n_blocks * 512 * 2 ** i for i, g in enumerate(setup[0])) * steps
print(f"HIP: {steps} steps in {dt:.1f} s incl. diagnostics -> {upd / dt / 1e6:.0f} MLUPS (true count)")
mine = {int(l.split(",")[0]): [float(v) for v in l.split(",")[1:]] for l in open(os.path.join(out, "forces.csv")) if l[0].isdigit()}
print("step   Fx hip / ref  (rel)      Cd hip / ref      Cl hip / ref     Fx_v hip / ref")
for s in sorted(mine):
    if s in ref:
        a, b = mine[s], ref[s]
        print(f"{s:5d}  {a[2]:.6e} / {b[2]:.6e} ({abs(a[2]-b[2])/abs(b[2]):.1e})   {a[10]:.6f} / {b[10]:.6f}   {a[11]:+.6f} / {b[11]:+.6f}   {a[6]:.4e} / {b[6]:.4e}")
