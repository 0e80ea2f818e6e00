"""ball1m at the settings of RESULTS_SPHERE_RE1M.txt (3 levels, 14.8 m/s, 12 000 steps): HIP path vs every row of the log."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from open_ludwig_amd import preprocess as pp, case
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
G = os.path.join(ROOT, "tests", "golden")
cfg = pp.load_case_configuration(os.path.join(G, "ball1m_config.yaml"), {"basic": {"surface_resolution": 25, "num_levels": 3, "flow": {"velocity": 14.8}}})
cfg.diag_freq = 200
setup = pp.setup_multilevel_domain(cfg, os.path.join(G, "ball1m.stl"))
log = {}
for name in ("sphere_re1m_log.csv", "sphere_re1m_log_late.csv"):
    log.update({int(l.split(",")[0]): [float(v) for v in l.split(",")[1:]] for l in open(os.path.join(G, name)) if l[0].isdigit()})
t = time.time(); rows, _, _ = case.run_case(cfg, case.HipStepper, steps=steps, setup=setup); dt = time.time() - t
print(f"HIP: {steps} steps in {dt:.1f} s incl. diagnostics")
print("step   rho_min(hip/log)   Cd(hip/log)  |dCd|     Cl(hip/log)  |dCl|")
for r in rows:
    if r.step in log:
        L = log[r.step]
        print(f"{r.step:5d}  {r.rho_min:.4f}/{L[1]:.4f}   {r.cd:.4f}/{L[2]:.4f}  {abs(r.cd - L[2]):.1e}   {r.cl:+.4f}/{L[3]:+.4f}  {abs(r.cl - L[3]):.1e}")
