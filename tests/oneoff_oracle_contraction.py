"""Step-200 question: the reference's logs print Cd = 0.0633 (Re 266k) and Fx = 997.81 (Re 9.87 M) at
step 200, this repository 0.0637 / 1004.04 - 0.6 % - while every later row agrees to the printed digits. The logs are CUDA
runs, whose compiler fuses a*b+c; the oracle and the HIP kernels never do. This script steps the CPU oracle on ball1m to
step N in two builds of the SAME source - contraction off (the parity build) and -ffp-contract=fast -mfma - and prints
the force rows, so the size of the effect of fusing is measured instead of asserted.
usage: tests/oneoff_oracle_contraction.py re266k|re10m [steps=200] [threads]   (run once per flavour: LUDWIG_ORACLE_LIB selects the build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from open_ludwig_amd import case, preprocess as pp
from oracle import oracle
from _steppers import OracleStepper
which = sys.argv[1] if len(sys.argv) > 1 else "re266k"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if len(sys.argv) > 3:
    oracle.set_num_threads(int(sys.argv[3]))
G = os.path.join(ROOT, "tests", "golden")
ov = {"re266k": {"basic": {"surface_resolution": 25, "flow": {"velocity": 4.0}}}, "re10m": {}}[which]
cfg = pp.load_case_configuration(os.path.join(G, "ball1m_config.yaml"), ov)
t0 = time.time()
rows, rep, params = case.run_case(cfg, OracleStepper, steps=steps, stl_path=os.path.join(G, "ball1m.stl"))
fl = os.path.basename(oracle.LIB_PATH)
for r in rows:
    print(f"{fl:32s} {which} step {r.step:5d} u_lat {r.u_lat:.6f} rho_min {r.rho_min:.6f} Cd {r.cd:.6f} Cl {r.cl:.6f}", flush=True)
print(f"# {time.time() - t0:.0f} s")
