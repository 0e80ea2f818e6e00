"""BASELINE configs[4] as stated: Wing_5_deg on its own STL (tests/golden/wing5deg_model.stl, 63 196 triangles), 3 levels,
surface_resolution 200, the SHIPPED run length - steps 10 000, ramp_steps 2 000 (CASES/Wing_5_deg/config.yaml:78-79 of the reference) -
through case.run_case on the HIP path. Writes the Cd / Cl / Cs / Cmy / rho_min series and a convergence summary (mean and standard
deviation over the last 2 000 steps). parity unpinned: the reference holds no log or result file for this case.
usage: run_wing.py [out_prefix] [steps] [diag_freq] [shipped]     (diag_freq only decides how often a row is taken - the shipped 500 gives 20)
`shipped` as fourth argument: NO overrides at all - num_levels 5, surface_resolution 1100 (151 020 blocks, 77.3 M cells, 1.01 G cell
updates per coarse step), which the native set-up library makes practical (open_ludwig_amd/csrc/setup_host.cpp: under a minute)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from open_ludwig_amd import case, preprocess as pp

G = os.path.join(ROOT, "tests", "golden")
OVERRIDES = {"basic": {"surface_resolution": 200, "num_levels": 3}}


def run(steps=None, diag_freq=100, log=None, shipped=False):
    over = None if shipped else OVERRIDES
    if os.environ.get("WING_OVERRIDES"):          # experiments only, e.g. '{"advanced": {"boundary": {"method": "bounce_back"}}}'
        over = json.loads(os.environ["WING_OVERRIDES"])
    cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), over)
    assert (cfg.steps, cfg.ramp_steps) == (10000, 2000), "the shipped run length"
    cfg.diag_freq = diag_freq
    t0 = time.time()
    setup = pp.setup_multilevel_domain(cfg, os.path.join(G, "wing5deg_model.stl"))
    t_setup = time.time() - t0
    if log:
        log(f"set-up {t_setup:.1f} s: blocks {setup[3].level_blocks}, Bouzidi cells {setup[3].bouzidi_cells}, tau {[float(g.tau) for g in setup[0]]}")
    t0 = time.time()
    stamps = []

    def stamped(line):
        stamps.append(time.time())
        if log:
            log(line)

    rows, rep, params = case.run_case(cfg, case.HipStepper, steps=steps, setup=setup, log=stamped)
    t_run = time.time() - t0
    total = steps or cfg.steps
    work = sum(g.n_blocks * 512 * 2 ** i for i, g in enumerate(setup[0]))
    a = np.array([[r.step, r.u_lat, r.rho_min, r.cd, r.cl, r.cs, r.cmy] for r in rows], dtype=np.float64)
    tail = a[a[:, 0] > total - 2000] if total > 2000 else a
    summary = {
        "case": ("Wing_5_deg exactly as shipped: tests/golden/wing5deg_model.stl, num_levels 5, surface_resolution 1100" if shipped else
                 "Wing_5_deg, tests/golden/wing5deg_model.stl, num_levels 3, surface_resolution 200, everything else as shipped"),
        "steps": total, "ramp_steps": cfg.ramp_steps, "diag_freq": diag_freq, "rows": len(rows),
        "level_blocks": rep.level_blocks, "bouzidi_cells": rep.bouzidi_cells, "cells": int(sum(g.n_blocks for g in setup[0]) * 512),
        "cell_updates_per_coarse_step": int(work), "setup_s": round(t_setup, 1), "run_s": round(t_run, 1),
        "ms_per_coarse_step_incl_diagnostics": round(t_run / total * 1e3, 4),
        # between diagnostics rows, the first interval (device set-up, first-launch costs) left out
        "ms_per_coarse_step_steady": (round((stamps[-1] - stamps[0]) / ((len(stamps) - 1) * diag_freq) * 1e3, 4) if len(stamps) > 1 else None),
        "glups_true_count_steady": (round(work * (len(stamps) - 1) * diag_freq / (stamps[-1] - stamps[0]) / 1e9, 3) if len(stamps) > 1 else None),
        "glups_true_count_incl_diagnostics": round(work * total / t_run / 1e9, 3),
        "frac_of_8TBs_at_216B_incl_diagnostics": round(work * total / t_run * 216 / 8e12, 4),
        "all_finite": bool(np.isfinite(a).all()), "rho_min_over_run": float(a[:, 2].min()), "rho_min_last_row": float(a[-1, 2]),
        "last_2000_steps": {name: {"mean": float(tail[:, c].mean()), "std": float(tail[:, c].std()), "min": float(tail[:, c].min()), "max": float(tail[:, c].max())}
                            for name, c in (("Cd", 3), ("Cl", 4), ("Cs", 5), ("Cmy", 6))},
        "parity": "unpinned: the reference holds no wing log; HIP = oracle bit for bit over the first 200 coarse steps of the 3-level variant (tests/test_case_wing.py)",
    }
    return a, summary


if __name__ == "__main__":
    prefix = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "wing5deg")
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    freq = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    shipped = len(sys.argv) > 4 and sys.argv[4] == "shipped"
    a, summary = run(steps, freq, log=lambda s: print(s, flush=True), shipped=shipped)
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    np.savetxt(prefix + "_series.csv", a, delimiter=",", header="step,u_lat,rho_min,Cd,Cl,Cs,Cmy", comments="", fmt=["%d", "%.6f", "%.6f", "%.6e", "%.6e", "%.6e", "%.6e"])
    json.dump(summary, open(prefix + "_summary.json", "w"), indent=1)
    print(json.dumps(summary, indent=1))
