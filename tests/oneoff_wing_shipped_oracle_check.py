"""One-off parity check at the size the case ships at: Wing_5_deg with no overrides (5 levels, resolution 1100: 151 020 blocks, 77.3 M cells,
2.1 M Bouzidi cells, the WIDE instantiations on the finest level) stepped N coarse steps by the HIP library and by the CPU oracle from the
same state, every level's newest f, velocity and rho compared bit for bit. The inlet speed is the shipped u_lattice at once (no ramp):
after 2 coarse steps the ramp's own value would be 1e-8 and the comparison next to nothing. Start state `rest` (as the case starts) or
`uniform` (default): every level in equilibrium at rho = 1, u = (u_lattice, 0, 0) - the body, its Bouzidi links, the wall model and
the refinement interfaces meet a real flow from the first step on, where from rest the finest levels see 1e-6 in two steps (the inlet is
50 coarse cells away). Too slow for the GPU suite (the oracle needs most of a minute per coarse step on 16 cores), so it lives here and its
output is kept under profiles/.
usage: tests/oneoff_wing_shipped_oracle_check.py [coarse steps = 2] [uniform | rest]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from open_ludwig_amd import adapt, case, execute_timestep_batch, preprocess as pp
from oracle import oracle                      # test infrastructure: this tool is a checker, not the product path

G = os.path.join(ROOT, "tests", "golden")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
start = sys.argv[2] if len(sys.argv) > 2 else "uniform"
cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"))
t0 = time.time()
grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, os.path.join(G, "wing5deg_model.stl"))
sp = pp.solver_params(cfg, params)
print(f"set-up {time.time() - t0:.1f} s: blocks {rep.level_blocks}, Bouzidi cells {rep.bouzidi_cells}", flush=True)
u = np.float32(cfg.u_lattice)
oracle.build()
for g in grids:
    oracle.init_equilibrium(g)
if start == "uniform":
    w = [1.0 / 216, 1.0 / 54, 2.0 / 27, 8.0 / 27]
    for g in grids:
        for k in range(27):
            c = (k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1)
            wk = np.float32(w[3 - sum(abs(v) for v in c)])
            cu = np.float32(c[0]) * u
            feq = wk * (np.float32(1.0) + np.float32(3.0) * cu + np.float32(4.5) * cu * cu - np.float32(1.5) * u * u)
            for name in ("f", "f_temp", "f_old"):
                getattr(g, name)[..., k] = feq
        for name in ("vel", "vel_temp", "vel_old"):
            getattr(g, name)[..., 0] = u
    print(f"start state: uniform flow u = ({float(u)}, 0, 0) on every level", flush=True)
dev = [adapt(g, 0) for g in grids]                      # uploads the host state: both sides start from the same arrays
t0 = time.time()
execute_timestep_batch(dev, 1, steps, u, sp)
for d in dev:
    d.synchronize()
print(f"HIP: {steps} coarse steps in {time.time() - t0:.2f} s", flush=True)


class _Hip:
    def field(self, i, name):
        return dev[i].download(name)

    def close(self):
        for d in dev:
            d.close()


hip = _Hip()
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
