"""One-off: Wing_5_deg AS SHIPPED (5 levels, resolution 1100) on WORLD ranks against one device - fields of the finest and the coarsest
level and the diagnostics rows after a few coarse steps. The box has one GPU, so the ranks share it and talk over gloo with host
staging (the rehearsal transport of tests/test_case_wing.py::test_real_wing_on_ranks_equals_single_device, at the shipped size): what is
checked is the partition machinery at 151 020 blocks - level cuts, ghost and parent-data plans, the f_post_collision readers - not RCCL.
The ramp is shortened (ramp_steps 40) so that the flow has a signal after a few steps.
usage: wing_shipped_on_ranks.py [world = 2] [coarse steps = 8]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

G = os.path.join(ROOT, "tests", "golden")
OVER = {"basic": {"simulation": {"ramp_steps": 40}}}
GATHER = [(4, "rho"), (4, "vel"), (0, "rho")]


def load():
    from open_ludwig_amd import preprocess as pp
    cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), OVER)
    cfg.diag_freq = 4
    cfg.output_freq = 10 ** 9
    return cfg, pp.setup_multilevel_domain(cfg, os.path.join(G, "wing5deg_model.stl"))


def main():
    from open_ludwig_amd import case
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out = os.path.join(ROOT, "gpurun_out", "wing_ranks")
    os.makedirs(out, exist_ok=True)
    if "RANK" not in os.environ:
        cfg, setup = load()
        holder = {}

        def factory(grids):
            holder["st"] = case.HipStepper(grids)
            close = holder["st"].close
            holder["close"], holder["st"].close = close, (lambda: None)
            return holder["st"]

        t0 = time.time()
        rows, rep, _ = case.run_case(cfg, factory, steps=steps, setup=setup)
        print(f"one device: {steps} coarse steps in {time.time() - t0:.1f} s; blocks {rep.level_blocks}", flush=True)
        np.savez(os.path.join(out, "single.npz"), **{f"{n}{l}": holder["st"].field(l, n) for l, n in GATHER})
        json.dump([[r.step, r.u_lat, r.rho_min, r.cd, r.cl, r.cs, r.cmy] for r in rows], open(os.path.join(out, "single_rows.json"), "w"))
        holder["close"]()
        del holder, setup
        env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", "29517", os.path.abspath(__file__), str(world), str(steps)]
        res = subprocess.run(cmd, env=env, cwd=ROOT)
        if res.returncode != 0:
            sys.exit(res.returncode)
        want, got = np.load(os.path.join(out, "single.npz")), np.load(os.path.join(out, "ranks.npz"))
        bad = 0
        for k in want.files:
            n = int((want[k] != got[k]).sum())
            bad += n
            print(f"{k:6s}: {want[k].size:11d} values, {n} differ, max |value - rest| {float(np.abs(want[k] - (1.0 if k.startswith('rho') else 0.0)).max()):.3e}")
        r1, r2 = json.load(open(os.path.join(out, "single_rows.json"))), json.load(open(os.path.join(out, "ranks_rows.json")))
        scale = max(abs(r[3]) for r in r1)
        for a, b in zip(r1, r2):
            print("row", a[0], "one device", a[2:], "| ranks", b[2:])
            assert a[:3] == b[:3] and all(abs(x - y) <= 2e-5 * scale for x, y in zip(a[3:], b[3:])), (a, b)
        for r in range(world):
            print(f"rank {r}: [owned, local blocks, bytes per level step] per level", json.load(open(os.path.join(out, f"stats{r}.json"))))
        print("FIELDS IDENTICAL, coefficients within 2e-5 of the largest Cd (per-rank partial sums)" if bad == 0 else f"{bad} VALUES DIFFER")
        sys.exit(0 if bad == 0 else 1)
    # ---- a rank ----
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    cfg, setup = load()
    holder = {}

    def factory(grids):
        holder["st"] = case.DistributedStepper(grids, device=0, stage_through_host=True)
        holder["st"].close = lambda: None
        return holder["st"]

    t0 = time.time()
    rows, rep, _ = case.run_case(cfg, factory, steps=steps, setup=setup, write_files=False)
    st = holder["st"]
    if rank == 0:
        print(f"{world} ranks: {steps} coarse steps (plans included) in {time.time() - t0:.1f} s", flush=True)
    got = {f"{n}{l}": st.field(l, n) for l, n in GATHER}
    stats = [[v.n_owned, v.level.n_blocks, st.runner.plans[i].bytes_per_step()] for i, v in enumerate(st.runner.views)]
    json.dump(stats, open(os.path.join(out, f"stats{rank}.json"), "w"))
    if rank == 0:
        np.savez(os.path.join(out, "ranks.npz"), **got)
        json.dump([[r.step, r.u_lat, r.rho_min, r.cd, r.cl, r.cs, r.cmy] for r in rows], open(os.path.join(out, "ranks_rows.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
