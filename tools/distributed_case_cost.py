"""The case loop (run_case: batches, ramp, diagnostics every 100 steps) on the real wing, 3 levels: one device (HipStepper, C batch driver)
against the distributed stepper with a world of one rank (DistributedStepper -> MultiLevelRunner, per-rank force sums, scalar collectives):
what the N > 1 host path costs before any message is sent. usage: distributed_case_cost.py [steps = 2000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import numpy as np
import torch.distributed as dist
from open_ludwig_amd import case, partition, preprocess as pp

G = os.path.join(ROOT, "tests", "golden")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), {"basic": {"surface_resolution": 200, "num_levels": 3}})
cfg.diag_freq = 100
partition.init_rccl(0)
out = {}
for name, factory in (("one device (HipStepper)", case.HipStepper), ("world of one rank (DistributedStepper)", lambda g: case.DistributedStepper(g, device=0))):
    setup = pp.setup_multilevel_domain(cfg, os.path.join(G, "wing5deg_model.stl"))
    prof = None
    if "profile" in sys.argv and "Distributed" in name:
        import cProfile
        prof = cProfile.Profile(); prof.enable()
    t0 = time.perf_counter()
    rows, _, _ = case.run_case(cfg, factory, steps=steps, setup=setup)
    dt = time.perf_counter() - t0
    if prof:
        import pstats
        prof.disable(); pstats.Stats(prof).sort_stats("cumulative").print_stats(28)
    out[name] = rows
    print(f"{name:40s}: {steps} coarse steps, {len(rows)} diagnostics rows in {dt:.2f} s = {dt / steps * 1e3:.3f} ms per coarse step", flush=True)
a, b = out.values()
scale = max(abs(r.cd) for r in a)
worst = max(max(abs(x.cd - y.cd), abs(x.cl - y.cl), abs(x.cmy - y.cmy)) for x, y in zip(a, b)) / scale
print(f"rho_min rows identical: {all(x.rho_min == y.rho_min for x, y in zip(a, b))}; largest coefficient difference / largest |Cd| = {worst:.2e} (per-rank pairwise sums)")
dist.destroy_process_group()
