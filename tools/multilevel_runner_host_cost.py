"""Host time of partition.MultiLevelRunner (the N > 1 driver of nested levels, Python) per level step: a world of ONE rank (no peers, so
nothing but the driver's own calls), the real wing (3 levels, resolution 200; `shipped` as argument: 5 levels, resolution 1100), native
transport. Printed: coarse steps per second with the C batch driver (HipStepper), with MultiLevelRunner, and the runner's enqueue-only time.
On 8 GPUs a rank of the shipped wing has about 7 ms of GPU work per coarse step and 31 level steps: the budget is 225 us per level step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import numpy as np
import torch, torch.distributed as dist
from open_ludwig_amd import case, partition, preprocess as pp

G = os.path.join(ROOT, "tests", "golden")
shipped = "shipped" in sys.argv
cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), None if shipped else {"basic": {"surface_resolution": 200, "num_levels": 3}})
grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, os.path.join(G, "wing5deg_model.stl"))
sp = pp.solver_params(cfg, params)
n = 40 if shipped else 400
levels = len(grids)
level_steps = 2 ** levels - 1
u = np.float32(0.003)
partition.init_rccl(0)
t_c = float("nan")
if "runner_only" not in sys.argv:          # (for rocprofv3 --stats of the runner alone)
    st = case.HipStepper(grids)
    st.batch(1, 8, u, sp)
    t0 = time.perf_counter(); st.batch(9, n, u, sp); t_c = (time.perf_counter() - t0) / n
    st.close()
ds = case.DistributedStepper(grids, device=0)
ds.batch(1, 8, u, sp)
torch.cuda.synchronize()
prof = None
if "profile" in sys.argv:
    import cProfile
    prof = cProfile.Profile(); prof.enable()
t0 = time.perf_counter()
for t in range(9, 9 + n):
    ds.runner.step(t, u)
t_enq = (time.perf_counter() - t0) / n
if prof:
    import pstats
    prof.disable(); pstats.Stats(prof).sort_stats("tottime").print_stats(14)
ds.runner.synchronize()
t_py = (time.perf_counter() - t0) / n
print(f"blocks {rep.level_blocks}: {level_steps} level steps per coarse step")
print(f"C batch driver (level streams, one device):   {t_c * 1e3:8.3f} ms per coarse step")
print(f"MultiLevelRunner, world 1, transport {ds.runner.transport}: {t_py * 1e3:8.3f} ms per coarse step; host enqueue only {t_enq * 1e3:8.3f} ms = {t_enq / level_steps * 1e6:6.1f} us per level step")
ds.close()
dist.destroy_process_group()
