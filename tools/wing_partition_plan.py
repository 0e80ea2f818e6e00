"""Plan-only view of Wing_5_deg AS SHIPPED (5 levels, resolution 1100) cut over N ranks, on the CPU: per level and rank the owned
blocks, the ghost blocks (same-level halo + parent data of finer blocks) and the bytes each rank sends per coarse step - what
`MultiLevelRunner` would set up on N MI355X (BASELINE configs[4] names 8). No device, no process group: the requests every rank would
send are computed here for all ranks and inverted.
usage: wing_partition_plan.py [world=8] [3level]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from open_ludwig_amd import partition as pt, preprocess as pp

G = os.path.join(ROOT, "tests", "golden")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
over = {"basic": {"surface_resolution": 200, "num_levels": 3}} if "3level" in sys.argv else None
cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), over)
t0 = time.time()
grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, os.path.join(G, "wing5deg_model.stl"))
sp = pp.solver_params(cfg, params)
print(f"set-up {time.time() - t0:.1f} s, blocks {rep.level_blocks}", flush=True)
dims = (sp.domain_nx, sp.domain_ny, sp.domain_nz)
t0 = time.time()
owners = pt.level_owners(grids, world)
nl = len(grids)
views = [[None] * nl for _ in range(world)]
mine = [[None] * nl for _ in range(world)]
for r in range(world):
    for i in range(nl - 1, -1, -1):
        g = grids[i]
        extra = None
        if i + 1 < nl:
            extra = pt.required_parent_blocks(grids[i + 1], np.flatnonzero(np.asarray(owners[i + 1]) == r), g)
        views[r][i] = pt.build_local_level(g.level_id, g.active_block_coords, g.neighbor_table, owners[i], r, float(g.tau),
                                          temporal=g.f_old.size > 27, extra_ghosts=extra)
    for i in range(nl):
        v, g = views[r][i], grids[i]
        if g.n_boundary_cells > 0:           # what slice_level_fields would have set: the Bouzidi cells decide the f_post needs
            pt.slice_level_fields(v, g)
        needs = pt.compute_needs(v) if v.n_owned > 0 else {"f": np.zeros(0, np.int64), "vel": np.zeros(0, np.int64)}
        needs.setdefault("f_post", np.zeros(0, np.int64)); needs.setdefault("rho", np.zeros(0, np.int64))
        if i + 1 < nl and views[r][i + 1].n_owned > 0:
            extra = pt.interpolation_needs(views[r][i + 1], v, dims)
            for name in ("f", "rho", "vel"):
                needs[name] = np.unique(np.concatenate([needs[name], extra[name]]))
        mine[r][i] = pt.make_requests(v, g.n_blocks, needs)
    print(f"rank {r}: views + requests {time.time() - t0:.1f} s", flush=True)
print(f"{'level':>5} {'rank':>4} {'owned':>8} {'ghost':>8} {'peers':>5} {'sent MB / level step':>20} {'x sub-steps = MB / coarse step':>30}")
tot = np.zeros(world)
for i in range(nl):
    for r in range(world):
        v = views[r][i]
        sent = 0
        peers = set()
        for q in range(world):
            req = mine[q][i].get(r) if q != r else None          # what rank q asks of rank r
            if req:
                n = sum(len(a) for a in req.values())
                if n:
                    peers.add(q)
                    sent += 4 * n
        mb = sent / 1e6
        tot[r] += mb * 2 ** i
        print(f"{i + 1:5d} {r:4d} {v.n_owned:8d} {v.level.n_blocks - v.n_owned:8d} {len(peers):5d} {mb:20.2f} {mb * 2 ** i:30.2f}")
work = np.array([[views[r][i].n_owned * 2 ** i for i in range(nl)] for r in range(world)]).sum(axis=1)
print("cell updates per coarse step and rank (M):", np.round(work * 512 / 1e6, 1).tolist(), " max / mean = %.3f" % (work.max() / work.mean()))
print("MB sent per coarse step and rank:", np.round(tot, 1).tolist())
