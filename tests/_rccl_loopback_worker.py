"""RCCL on a box with ONE GPU: the product's halo exchange over the real transport, in a single process.

RCCL refuses two ranks on one device, so every N > 1 test of this repository runs over gloo. This worker gets RCCL itself to
carry the messages: it is rank 0 of a 2x1x1 or 2x2x2 brick decomposition of a periodic box whose flow has the period of ONE brick,
so all bricks hold identical data at every step and the message peer p would send is, element for element, what rank 0 itself
holds at the same place of its own brick. Every peer of the plan is therefore mapped to rank 0 (HaloExchanger.wire_rank) and the
messages go rank 0 -> rank 0 through `batch_isend_irecv` on the `nccl` backend: communicator creation, the grouped send/recv on
RCCL's stream, its ordering against the comm / compute streams of DistributedLevelRunner (interior blocks step under the exchange)
and the pack / unpack kernels are the production code. What it cannot show: a message crossing xGMI.
The result must equal a single-device run of the one-brick periodic box, bit for bit.

Also runs, once each on device tensors, the collectives bench.py and case.DistributedStepper use (world size 1).

usage (env RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=...): _rccl_loopback_worker.py 2x2x2 <blocks per brick edge> <steps> <out.json>
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


SPHERE_RADIUS = 5.3      # cells; centred on the brick corners, so every cut runs through a body


def add_corner_spheres(level, brick_cells, n_owned=None):
    """one sphere per brick, centred on the brick's corner: Bouzidi cells on both sides of every cut, links with q < 1/2 that read
    f_post_collision across it - and still the period of one brick"""
    from open_ludwig_amd import cases
    cases.add_sphere(level, (0.0, 0.0, 0.0), SPHERE_RADIUS, bouzidi=True, wall_dist=False, period=brick_cells, list_blocks=n_owned)
    level.force_post_collision = True          # a peer's Bouzidi cells read this rank's face layer


def symmetric_brick_plan(grid, nb, upload_only=True, spheres=False):
    """Rank 0's view of a periodic box of grid[0] x grid[1] x grid[2] bricks of nb^3 blocks, Taylor-Green flow with the period of
    one brick, and a halo plan whose send lists are what the PEERS would send - read from the same places of rank 0's own brick.
    spheres: a body on every brick corner (add_corner_spheres): the f_post_collision halo of the Bouzidi correction joins the plan."""
    from open_ludwig_amd import cases, partition
    from open_ludwig_amd.physics import SolverParams
    nb3 = (nb, nb, nb) if isinstance(nb, int) else tuple(nb)          # blocks per brick edge, or per axis
    nbg = tuple(nb3[i] * grid[i] for i in range(3))
    coords, table, owner = partition.periodic_box_topology(nbg, grid)
    widen = os.environ.get("LUDWIG_WIDEN_X_RUNS", "1") != "0"
    view = partition.build_local_level(1, coords, table, owner, 0, 0.5006, widen_x_runs=widen)
    cases.init_taylor_green(view.level, tuple(8 * n for n in nb3), 0.03, share_ab_buffers=upload_only)      # period = one brick
    if spheres:
        add_corner_spheres(view.level, tuple(8 * n for n in nb3), view.n_owned)
    params = SolverParams(domain_nx=8 * nbg[0], domain_ny=8 * nbg[1], domain_nz=8 * nbg[2], wall_model_active=False, c_wale=0.5,
                          nu_sgs_bg=0.0005, inlet_turbulence=0.0, use_temporal_interp=False, sponge_blend_dist=False)
    n_global = len(coords)
    mine = partition.make_requests(view, n_global)

    def in_my_brick(goff: np.ndarray, p: int) -> np.ndarray:
        """global element offsets inside brick p -> the same places of brick 0 (owner = (ix * g1 + iy) * g2 + iz)"""
        shift = (p // (grid[1] * grid[2]) * nb3[0], (p // grid[2]) % grid[1] * nb3[1], p % grid[2] * nb3[2])
        comp, rem = np.divmod(goff, n_global * 512)
        gblk, cell = np.divmod(rem, 512)
        bx, r2 = np.divmod(gblk, nbg[1] * nbg[2])
        by, bz = np.divmod(r2, nbg[2])
        g0 = (((bx - shift[0]) % nbg[0]) * nbg[1] + (by - shift[1]) % nbg[1]) * nbg[2] + (bz - shift[2]) % nbg[2]
        return (comp * n_global + g0) * 512 + cell

    # what peer p would ask of me is replaced by "what peer p would SEND me", read from my own brick, in my receive order
    to_me = {p: {name: in_my_brick(a, p) for name, a in req.items()} for p, req in mine.items()}
    return view, partition.build_plan(view, n_global, mine, to_me), params


def main():
    grid = tuple(int(v) for v in sys.argv[1].split("x"))
    nb = int(sys.argv[2]) if "," not in sys.argv[2] else tuple(int(v) for v in sys.argv[2].split(","))
    steps, out_path = int(sys.argv[3]), sys.argv[4]
    opts = set(sys.argv[5:])
    compare, spheres = "nocompare" not in opts, "spheres" in opts
    torch.cuda.set_device(0)
    from open_ludwig_amd import partition as _p
    if os.environ.get("LOOPBACK_PLAIN_INIT"):      # the trace that showed the queue collision
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        _p.init_rccl(0)
    rep = {"backend": dist.get_backend(), "world": dist.get_world_size(), "grid": list(grid), "blocks_per_brick_edge": nb, "steps": steps}

    # the collectives of bench.py (all_reduce MAX, barrier) and case.DistributedStepper (all_reduce MIN, all_gather, gather_object)
    w = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    m = torch.tensor([0.25], dtype=torch.float32, device="cuda")
    dist.all_reduce(m, op=dist.ReduceOp.MIN)
    parts = [torch.empty(10, dtype=torch.float32, device="cuda")]
    dist.all_gather(parts, torch.arange(10, dtype=torch.float32, device="cuda"))
    objs = [None]
    dist.all_gather_object(objs, {"a": np.arange(3)})
    gathered = [None]
    dist.gather_object({"b": 1}, gathered, dst=0)
    dist.barrier()
    torch.cuda.synchronize()
    rep["collectives_ok"] = bool(w.tolist() == [1.5, 2.5] and m.item() == 0.25 and parts[0].tolist() == list(range(10))
                                 and objs[0]["a"].tolist() == [0, 1, 2] and gathered[0] == {"b": 1})

    from open_ludwig_amd import adapt, cases, partition
    from open_ludwig_amd.physics import stream_collide
    view, plan, params = symmetric_brick_plan(grid, nb, spheres=spheres)
    # LOOPBACK_TRANSPORT=native (default): the library calls RCCL itself (ludwig_step_distributed; messages to this rank itself are
    # forced through ncclSend / ncclRecv: LUDWIG_HALO_SELF_VIA_RCCL); torch: round 2's path, batch_isend_irecv from Python
    transport = os.environ.get("LOOPBACK_TRANSPORT", "native")
    if transport == "native":
        os.environ["LUDWIG_HALO_SELF_VIA_RCCL"] = "1"
    rep["transport"] = transport
    runner = partition.DistributedLevelRunner(view, plan, params, 0, overlap=True, transport=transport, wire_rank={p: 0 for p in plan.peers})
    skip = set(filter(None, os.environ.get("LOOPBACK_SKIP", "").split(",")))      # timing-only diagnostics: which piece costs what
    if skip:
        assert not compare, "LOOPBACK_SKIP leaves the ghosts wrong: timing only (nocompare)"
        assert transport == "torch", "LOOPBACK_SKIP takes pieces out of the Python-driven exchange"
        if "pack" in skip:
            runner.ex.ex.pack = lambda *a: None
        if "unpack" in skip:
            runner.ex.ex.unpack = lambda *a: None
        if "transfer" in skip:
            class _Done:
                def wait(self):
                    pass
            dist.batch_isend_irecv = lambda ops: [_Done()]
        rep["skipped"] = sorted(skip)
    rep["peers"] = len(plan.peers)
    rep["bouzidi_cells"] = int(view.level.n_boundary_cells)
    rep["f_post_halo_elements"] = int(sum(plan.recv[p]["f_post"].size for p in plan.peers))
    rep["view_blocks"] = int(view.level.n_blocks)
    rep["halo_bytes_per_step"] = plan.bytes_per_step()
    # ghosts of the start state: sin(x + one period) is not bit-equal to sin(x) in floating point, so fetch them the same way
    runner.exchange_now("f", "vel")
    runner.exchange_now("f_temp", "vel_temp")
    runner.synchronize()
    warm = min(10, steps // 2)          # the first step pays for communicator set-up and code loading (12 ms)
    if not compare:
        # timing runs: the same device pre-heat as bench.py (plain device-to-device copies; the step time shows a power-management
        # transient of ~7 % for the first tens of milliseconds after idle) and a longer warm-up
        a = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        b = torch.empty_like(a)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.1:
            for _ in range(20):
                b.copy_(a)
            torch.cuda.synchronize()
        del a, b
        warm = min(40, steps // 2)
    for t in range(1, warm + 1):
        runner.step(t)
    runner.synchronize()
    runner.ex.timing = True
    t0 = time.perf_counter()
    for t in range(warm + 1, steps + 1):
        runner.step(t)
    rep["host_us_per_step_enqueue"] = (time.perf_counter() - t0) / (steps - warm) * 1e6      # host time to queue a whole step (launches + exchange)
    runner.synchronize()
    rep["ms_per_step_wall"] = (time.perf_counter() - t0) / (steps - warm) * 1e3
    rep["compute_units_left_to_the_exchange"] = runner.reserved_cus
    ms = runner.ex.exchange_ms()
    rep["exchange_ms_first"] = ms[0]
    rep["exchange_ms_median_after_first"] = float(np.median(ms[1:])) if len(ms) > 1 else None
    rep["exchange_ms_max_after_first"] = float(np.max(ms[1:])) if len(ms) > 1 else None

    if compare:
        fn, vn = ("f_temp", "vel_temp") if steps % 2 == 0 else ("f", "vel")
        got = {n: runner.level.download(n)[:, :, :, : view.n_owned] for n in (fn, vn, "rho")}
        own_coords = [tuple(c) for c in np.asarray(view.level.active_block_coords)[: view.n_owned]]
        grids, params1 = cases.periodic_box((nb, nb, nb) if isinstance(nb, int) else nb, upload_only=True)
        pos = {tuple(c): i for i, c in enumerate(grids[0].active_block_coords)}
        sel = np.array([pos[c] for c in own_coords])
        if spheres:
            add_corner_spheres(grids[0], tuple(8 * n for n in ((nb,) * 3 if isinstance(nb, int) else nb)))
        single = adapt(grids[0], 0)
        single.set_stream(torch.cuda.current_stream().cuda_stream)
        from open_ludwig_amd.physics import perform_timestep_v2
        for t in range(1, steps + 1):
            perform_timestep_v2(single, None, np.float32(0.5), np.float32(0.0), params1, t)      # collision + Bouzidi correction
        torch.cuda.synchronize()
        rep["identical"] = {n: bool(np.array_equal(got[n], single.download(n)[:, :, :, sel])) for n in (fn, vn, "rho")}
        rep["moved"] = bool(got[vn].std() > 0)
        single.close()
    json.dump(rep, open(out_path, "w"))
    print(json.dumps(rep), flush=True)
    runner.close(dist)          # device idle -> wrappers dropped -> process group destroyed -> level freed -> masked stream destroyed


if __name__ == "__main__":
    main()
