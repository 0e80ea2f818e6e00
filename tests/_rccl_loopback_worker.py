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


def nested_symmetric_bricks(grid, nb):
    """Two nested levels on a periodic box of grid[0] x 1 x 1 bricks of nb^3 level-1 blocks, every brick the same: the refined region is
    the two layers of level-1 blocks on either side of every x-cut plane (whole y-z slabs), so every cut runs through level 2 -
    same-level ghosts on both levels, exchanged in MultiLevelRunner's order around the part launches of a parent and a finest level.
    (Parent-data ghosts do not occur: the slabs' faces lie two level-1 blocks inside the bricks.) Only x is cut, and the slabs are
    whole in y and z, because the reference's domain-edge chain
    (src/physics_kernels.jl:88-140) works on GLOBAL coordinates without wrap: a link whose source block is missing AND whose source
    cell lies across the global edge takes the edge condition, and that must be the same links in the one-brick box as in the
    G-brick box - with x-slabs no missing-source link comes near an edge in x, and the y / z extents of the two boxes are equal.
    Returns the GLOBAL host levels, the per-level owners (brick of a block), the step parameters and, per level, the function that
    maps the global element offsets a peer p would SEND to the same places of brick 0 (everything is periodic with one brick)."""
    from open_ludwig_amd import cases
    from open_ludwig_amd.physics import SolverParams
    nb3 = (nb, nb, nb) if isinstance(nb, int) else tuple(nb)
    assert grid[1] == 1 and grid[2] == 1, "the nested loop-back cuts x only (see the docstring)"
    nbg = tuple(nb3[i] * grid[i] for i in range(3))
    tau1 = 0.5006
    l1 = cases.make_level(1, cases.full_box_coords(*nbg), nbg, tau1, periodic=(True, True, True), temporal=True)
    near = lambda b, n: (b - 1) % n in (0, n - 1)              # level-1 block next to a cut plane
    c2 = []
    for (bx, by, bz) in l1.active_block_coords:
        if near(bx, nb3[0]):
            c2 += [(2 * bx - 1 + (d & 1), 2 * by - 1 + ((d >> 1) & 1), 2 * bz - 1 + ((d >> 2) & 1)) for d in range(8)]
    l2 = cases.make_level(2, c2, tuple(2 * n for n in nbg), 0.5 + (tau1 - 0.5) / 2, periodic=(True, True, True), temporal=True)
    cases.init_taylor_green(l1, tuple(8 * n for n in nb3), 0.03)        # period = one brick, on both levels
    cases.init_taylor_green(l2, tuple(16 * n for n in nb3), 0.03)
    grids = [l1, l2]
    owners = []
    for i, g in enumerate(grids):
        c = (np.asarray(g.active_block_coords, dtype=np.int64) - 1) >> i           # level-1 block that contains it
        owners.append(((c[:, 0] // nb3[0]) * grid[1] + c[:, 1] // nb3[1]) * grid[2] + c[:, 2] // nb3[2])
    params = SolverParams(domain_nx=8 * nbg[0], domain_ny=8 * nbg[1], domain_nz=8 * nbg[2], wall_model_active=False, c_wale=0.5,
                          nu_sgs_bg=0.0005, inlet_turbulence=0.0, use_temporal_interp=True, sponge_blend_dist=False)

    def mapper(i):
        g = grids[i]
        coords = np.asarray(g.active_block_coords, dtype=np.int64)
        dims = np.array(nbg) << i
        lut = {tuple(c): j for j, c in enumerate(coords)}
        n = g.n_blocks

        def in_my_brick(goff, p):
            shift = np.array([p // (grid[1] * grid[2]) * nb3[0], (p // grid[2]) % grid[1] * nb3[1], p % grid[2] * nb3[2]]) << i
            comp, rem = np.divmod(goff, n * 512)
            gblk, cell = np.divmod(rem, 512)
            c0 = (coords[gblk] - 1 - shift) % dims + 1
            g0 = np.array([lut[tuple(c)] for c in c0], dtype=np.int64)
            return (comp * n + g0) * 512 + cell
        return in_my_brick
    return grids, owners, params, [mapper(0), mapper(1)]


def nested_main(grid, nb, steps, out_path, transport):
    """2-level loop-back: MultiLevelRunner as rank 0 of the brick decomposition, every peer wired to rank 0, against the single-device
    run of the one-brick box with the same refined corners."""
    from open_ludwig_amd import adapt, cases, execute_timestep_batch, partition
    from oracle import oracle as _o          # buffer-parity helper only
    from open_ludwig_amd.physics import SolverParams
    grids, owners, params, maps = nested_symmetric_bricks(grid, nb)
    if transport == "native":
        os.environ["LUDWIG_HALO_SELF_VIA_RCCL"] = "1"
    world_bricks = grid[0] * grid[1] * grid[2]

    def to_me(i, mine):
        return {p: {name: maps[i](a, p) for name, a in req.items()} for p, req in mine.items()}

    wire = [{p: 0 for p in range(world_bricks)} for _ in grids]
    runner = partition.MultiLevelRunner(grids, owners, params, 0, 1, 0, transport=transport, wire_ranks=wire, requests_to_me=to_me)
    rep = {"backend": dist.get_backend(), "transport": transport, "nested": True, "grid": list(grid),
           "blocks": [int(g.n_blocks) for g in grids], "owned": [int(v.n_owned) for v in runner.views],
           "halo_bytes_per_level": [int(pl.bytes_per_step()) for pl in runner.plans],
           "parent_data_in_level1_halo": bool(runner.plans[0].has("rho")),
           "exchange_in_stream": [bool(getattr(ex, "in_stream", False)) for ex in runner.ex]}
    # ghosts of the start state, fetched the way every later one is
    for i, ex in enumerate(runner.ex):
        for fields in ({"f": "f", "vel": "vel"}, {"f": "f_temp", "vel": "vel_temp"}):
            if runner.plans[i].has("rho"):
                fields = dict(fields, rho="rho")
            ex.post(fields); ex.join()
    runner.synchronize()
    t0 = time.perf_counter()
    for t in range(1, steps + 1):
        runner.step(t)
    runner.synchronize()
    rep["ms_per_coarse_step_wall"] = (time.perf_counter() - t0) / steps * 1e3
    nb3 = (nb, nb, nb) if isinstance(nb, int) else tuple(nb)
    one, _, params1, _ = nested_symmetric_bricks((1, 1, 1), nb3)
    dev = [adapt(g, 0) for g in one]
    execute_timestep_batch(dev, 1, steps, np.float32(0.0), params1)
    same = {}
    for i, (lv, v) in enumerate(zip(runner.levels, runner.views)):
        fn, vn = _o.newest_buffers(i, steps)
        pos = {tuple(c): j for j, c in enumerate(one[i].active_block_coords)}
        sel = np.array([pos[tuple(c)] for c in np.asarray(v.level.active_block_coords)[: v.n_owned]])
        for name in (fn, vn, "rho"):
            same[f"level{i + 1}_{name}"] = bool(np.array_equal(lv.download(name)[:, :, :, : v.n_owned], dev[i].download(name)[:, :, :, sel]))
        rep[f"moved{i + 1}"] = bool(lv.download(vn)[:, :, :, : v.n_owned].std() > 0)
    rep["identical"] = same
    for d in dev:
        d.close()
    json.dump(rep, open(out_path, "w"))
    print(json.dumps(rep), flush=True)
    runner.close(dist)


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
    if "nested" in opts:
        return nested_main(grid, nb, steps, out_path, os.environ.get("LOOPBACK_TRANSPORT", "native"))

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
