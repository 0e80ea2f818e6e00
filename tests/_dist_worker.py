"""Worker for the multi-process halo-exchange tests (launched by torch.distributed.run, one process per rank).

mode cpu : gloo, CPU tensors; stepping by the CPU oracle, pack/unpack by numpy -> exercises partition planning, the
           request hand-shake and HaloExchanger without a GPU.
mode gpu : gloo with host staging, every rank on cuda:0 (RCCL refuses two ranks on one device); stepping, pack and
           unpack by libludwig_hip.so through DistributedLevelRunner -> the real N > 1 GPU path minus RCCL itself.
Each rank writes its owned blocks' newest f / vel / rho to <outdir>/rank<r>.npz.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def main():
    mode, outdir, nbx, nby, nbz, steps, overlap = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from open_ludwig_amd import cases, partition
    from open_ludwig_amd.physics import SolverParams
    grid = partition.rank_grid(world) if world & (world - 1) == 0 else None      # bricks: power-of-two worlds only
    nbg = (nbx, nby, nbz)
    fn, vn = ("f_temp", "vel_temp") if steps % 2 == 0 else ("f", "vel")
    if mode == "cpu":
        from oracle import oracle
        oracle.set_num_threads(2)
        coords, table, owner = partition.periodic_box_topology(nbg, grid)
        view = partition.build_local_level(1, coords, table, owner, rank, 0.5006)
        cases.init_taylor_green(view.level, tuple(8 * n for n in nbg), 0.03)
        params = SolverParams(domain_nx=8 * nbx, domain_ny=8 * nby, domain_nz=8 * nbz, use_temporal_interp=False)
        mine = partition.make_requests(view, len(coords))
        to_me = partition.exchange_requests(mine, world, rank)
        plan = partition.build_plan(view, len(coords), mine, to_me)
        lvl = view.level

        def pack(name, idx, out):
            out.copy_(torch.from_numpy(getattr(lvl, name).reshape(-1, order="F")[idx.numpy()]))

        def unpack(name, idx, src):
            a = getattr(lvl, name)
            flat = a.reshape(-1, order="F")
            flat[idx.numpy()] = src.numpy()
            a[...] = flat.reshape(a.shape, order="F")

        ex = partition.HaloExchanger(plan, rank, torch.device("cpu"), pack, unpack)
        for t in range(1, steps + 1):
            oracle.execute_timestep_batch([lvl], t, 1, np.float32(0.0), params)
            ex.exchange(*(("f_temp", "vel_temp") if t % 2 == 0 else ("f", "vel")))
        res = {n: getattr(lvl, n)[:, :, :, : view.n_owned] for n in (fn, vn, "rho")}
    elif mode == "cpu_forces":
        # distributed diagnostics without a GPU: a sphere in a tunnel, a flow-like field made up on the host; every rank owns a
        # slab of blocks, evaluates the stresses of ITS triangles and the nine partial sums, one all-gather combines them
        import json
        from open_ludwig_amd import forces, preprocess as pp
        G = os.path.join(ROOT, "tests", "golden")
        cfg = pp.load_case_configuration(os.path.join(G, "ball1m_config.yaml"),
                                         {"basic": {"surface_resolution": 25, "num_levels": 2, "flow": {"velocity": 4.0}}})
        grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, os.path.join(G, "ball1m.stl"))
        g = grids[-1]
        rng = np.random.default_rng(5)
        gx, gy, gz = cases.global_cell_coords(g)
        rho = (1.0 + 1e-3 * np.sin(0.11 * gx) * np.cos(0.07 * gz) + 1e-5 * rng.standard_normal(g.rho.shape)).astype(np.float32)
        vel = np.zeros(g.vel.shape, dtype=np.float32, order="F")
        vel[..., 0] = 0.03 * np.cos(0.05 * gy); vel[..., 1] = 0.01 * np.sin(0.09 * gx); vel[..., 2] = 0.02 * np.sin(0.04 * gz + 0.3)
        owner = (np.asarray(g.active_block_coords)[:, 0] * 7 + np.asarray(g.active_block_coords)[:, 2] * 3) % world      # scattered ownership
        nc = forces.nearest_fluid_cells(mesh, g.obstacle, g.block_pointer, g.dx, params, 5)
        sel = np.flatnonzero(nc.found & (owner[nc.block] == rank))
        rho_c = rho[nc.lx[sel], nc.ly[sel], nc.lz[sel], nc.block[sel]]
        u_c = np.stack([vel[nc.lx[sel], nc.ly[sel], nc.lz[sel], nc.block[sel], c] for c in range(3)], axis=1)
        p, tx, ty, tz = forces.stress_from_cells(rho_c, u_c, nc.wall_dist[sel], np.ones(sel.size, bool), mesh.normals[sel], g.tau, params)
        part = np.zeros(10, np.float32)
        part[:9] = forces.partial_force_sums(mesh, p, tx, ty, tz, params, select=sel)
        part[9] = np.count_nonzero(np.abs(p) > 1e-10)
        total, cov = forces.combine_partial_sums(part)
        fr = forces.finish_forces(total, cov, params, False)
        rmin = torch.tensor([float(rho[:, :, :, owner == rank][~g.obstacle[:, :, :, owner == rank]].min())], dtype=torch.float32)
        dist.all_reduce(rmin, op=dist.ReduceOp.MIN)
        single = forces.compute_aerodynamics(mesh, g, rho, vel, params, False)
        json.dump({"dist": [fr.Cd, fr.Cl, fr.Cs, fr.Cmy, fr.Fx_pressure, fr.Fx_viscous, fr.coverage, float(rmin.item())],
                   "single": [single.Cd, single.Cl, single.Cs, single.Cmy, single.Fx_pressure, single.Fx_viscous, single.coverage,
                              float(rho[~g.obstacle].min())], "n_mine": int(sel.size)}, open(os.path.join(outdir, f"forces{rank}.json"), "w"))
        dist.barrier()
        dist.destroy_process_group()
        return
    elif mode == "gpu_tunnel":
        # single-level tunnel with sphere, sponge, Bouzidi cells on both sides of the cut; split along x
        grids, params = cases.tunnel_with_sphere(nbg, levels=1, wall_model=False, temporal=False)
        g = grids[0]
        bx = np.asarray(g.active_block_coords)[:, 0]
        owner = ((bx - 1) * world // nbx).astype(np.int64)
        runner = partition.distributed_level(g, owner, params, rank, world, device=0, overlap=bool(overlap), stage_through_host=True)
        view = runner.view
        for t in range(1, steps + 1):
            runner.step(t, np.float32(0.05))
        runner.synchronize()
        res = {n: runner.level.download(n)[:, :, :, : view.n_owned] for n in (fn, vn, "rho")}
        res["nbc"] = np.array([runner.level.n_boundary_cells, int(runner.ex.plan.has("f_post"))])
    elif mode == "gpu_case":
        # the ball1m case (YAML + STL -> 3 nested levels, wall model, Bouzidi sphere) through run_case on `world` ranks
        import json
        from open_ludwig_amd import case, preprocess as pp
        G = os.path.join(ROOT, "tests", "golden")
        gather = []
        if nbx == 2:      # the REAL wing: CASES/Wing_5_deg/model5deg.stl kept as tests/golden/wing5deg_model.stl, 3 levels (test_case_wing.py)
            stl = os.path.join(G, "wing5deg_model.stl")
            cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), {"basic": {"surface_resolution": 200, "num_levels": 3, "simulation": {"ramp_steps": 40}}})
            gather = [(2, "rho"), (2, "vel"), (0, "rho")]
        elif nbx == 1:    # the synthetic half wing (symmetry plane, inlet turbulence, wall model): tests/_wing.py
            import _wing
            stl = os.path.join(outdir, f"wing_rank{rank}.stl")
            _wing.write_binary_stl(stl, _wing.half_wing_triangles())
            cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), _wing.TEST_OVERRIDES)
        else:
            stl = os.path.join(G, "ball1m.stl")
            cfg = pp.load_case_configuration(os.path.join(G, "ball1m_config.yaml"),
                                             {"basic": {"surface_resolution": 25, "flow": {"velocity": 4.0}, "simulation": {"steps": 6000, "output_freq": 1000}}})
        cfg.diag_freq = 8 if nbx == 2 else 20
        setup = pp.setup_multilevel_domain(cfg, stl)
        holder = {}

        def factory(grids):
            holder["st"] = case.DistributedStepper(grids, device=0, stage_through_host=True)
            if gather:
                holder["st"].close = lambda: None          # the fields are gathered after run_case
            return holder["st"]

        cfg.output_freq = 40
        rows, rep, _ = case.run_case(cfg, factory, steps=steps, setup=setup, out_dir=None if gather else os.path.join(outdir, "results"),
                                     write_files=(rank == 0))
        st = holder["st"]
        if gather:
            got = {f"{name}{lvl}": st.field(lvl, name) for lvl, name in gather}      # collective; rank 0 holds the global arrays
            if rank == 0:
                np.savez(os.path.join(outdir, "fields.npz"), **got)
        stats = [[v.n_owned, v.level.n_blocks, st.runner.plans[i].bytes_per_step()] for i, v in enumerate(st.runner.views)]
        if rank == 0:
            json.dump({"rows": [[r.step, r.u_lat, r.rho_min, r.cd, r.cl, r.cs, r.cmy] for r in rows]}, open(os.path.join(outdir, "rows.json"), "w"))
        json.dump(stats, open(os.path.join(outdir, f"stats{rank}.json"), "w"))
        dist.barrier()
        dist.destroy_process_group()
        return
    elif mode == "gpu_multilevel":
        # nested levels (interface interpolation from parent-data ghosts, temporal blend, Bouzidi sphere), split along x
        levels, wall, per_level = overlap & 7, bool(overlap & 8), bool(overlap & 16)
        grids, params = cases.tunnel_with_sphere(nbg, levels=levels, wall_model=wall, temporal=True)
        if overlap & 32:   # staggered: level 1 entirely on rank 0, the finest level entirely on the last rank, the rest bisected
            owners = partition.level_owners(grids, world)
            owners[0] = np.zeros(grids[0].n_blocks, dtype=np.int64)
            owners[-1] = np.full(grids[-1].n_blocks, world - 1, dtype=np.int64)
        elif per_level:    # every level cut on its own: a fine block's parent may live on another rank
            owners = partition.level_owners(grids, world)
        else:              # x slabs of level-1 blocks, whole hierarchies per rank
            bx = np.asarray(grids[0].active_block_coords)[:, 0]
            owners = ((bx - 1) * world // nbx).astype(np.int64)
        runner = partition.MultiLevelRunner(grids, owners, params, rank, world, device=0, stage_through_host=True)
        for t in range(1, steps + 1):
            runner.step(t, np.float32(0.05))
        runner.synchronize()
        from oracle import oracle as _o   # only for the buffer-parity helper
        res = {}
        for i, (lv, v) in enumerate(zip(runner.levels, runner.views)):
            f_name, v_name = _o.newest_buffers(i, steps)
            res[f"l2g{i}"] = v.local_to_global[: v.n_owned]
            if lv is None or v.n_owned == 0:
                res[f"f{i}"], res[f"vel{i}"], res[f"rho{i}"] = np.zeros((8, 8, 8, 0, 27), np.float32), np.zeros((8, 8, 8, 0, 3), np.float32), np.zeros((8, 8, 8, 0), np.float32)
                res[f"stats{i}"] = np.array([0, v.level.n_blocks, 0, 0])
                continue
            res[f"f{i}"] = lv.download(f_name)[:, :, :, : v.n_owned]
            res[f"vel{i}"] = lv.download(v_name)[:, :, :, : v.n_owned]
            res[f"rho{i}"] = lv.download("rho")[:, :, :, : v.n_owned]
            res[f"stats{i}"] = np.array([v.n_owned, v.level.n_blocks, runner.plans[i].bytes_per_step(), int(runner.plans[i].has("rho"))])
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **res)
        dist.barrier()
        dist.destroy_process_group()
        return
    else:
        per = tuple(nbg[i] // grid[i] for i in range(3))
        runner = partition.periodic_weak_scaling_box(rank, world, per, device=0, overlap=bool(overlap), stage_through_host=True)
        view = runner.view
        for t in range(1, steps + 1):
            runner.step(t)
        runner.synchronize()
        res = {n: runner.level.download(n)[:, :, :, : view.n_owned] for n in (fn, vn, "rho")}
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), l2g=view.local_to_global[: view.n_owned], **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
