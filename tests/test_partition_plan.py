"""Partition + halo plan logic on CPU, no torch.distributed: all ranks simulated in one process, the exchange done
with numpy copies, the stepping done by the CPU oracle. Owned blocks of every rank must equal the single-domain run
bit for bit (the exchange moves values, it computes nothing)."""
import numpy as np
import pytest

from open_ludwig_amd import cases, partition
from open_ludwig_amd.physics import SolverParams
from oracle import oracle


def _run_partitioned(nbg, world, steps, grid=None):
    grid = grid or partition.rank_grid(world)
    coords, table, owner = partition.periodic_box_topology(nbg, grid)
    n_global = len(coords)
    params = SolverParams(domain_nx=8 * nbg[0], domain_ny=8 * nbg[1], domain_nz=8 * nbg[2], use_temporal_interp=False)
    views = [partition.build_local_level(1, coords, table, owner, r, 0.5006) for r in range(world)]
    for v in views:
        cases.init_taylor_green(v.level, tuple(8 * n for n in nbg), 0.03)
    reqs = [partition.make_requests(v, n_global) for v in views]
    plans = []
    for r, v in enumerate(views):
        to_me = {q: reqs[q][r] for q in range(world) if r in reqs[q]}
        plans.append(partition.build_plan(v, n_global, reqs[r], to_me))
    for t in range(1, steps + 1):
        for v in views:
            oracle.execute_timestep_batch([v.level], t, 1, np.float32(0.0), params)
        fn, vn = ("f_temp", "vel_temp") if t % 2 == 0 else ("f", "vel")
        for r, v in enumerate(views):           # "send": gather on the owner, "recv": scatter on the asker
            for p in plans[r].peers:
                for name, fld in (("f", fn), ("vel", vn)):
                    src = getattr(views[p].level, fld).reshape(-1, order="F")
                    dst = getattr(v.level, fld).reshape(-1, order="F")
                    snd = plans[p].send[r][name]
                    rcv = plans[r].recv[p][name]
                    assert snd.size == rcv.size
                    dst[rcv] = src[snd]
                    getattr(v.level, fld)[...] = dst.reshape(getattr(v.level, fld).shape, order="F")
    return views, plans, (coords, table, params)


@pytest.mark.parametrize("nbg,world", [((4, 2, 2), 2), ((4, 4, 2), 4), ((4, 4, 4), 8), ((2, 2, 2), 8), ((6, 2, 3), 2)])
def test_partitioned_oracle_matches_single_domain(nbg, world):
    steps = 4
    views, plans, (coords, table, params) = _run_partitioned(nbg, world, steps)
    grids, params1 = cases.periodic_box(nbg)
    oracle.execute_timestep_batch(grids, 1, steps, np.float32(0.0), params1)
    g = grids[0]
    fn, vn = oracle.newest_buffers(0, steps)
    for v in views:
        gl = v.local_to_global[: v.n_owned]
        for name in (fn, vn, "rho"):
            a = getattr(v.level, name)[:, :, :, : v.n_owned]
            b = getattr(g, name)[:, :, :, gl]
            assert np.array_equal(a, b), f"rank {v.rank} {name} differs from the single-domain run"


def test_plan_sizes_match_face_layer_count():
    """Size of the exchange, derived independently of the plan code: population k of an n^3 brick pulls from outside
    the brick for n^3 - (n-|cx|)(n-|cy|)(n-|cz|) cells, each from a distinct ghost element; the WALE stencil reads
    the 6 n^2 face-adjacent ghost cells, 3 components each. (SURVEY 8e's "9 per face, 3 per edge, 1 per corner" is
    the same count before removing the rim cells whose pulling cell lies outside the brick.)"""
    nbg, world = (4, 4, 4), 8      # bricks of 2^3 blocks = 16^3 cells
    coords, table, owner = partition.periodic_box_topology(nbg, partition.rank_grid(world))
    v = partition.build_local_level(1, coords, table, owner, 0, 0.5006)
    needs = partition.compute_needs(v)
    n = 16
    f_expected = sum(n ** 3 - (n - abs(cx)) * (n - abs(cy)) * (n - abs(cz))
                     for cx in (-1, 0, 1) for cy in (-1, 0, 1) for cz in (-1, 0, 1))
    assert needs["f"].size == f_expected == 13256
    assert needs["vel"].size == 6 * n * n * 3
    assert int(v.level.comm_boundary.sum()) == 8      # every block of a 2^3 brick touches a ghost


def test_owned_blocks_keep_reference_order_and_ghosts_follow():
    coords, table, owner = partition.periodic_box_topology((4, 4, 2), (2, 2, 1))
    for r in range(4):
        v = partition.build_local_level(1, coords, table, owner, r, 0.6)
        assert (np.diff(v.local_to_global[: v.n_owned]) > 0).all()
        assert (owner[v.local_to_global[: v.n_owned]] == r).all()
        assert (owner[v.local_to_global[v.n_owned:]] != r).all()
        assert (np.asarray(v.level.neighbor_table)[v.n_owned:] == 0).all()
        # every neighbour of an owned block is present locally (periodic box: 26 neighbours)
        assert (np.asarray(v.level.neighbor_table)[: v.n_owned] > 0).all()
