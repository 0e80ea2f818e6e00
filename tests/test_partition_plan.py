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


@pytest.mark.parametrize("nbg,world,grid", [((4, 2, 2), 2, None), ((4, 4, 2), 4, None), ((4, 4, 4), 8, None), ((2, 2, 2), 8, None),
                                            ((3, 2, 6), 2, None), ((4, 4, 4), 8, (1, 2, 4)), ((2, 2, 8), 8, (1, 1, 8))])
def test_partitioned_oracle_matches_single_domain(nbg, world, grid):
    steps = 4
    views, plans, (coords, table, params) = _run_partitioned(nbg, world, steps, grid)
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


def test_interpolation_needs_match_a_per_link_walk():
    """Parent-data ghosts (scope row N3): the vectorised list equals a plain per-link walk of the interface rule
    (src/physics_interpolation.jl:29-62: source cell outside the fine level, 8 parent cells around (g - 0.5) / 2)."""
    from open_ludwig_amd import cases
    from open_ludwig_amd.blocks import build_lattice_arrays
    nbg = (6, 4, 4)
    grids, params = cases.tunnel_with_sphere(nbg, levels=2, wall_model=False, temporal=True)
    cx, cy, cz = (np.asarray(a) for a in build_lattice_arrays()[:3])
    bx = np.asarray(grids[0].active_block_coords)[:, 0]
    owner1 = ((bx - 1) * 2 // nbg[0]).astype(np.int64)
    total = 0
    for rank in range(2):
        views = []
        for g in grids:
            own = owner1 if g.level_id == 1 else partition.ancestor_owner(g.level_id, g.active_block_coords, grids[0].active_block_coords, owner1)
            views.append(partition.build_local_level(g.level_id, g.active_block_coords, g.neighbor_table, own, rank, float(g.tau), temporal=True))
        child, parent = views[1], views[0]
        got = partition.interpolation_needs(child, parent, (params.domain_nx, params.domain_ny, params.domain_nz))
        cl, pl = child.level, parent.level
        pnb = pl.n_blocks
        want_f, want_r = set(), set()
        lim = (2 * params.domain_nx, 2 * params.domain_ny, 2 * params.domain_nz)
        for b in range(child.n_owned):
            if (cl.neighbor_table[b] != 0).all():
                continue
            org = [(int(m[b]) - 1) * 8 for m in (cl.map_x, cl.map_y, cl.map_z)]
            for k in range(27):
                c = (int(cx[k]), int(cy[k]), int(cz[k]))
                for z in range(8):
                    for y in range(8):
                        for x in range(8):
                            s = (x - c[0], y - c[1], z - c[2])
                            o = [(-1 if v < 0 else (1 if v > 7 else 0)) for v in s]
                            if o == [0, 0, 0] or cl.neighbor_table[b, (o[0] + 1) + 3 * (o[1] + 1) + 9 * (o[2] + 1)] != 0:
                                continue
                            g = [org[a] + s[a] + 1 for a in range(3)]
                            if any(g[a] < 1 or g[a] > lim[a] for a in range(3)):
                                continue
                            lo = [max(int(np.floor((g[a] - 0.5) * 0.5)), 1) for a in range(3)]
                            hi = [int(np.floor((g[a] - 0.5) * 0.5)) + 1 for a in range(3)]
                            for n in range(8):
                                pg = [hi[a] if (n >> a) & 1 else lo[a] for a in range(3)]
                                pb = [(v - 1) // 8 for v in pg]
                                if any(pb[a] < 0 or pb[a] >= pl.block_pointer.shape[a] for a in range(3)):
                                    continue
                                idx = int(pl.block_pointer[pb[0], pb[1], pb[2]])
                                if idx > parent.n_owned:
                                    cell = (pg[0] - 1) % 8 + 8 * ((pg[1] - 1) % 8) + 64 * ((pg[2] - 1) % 8)
                                    want_f.add((k * pnb + idx - 1) * 512 + cell)
                                    want_r.add((idx - 1) * 512 + cell)
        assert set(got["f"].tolist()) == want_f
        assert set(got["rho"].tolist()) == want_r
        assert got["vel"].size == 3 * len(want_r)
        total += len(want_f)
    assert total > 0, "the cut was meant to pass through the refined region"


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_balanced_owner_cuts_nested_levels_by_work(world):
    """Level-1 blocks are dealt by weighted bisection (work = own step + 2^(l-1) sub-steps of every descendant block); a
    fine block always lands on the rank of its level-1 ancestor, so hierarchies never straddle ranks."""
    from open_ludwig_amd import cases
    grids, _ = cases.tunnel_with_sphere((8, 4, 4), levels=3, wall_model=False, temporal=True)
    owner1 = partition.balanced_owner(grids, world)
    c1 = np.asarray(grids[0].active_block_coords)
    assert owner1.shape == (grids[0].n_blocks,) and set(owner1.tolist()) == set(range(world)), "every rank owns something"
    lut = {tuple(c): i for i, c in enumerate(c1)}
    work = np.ones(len(c1))
    for g in grids[1:]:
        own = partition.ancestor_owner(g.level_id, g.active_block_coords, grids[0].active_block_coords, owner1)
        for c, o in zip(np.asarray(g.active_block_coords), own):
            a = tuple(((c - 1) >> (g.level_id - 1)) + 1)
            assert owner1[lut[a]] == o
            work[lut[a]] += 2 ** (g.level_id - 1)
    per_rank = np.array([work[owner1 == r].sum() for r in range(world)])
    # cuts are planes between level-1 blocks, and in this small case the refined core is only one or two such blocks thick:
    # the bound is what that granularity allows, not a balance claim (ball1m, refined core 3 level-1 blocks wide: 64 % / 36 % on 2 ranks)
    assert per_rank.max() <= 0.75 * per_rank.sum(), per_rank
    # every rank's region is a box of level-1 blocks (planar cuts)
    for r in range(world):
        c = c1[owner1 == r]
        lo, hi = c.min(axis=0), c.max(axis=0)
        assert len(c) == np.prod(hi - lo + 1)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_parent_ghosts_cover_every_interpolation_stencil(world):
    """Per-level cuts (level_owners): a rank keeps ghost copies of the parent blocks required_parent_blocks lists. They must
    hold every parent cell the interface rule reads: the needs computed against that local parent level equal (as GLOBAL
    elements) the needs computed against a parent level that has ALL remote parent blocks as ghosts."""
    from open_ludwig_amd import cases
    grids, params = cases.tunnel_with_sphere((8, 4, 4), levels=3, wall_model=False, temporal=True)
    dims = (params.domain_nx, params.domain_ny, params.domain_nz)
    owners = partition.level_owners(grids, world)
    for o, g in zip(owners, grids):
        counts = np.bincount(o, minlength=world)
        assert counts.sum() == g.n_blocks and counts.max() - counts.min() <= max(2, 0.2 * counts.mean()), counts
    total = 0
    for rank in range(world):
        for i in (0, 1):                                   # parent level index; child = i + 1
            gp, gc = grids[i], grids[i + 1]
            child = partition.build_local_level(gc.level_id, gc.active_block_coords, gc.neighbor_table, owners[i + 1], rank, float(gc.tau), temporal=True)
            extra = partition.required_parent_blocks(gc, np.flatnonzero(owners[i + 1] == rank), gp)
            local = partition.build_local_level(gp.level_id, gp.active_block_coords, gp.neighbor_table, owners[i], rank, float(gp.tau), temporal=True, extra_ghosts=extra)
            full = partition.build_local_level(gp.level_id, gp.active_block_coords, gp.neighbor_table, owners[i], rank, float(gp.tau), temporal=True,
                                               extra_ghosts=np.arange(gp.n_blocks))
            assert full.level.n_blocks == gp.n_blocks and local.level.n_blocks <= full.level.n_blocks
            a = partition.interpolation_needs(child, local, dims)
            b = partition.interpolation_needs(child, full, dims)
            for name in ("f", "rho", "vel"):
                ga = np.sort(partition._to_global(local, a[name], gp.n_blocks))
                gb = np.sort(partition._to_global(full, b[name], gp.n_blocks))
                assert np.array_equal(ga, gb), (rank, i, name, ga.size, gb.size)
            total += a["f"].size
    assert total > 0


def test_widened_boundary_part_keeps_whole_x_runs():
    """build_local_level(widen_x_runs=True): the boundary part is a superset of the blocks with a ghost neighbour, and along x it
    consists of whole aligned groups of 4 blocks (the stream-collide kernel's workgroup), so no boundary block steps without its run."""
    nb, grid = 8, (2, 2, 2)
    nbg = tuple(nb * g for g in grid)
    coords, table, owner = partition.periodic_box_topology(nbg, grid)
    plain = partition.build_local_level(1, coords, table, owner, 0, 0.5006)
    wide = partition.build_local_level(1, coords, table, owner, 0, 0.5006, widen_x_runs=True)
    n = plain.n_owned
    a, b = plain.level.comm_boundary[:n] != 0, wide.level.comm_boundary[:n] != 0
    assert a.sum() == nb ** 3 - (nb - 2) ** 3 and (b | a).sum() == b.sum() and b.sum() > a.sum()
    c = np.asarray(wide.level.active_block_coords)[:n]
    groups = {}
    for (bx, by, bz), m in zip(c, b):
        groups.setdefault(((bx - 1) // 4, by, bz), []).append(bool(m))
    assert all(len(v) == 4 and len(set(v)) == 1 for v in groups.values())
    assert not wide.level.comm_boundary[n:].any()
    # interior blocks still have no ghost neighbour
    ghosty = (np.asarray(wide.level.neighbor_table)[:n] > n).any(axis=1)
    assert not (ghosty & ~b).any()


def test_weak_scaling_layout_keeps_the_cells_per_rank_and_the_global_box():
    """bench.py's N > 1 boxes: nb^3 blocks per rank for every world size; 8 ranks = the (2 nb)^3 box (BASELINE configs[3]) cut
    1 x 2 x 4 - no x face; 2 and 4 ranks cut z, then y."""
    for world in (1, 2, 4, 8, 16):
        brick, grid = partition.weak_scaling_layout(world, 32)
        assert brick[0] * brick[1] * brick[2] == 32 ** 3 and grid[0] * grid[1] * grid[2] == world
    assert partition.weak_scaling_layout(8, 32) == ((64, 32, 16), (1, 2, 4))
    assert tuple(b * g for b, g in zip(*partition.weak_scaling_layout(8, 32))) == (64, 64, 64)
    assert partition.weak_scaling_layout(2, 32) == ((32, 32, 32), (1, 1, 2)) and partition.weak_scaling_layout(4, 32)[1] == (1, 2, 2)


def test_strong_scaling_layout_keeps_the_global_box():
    """bench.py --scaling strong: BASELINE configs[3], the SAME 512^3 box (64^3 blocks) at 1 / 2 / 4 / 8 ranks, z and y cuts only."""
    want = {1: ((64, 64, 64), (1, 1, 1)), 2: ((64, 64, 32), (1, 1, 2)), 4: ((64, 32, 32), (1, 2, 2)), 8: ((64, 32, 16), (1, 2, 4))}
    for world, (brick, grid) in want.items():
        assert partition.strong_scaling_layout(world, 64) == (brick, grid)
        assert tuple(b * g for b, g in zip(brick, grid)) == (64, 64, 64)
    assert partition.strong_scaling_layout(8, 64) == partition.weak_scaling_layout(8, 32)      # the two modes meet at configs[3]
    with pytest.raises(ValueError):
        partition.strong_scaling_layout(3, 64)
    with pytest.raises(ValueError):
        partition.strong_scaling_layout(8, 8)            # bricks thinner than an x-run


def test_eight_rank_layout_has_three_equal_face_peers_and_no_x_face():
    """weak_scaling_layout(8, nb): bricks of 2nb x nb x nb/2 blocks in a 1 x 2 x 4 rank grid - two z neighbours with one face each and
    one y neighbour with both faces, the same bytes each (6.3 MB at nb = 32), two edge peers with a sliver, nothing across x."""
    nb = 8
    brick, grid = partition.weak_scaling_layout(8, nb)
    nbg = tuple(brick[i] * grid[i] for i in range(3))
    coords, table, owner = partition.periodic_box_topology(nbg, grid)
    assert (np.bincount(owner) == nb ** 3).all()
    view = partition.build_local_level(1, coords, table, owner, 5, 0.5006, widen_x_runs=True)
    mine = partition.make_requests(view, len(coords))
    per_peer = {p: 4 * sum(a.size for a in r.values()) for p, r in mine.items()}
    assert len(per_peer) == 5
    big = sorted(per_peer.values())[-3:]
    face = brick[0] * brick[1] * 64 * (9 + 3) * 4              # one z face: 9 populations + 3 velocity components per cell
    assert all(0.98 * face <= b <= face for b in big) and sum(sorted(per_peer.values())[:2]) < 0.02 * face      # rim cells go to the edge peers
    gx = np.asarray(view.level.active_block_coords)[view.n_owned:, 0]
    assert gx.min() >= 1 and gx.max() <= nbg[0] and view.level.n_blocks - view.n_owned == 2 * brick[0] * brick[1] + 2 * brick[0] * brick[2] + 4 * brick[0]
    # the boundary part is whole x rows: widening to groups of 4 adds nothing
    plain = partition.build_local_level(1, coords, table, owner, 5, 0.5006)
    assert np.array_equal(plain.level.comm_boundary, view.level.comm_boundary)


def test_post_collision_readers_are_the_f_post_send_lists():
    """A Bouzidi level cut in two through the body: what a rank must keep of f_post_collision for OTHERS is exactly what its plan sends
    in the f_post group, and every such element is the cell one step behind a peer's link with 0 < q < 1/2 (src/bouzidi_kernel.jl:47-58).
    `name_post_collision_readers` hands that list to the level (DeviceLevel passes it to ludwig_level_add_post_collision_readers)."""
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=1, wall_model=False)
    g = grids[0]
    bx = np.asarray(g.active_block_coords)[:, 0]
    owner = (bx > bx.min() + 2).astype(np.int64)            # the cut runs through the sphere's blocks
    views = []
    for r in range(2):
        v = partition.build_local_level(1, g.active_block_coords, g.neighbor_table, owner, r, float(g.tau), temporal=g.f_old.size > 27)
        partition.slice_level_fields(v, g)
        views.append(v)
    reqs = [partition.make_requests(v, g.n_blocks, partition.compute_needs(v)) for v in views]
    plans = [partition.build_plan(views[r], g.n_blocks, reqs[r], {q: reqs[q][r] for q in range(2) if r in reqs[q]}) for r in range(2)]
    assert all(v.level.force_post_collision for v in views) and any(p.has("f_post") for p in plans)
    total = 0
    for r, (v, p) in enumerate(zip(views, plans)):
        partition.name_post_collision_readers(v.level, p)
        readers = v.level.post_collision_readers
        want = np.concatenate([np.asarray(p.send[q]["f_post"], dtype=np.int64) for q in p.peers]) if p.peers else np.zeros(0, np.int64)
        assert np.array_equal(np.sort(readers), np.sort(want))
        # every named element lies in an OWNED block of this rank and is what the peer receives
        blk = (readers % (v.level.n_blocks * 512)) // 512
        assert (blk < v.n_owned).all()
        for q in p.peers:
            assert len(p.send[q]["f_post"]) == len(plans[q].recv[r]["f_post"])
        total += readers.size
    assert total > 0, "no link reaches across the cut: the case does not exercise the readers"
