"""Host-side mirror of the reference's data model and control logic (no GPU, no oracle)."""
import numpy as np
import pytest

from open_ludwig_amd import cases, order
from open_ludwig_amd.blocks import BlockLevel, build_neighbor_table
from open_ludwig_amd.physics import SolverParams


def _brute_neighbor_table(coords, bx_max, by_max, bz_max):
    """src/domain_topology.jl:135-160 restated with plain loops"""
    n = len(coords)
    table = np.zeros((n, 27), np.int32)
    ptr = {c: i + 1 for i, c in enumerate(coords)}
    for i, (bx, by, bz) in enumerate(coords):
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    d = (dx + 1) + (dy + 1) * 3 + (dz + 1) * 9
                    nb = (bx + dx, by + dy, bz + dz)
                    if 1 <= nb[0] <= bx_max and 1 <= nb[1] <= by_max and 1 <= nb[2] <= bz_max:
                        table[i, d] = ptr.get(nb, 0)
    return table


def test_build_neighbor_table_full_and_sparse():
    coords = cases.full_box_coords(3, 2, 4)
    assert np.array_equal(build_neighbor_table(coords, 3, 2, 4), _brute_neighbor_table(coords, 3, 2, 4))
    rng = np.random.default_rng(1)
    sparse = sorted({tuple(int(v) for v in rng.integers(1, 6, 3)) for _ in range(40)})
    assert np.array_equal(build_neighbor_table(sparse, 5, 5, 5), _brute_neighbor_table(sparse, 5, 5, 5))
    t = build_neighbor_table(coords, 3, 2, 4)
    assert (t[:, 13] == np.arange(1, len(coords) + 1)).all()       # direction 14 (1-based) is the block itself


def test_periodic_table_wraps():
    coords = cases.full_box_coords(2, 3, 2)
    t = build_neighbor_table(coords, 2, 3, 2, (True, True, True))
    assert (t > 0).all()
    idx = {c: i + 1 for i, c in enumerate(coords)}
    assert t[idx[(1, 1, 1)] - 1, 12] == idx[(2, 1, 1)]             # dx = -1 wraps to bx = 2
    assert t[idx[(1, 3, 1)] - 1, 16] == idx[(1, 1, 1)]             # dy = +1 wraps to by = 1


def test_block_order_is_reference_sort():
    coords = cases.full_box_coords(2, 2, 3)
    assert coords == sorted(coords) and coords[1] == (1, 1, 2)     # bz varies fastest (src/domain.jl:171)


def test_block_level_defaults_match_reference_constructor():
    coords = cases.full_box_coords(2, 1, 1)
    L = BlockLevel(1, coords, build_neighbor_table(coords, 2, 1, 1), 0.1, 1.0, 0.6)
    assert L.rho.shape == (8, 8, 8, 2) and L.rho.dtype == np.float32 and (L.rho == 1).all()
    assert L.f.shape == (8, 8, 8, 2, 27) and (L.f == 0).all() and L.f.flags.f_contiguous
    assert L.vel.shape == (8, 8, 8, 2, 3)
    assert (L.wall_dist == 100).all() and not L.obstacle.any() and (L.sponge == 0).all()
    assert L.f_post_collision.shape == (1, 1, 1, 1, 27)            # dummy without Bouzidi cells (src/blocks.jl:138)
    assert L.f_old.size > 27
    assert L.block_pointer.shape == (2, 1, 1) and list(L.block_pointer.reshape(-1)) == [1, 2]
    L2 = BlockLevel(1, coords, build_neighbor_table(coords, 2, 1, 1), 0.1, 1.0, 0.6, enable_temporal_interpolation=False)
    assert L2.f_old.size == 27 and L2.rho_old.size == 1


def test_memory_layout_is_julia_column_major():
    coords = cases.full_box_coords(1, 1, 2)
    L = BlockLevel(1, coords, build_neighbor_table(coords, 1, 1, 2), 1, 1, 0.6)
    L.f[2, 3, 4, 1, 5] = 7.0          # Julia f[3,4,5,2,6]
    flat = L.f.reshape(-1, order="F")
    assert flat[2 + 8 * 3 + 64 * 4 + 512 * 1 + 512 * 2 * 5] == 7.0


@pytest.mark.parametrize("name", list(order.BUILDERS))
def test_launch_orders_cover_every_block_plane_once(name):
    coords = np.array(cases.full_box_coords(4, 6, 3))
    items = order.build(name, coords)
    real = items[items >= 0]
    assert len(items) % 4 == 0
    assert sorted(real.tolist()) == list(range(len(coords) * 8))


def test_refine_region_children_cover_parents():
    grids, _ = cases.tunnel_with_sphere((6, 4, 4), levels=3, bouzidi=False)
    for parent, child in zip(grids[:-1], grids[1:]):
        pset = set(parent.active_block_coords)
        for (bx, by, bz) in child.active_block_coords:
            assert ((bx + 1) // 2, (by + 1) // 2, (bz + 1) // 2) in pset      # no orphans (src/domain.jl:114-127)
        assert child.n_blocks % 8 == 0
        assert child.tau < parent.tau                                          # tau - 1/2 halves per level


def test_solver_params_struct_roundtrip():
    p = SolverParams(domain_nx=64, domain_ny=56, domain_nz=48, wall_model_active=True, c_wale=0.325, nu_sgs_bg=0.0005,
                     inlet_turbulence=0.01, use_temporal_interp=True, sponge_blend_dist=True, symmetric_analysis=True,
                     q_min_threshold=0.001)
    c = p.to_c()
    assert (c.domain_nx, c.domain_ny, c.domain_nz) == (64, 56, 48)
    assert c.wall_model_active == 1 and c.is_symmetric == 1 and c.use_temporal_interp == 1 and c.sponge_blend_distributions == 1
    assert np.float32(c.c_wale) == np.float32(0.325) and np.float32(c.q_min_threshold) == np.float32(0.001)
