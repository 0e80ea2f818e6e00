"""ludwig_level_add_post_collision_readers: a rank of a Bouzidi level that is cut over ranks stores f_post_collision where ITS links
and the links of its PEERS read it, not in every block. GPU; the multi-rank equalities themselves are tests/test_partition_dist.py,
tests/test_rccl_loopback.py and tests/test_case_wing.py (real wing on 2 and 4 ranks), which all run through this call."""
import numpy as np
import pytest

from open_ludwig_amd import adapt, cases, execute_timestep_batch


@pytest.mark.gpu
def test_readers_narrow_the_forced_store_and_results_stay_the_same(gpu):
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=1, wall_model=False)
    g = grids[0]
    assert g.n_boundary_cells > 0
    want = {}
    # an element list as a peer would send it: population 5 of cell (0, 3, 3) of blocks 2 and 7, population 20 of cell (7, 0, 0) of block 2
    sk = g.n_blocks * 512
    readers = np.array([5 * sk + 2 * 512 + (0 + 8 * 3 + 64 * 3), 5 * sk + 7 * 512 + (0 + 8 * 3 + 64 * 3), 20 * sk + 2 * 512 + 7], dtype=np.int64)
    for mode in ("default", "everywhere", "readers"):
        import copy
        h = copy.deepcopy(g)
        if mode == "everywhere":
            h.force_post_collision = True
        if mode == "readers":
            h.force_post_collision = True
            h.post_collision_readers = readers
        d = adapt(h, 0)
        execute_timestep_batch([d], 1, 3, np.float32(0.05), params)
        want[mode] = (np.stack([d.download("f"), d.download("f_temp")]), d.download("f_post_collision"))
        d.close()
    f0, p0 = want["default"]
    f1, p1 = want["everywhere"]
    f2, p2 = want["readers"]
    assert np.array_equal(f0, f1) and np.array_equal(f0, f2), "the populations do not depend on how much of f_post_collision is kept"
    assert (p1 != 0).all(), "everywhere = every cell of every block"
    rows0, rows2 = (p0 != 0).any(axis=(0, 4)), (p2 != 0).any(axis=(0, 4))          # [y, z, block]: rows that were stored
    extra = np.zeros_like(rows0)
    extra[3, 3, 2] = extra[3, 3, 7] = extra[0, 0, 2] = True
    assert np.array_equal(rows2, rows0 | extra), "own links' rows + the three rows the readers named, nothing else"
    assert np.array_equal(p2[:, rows2], p1[:, rows2]), "what is stored is what the full store holds there"


@pytest.mark.gpu
def test_a_rank_without_cells_stores_nothing_once_nobody_reads(gpu):
    grids, params = cases.periodic_box((4, 4, 4))
    g = grids[0]
    g.force_post_collision = True                       # n_boundary_cells < 0 at the ABI: allocate, store everything ...
    d = adapt(g, 0)
    execute_timestep_batch([d], 1, 2, np.float32(0.0), params)
    assert (d.download("f_post_collision") != 0).all()
    d.close()
    g.post_collision_readers = np.zeros(0, np.int64)    # ... until the (empty) set of readers is named
    d = adapt(g, 0)
    execute_timestep_batch([d], 1, 2, np.float32(0.0), params)
    assert not d.download("f_post_collision").any()
    with pytest.raises(Exception):
        d.add_post_collision_readers(np.array([27 * g.n_blocks * 512], dtype=np.int64))       # one past the end
    d.close()
