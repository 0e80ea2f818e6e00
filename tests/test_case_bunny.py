"""A non-convex body from the reference's case set (CASES/Stanford_bunny, 4 968 triangles) at a reduced size: exercises
the pre-processing on a mesh with concavities and thin parts, sponge reaching level 2, 16 k Bouzidi cells, the wall model.
The reference keeps no output for this case: the setup numbers below are a regression of THIS repo's pre-processing
("parity unpinned" for them); what is pinned is HIP against the CPU oracle on the resulting levels."""
import os
import sys

import numpy as np
import pytest

from open_ludwig_amd import case, preprocess as pp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
G = os.path.join(HERE, "golden")
OVERRIDES = {"basic": {"surface_resolution": 64, "num_levels": 3, "simulation": {"ramp_steps": 40}}}


@pytest.fixture(scope="module")
def bunny():
    cfg = pp.load_case_configuration(os.path.join(G, "bunny_config.yaml"), OVERRIDES)
    cfg.diag_freq = 16
    return cfg, os.path.join(G, "bunny.stl")


def test_bunny_setup(bunny):
    cfg, stl = bunny
    grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, stl)
    assert mesh.triangles.shape[0] == 4968 and not cfg.symmetric_analysis and cfg.wall_model_enabled
    assert rep.level_blocks == [175, 1000, 2400] and rep.halo_blocks_added[1:] == [966, 2260]
    assert rep.bouzidi_cells == [16372] and rep.flood_fill_filled == [119, 1758, 4]
    fin = grids[-1]
    q = fin.bouzidi_q_map.astype(np.float32)
    assert ((q >= 0) & (q <= 1)).all()
    # every listed boundary cell is a fluid cell with at least one cut link, and no block of a fine level lacks its parent
    cb = fin.bouzidi_cell_block.astype(np.int64) - 1
    x, y, z = (a.astype(np.int64) - 1 for a in (fin.bouzidi_cell_x, fin.bouzidi_cell_y, fin.bouzidi_cell_z))
    assert (q[x, y, z, cb, :] > 0).any(axis=1).all()
    for lvl in (1, 2):
        parents = {tuple(c) for c in grids[lvl - 1].active_block_coords}
        assert all(((bx + 1) // 2, (by + 1) // 2, (bz + 1) // 2) in parents for bx, by, bz in grids[lvl].active_block_coords)


@pytest.mark.gpu
def test_bunny_hip_equals_oracle(gpu, bunny):
    from _steppers import OracleStepper
    from oracle import oracle
    oracle.set_num_threads(16)
    cfg, stl = bunny
    steps = 48
    setup_h, setup_o = pp.setup_multilevel_domain(cfg, stl), pp.setup_multilevel_domain(cfg, stl)
    keep = {}

    def hip_factory(grids):
        keep["st"] = case.HipStepper(grids)
        keep["st"].close = lambda: None
        return keep["st"]

    hip, _, _ = case.run_case(cfg, hip_factory, steps=steps, setup=setup_h)
    ora, _, _ = case.run_case(cfg, OracleStepper, steps=steps, setup=setup_o)
    assert [r.step for r in hip] == [16, 32, 48] == [r.step for r in ora]
    scale = max(abs(r.cd) for r in ora)
    assert scale > 1e-3
    for a, b in zip(hip, ora):
        for name in ("cd", "cl", "cs", "cmy"):
            assert getattr(a, name) == getattr(b, name), (a.step, name, getattr(a, name), getattr(b, name))
    for i, g in enumerate(setup_o[0]):
        fn, vn = oracle.newest_buffers(i, steps)
        for name in ("rho", vn, fn):
            xh, xo = keep["st"].dev[i].download(name), getattr(g, name)
            assert np.array_equal(xh, xo), (i + 1, name, int(np.count_nonzero(xh != xo)))
    for d in keep["st"].dev:
        d.close()
