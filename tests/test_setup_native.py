"""libludwig_setup.so (include/ludwig_setup.h): the native host-side case set-up against the numpy restatement of the same reference
functions (open_ludwig_amd/preprocess.py, method="numpy"), bit for bit, plus hand-made cases of each entry. CPU only.

What pins what: the set-up integers of the reference's logs (tests/test_case_ball1m.py, test_case_bunny.py) are checked on the DEFAULT
method, which is the native one; this file shows that the numpy restatement - the round-1/2 code those integers pinned before - gives
the same arrays, so both stand on the same anchors."""
import os
import re

import numpy as np
import pytest

from open_ludwig_amd import _setup_lib, preprocess as pp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    _setup_lib.build()
    header = open(os.path.join(ROOT, "include", "ludwig_setup.h")).read()
    declared = sorted(set(re.findall(r"\b(lws_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_setup_lib.EXPORTED_SYMBOLS)
    lib = _setup_lib.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_float16_conversion_is_numpys_single_rounding():
    lib = _setup_lib.load()
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.random(200000), rng.random(50000) * 1e-4, rng.random(20000) * 6e-8, 2.0 ** rng.integers(-30, 17, 2000),
                        [0.0, 1.0, 0.5, 65504.0, 65519.9, 65520.0, 1e9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 2.0 ** -14,
                         1.0 + 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -30, 1.0 + 3 * 2.0 ** -11, 0.99999999]])
    # ties: exactly half-way between two half-precision neighbours, both parities
    h = rng.integers(1, 0x7bff, 5000).astype(np.uint16)
    lo, hi = h.view(np.float16).astype(np.float64), (h + 1).astype(np.uint16).view(np.float16).astype(np.float64)
    x = np.concatenate([x, (lo + hi) / 2])
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    got = np.array([lib.lws_f64_to_f16(float(v)) for v in x], dtype=np.uint16)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]


def _same_levels(a, b):
    assert a[3] == b[3], (a[3], b[3])
    for ga, gb in zip(a[0], b[0]):
        assert ga.active_block_coords == gb.active_block_coords
        for f in ("obstacle", "sponge", "wall_dist", "bouzidi_cell_block", "bouzidi_cell_x", "bouzidi_cell_y", "bouzidi_cell_z", "neighbor_table"):
            x, y = getattr(ga, f), getattr(gb, f)
            assert x.shape == y.shape and x.dtype == y.dtype, (ga.level_id, f)
            assert np.array_equal(x, y), (ga.level_id, f, int((x != y).sum()))
        qa, qb = ga.bouzidi_q_map.view(np.uint16), gb.bouzidi_q_map.view(np.uint16)
        assert qa.shape == qb.shape and np.array_equal(qa, qb), (ga.level_id, "q_map", int((qa != qb).sum()))


@pytest.mark.parametrize("case, stl, overrides", [
    ("cube1m", "cube1m.stl", None),                                                                 # 4 levels, wall model, Bouzidi
    ("wing5deg", "wing5deg_model.stl", {"basic": {"num_levels": 3, "surface_resolution": 120}}),    # 63 196 triangles, symmetric
])
def test_native_setup_equals_the_numpy_restatement(case, stl, overrides):
    cfg = pp.load_case_configuration(os.path.join(G, case + "_config.yaml"), overrides)
    a = pp.setup_multilevel_domain(cfg, os.path.join(G, stl), method="native")
    b = pp.setup_multilevel_domain(cfg, os.path.join(G, stl), method="numpy")
    assert a[3].bouzidi_cells[-1] > 1000 and a[3].flood_fill_filled[-1] > 0
    _same_levels(a, b)


def test_flood_fill_by_hand():
    # 3 x 1 x 1 blocks; a solid wall at global x = 12 (0-based) closes off everything behind it: (24 - 13) * 64 cells get filled
    coords = [(1, 1, 1), (2, 1, 1), (3, 1, 1)]
    obs = np.zeros((8, 8, 8, 3), dtype=bool, order="F")
    obs[4, :, :, 1] = True
    for method in ("native", "numpy"):
        o = obs.copy(order="F")
        assert pp.perform_flood_fill(o, coords, method=method) == 11 * 64
        assert o[5:, :, :, 1].all() and o[:, :, :, 2].all() and not o[:4, :, :, 1].any() and not o[:, :, :, 0].any()
    # a missing block is a wall too: block 3 is reachable only through block 2, which does not exist here
    coords = [(1, 1, 1), (3, 1, 1)]
    for method in ("native", "numpy"):
        o = np.zeros((8, 8, 8, 2), dtype=bool, order="F")
        assert pp.perform_flood_fill(o, coords, method=method) == 512
        assert o[..., 1].all() and not o[..., 0].any()


def test_wall_distance_by_hand():
    coords = [(1, 1, 1), (2, 1, 1)]
    obs = np.zeros((8, 8, 8, 2), dtype=bool, order="F")
    obs[7, 3, 3, 0] = True                       # on the face shared with block 2
    dx = 0.3
    for method in ("native", "numpy"):
        w = pp.compute_wall_distances(coords, obs, dx, method=method)
        assert w.dtype == np.float32 and w.shape == obs.shape
        f = np.float32(dx)
        assert w[6, 3, 3, 0] == np.float32(1.0) * f and w[0, 3, 3, 1] == np.float32(1.0) * f          # across the block face
        assert w[0, 4, 3, 1] == np.sqrt(np.float32(2.0)) * f and w[0, 4, 4, 1] == np.sqrt(np.float32(3.0)) * f
        assert w[7, 3, 3, 0] == np.float32(100.0)                                                    # solid cells keep the default
        assert int((w != np.float32(100.0)).sum()) == 26


def test_bad_arguments_are_reported_not_crashed():
    lib = _setup_lib.load()
    assert lib.lws_voxelize(None, 0, 0.1, None, 0, None, 1) == -1
    assert b"lws_voxelize" in lib.lws_last_error()
    with pytest.raises(ValueError):
        pp.setup_multilevel_domain(pp.load_case_configuration(os.path.join(G, "cube1m_config.yaml")), os.path.join(G, "cube1m.stl"), method="fast")
