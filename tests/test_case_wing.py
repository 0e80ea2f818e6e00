"""The reference's symmetric half-model configuration (CASES/Wing_5_deg/config.yaml: symmetry plane at y = 0, wall model,
inlet turbulence 1 %, Bouzidi on the finest level, forces doubled for the full model) on a synthetic half wing of the same
proportions, reduced to 3 levels. BASELINE configs[4] names this case; its own STL (3 MB) and resolution (1100 cells per
14 m, 5 levels) do not fit a test, the code path does."""
import copy
import json
import os
import sys

import numpy as np
import pytest

from open_ludwig_amd import case, preprocess as pp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _wing  # noqa: E402

G = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def wing(tmp_path_factory):
    stl = str(tmp_path_factory.mktemp("wing") / "wing.stl")
    _wing.write_binary_stl(stl, _wing.half_wing_triangles())
    cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), _wing.TEST_OVERRIDES)
    cfg.diag_freq = 20
    return cfg, stl


def test_symmetric_setup(wing):
    cfg, stl = wing
    grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, stl)
    assert cfg.symmetric_analysis and cfg.wall_model_enabled and float(cfg.inlet_turbulence_intensity) == pytest.approx(0.01)
    assert mesh.triangles.shape[0] == 1440 and rep.level_blocks == [300, 640, 1248] and rep.bouzidi_cells == [7451]
    assert params.mesh_offset[1] == 0.0, "symmetric analysis: the model's y = 0 plane is the domain's y-min face"
    assert cfg.reference_area_config == pytest.approx(135.3 / 2)
    # the root section lies on the symmetry plane: solid cells touch the y-min face of the domain on the finest level
    fin = grids[-1]
    on_plane = fin.map_y == 1
    assert fin.obstacle[:, 0, :, on_plane].any()


@pytest.mark.gpu
def test_wing_hip_equals_oracle(gpu, wing):
    """60 coarse steps through run_case on HIP and on the CPU oracle: rows and the fields of every level identical (wall model
    active: shared jl_math.h)."""
    from _steppers import OracleStepper
    from oracle import oracle
    oracle.set_num_threads(16)
    cfg, stl = wing
    steps = 60
    setup_h, setup_o = pp.setup_multilevel_domain(cfg, stl), pp.setup_multilevel_domain(cfg, stl)
    keep = {}

    def hip_factory(grids):
        keep["st"] = case.HipStepper(grids)
        keep["st"].close = lambda: None
        return keep["st"]

    hip, _, _ = case.run_case(cfg, hip_factory, steps=steps, setup=setup_h)
    ora, _, _ = case.run_case(cfg, OracleStepper, steps=steps, setup=setup_o)
    assert [r.step for r in hip] == [20, 40, 60] == [r.step for r in ora]
    scale = max(abs(r.cd) for r in ora)
    assert scale > 1e-3, "the short ramp must have produced a real load"
    for a, b in zip(hip, ora):
        for name in ("cd", "cl", "cs", "cmy"):
            assert getattr(a, name) == getattr(b, name), (a.step, name, getattr(a, name), getattr(b, name))
        assert a.rho_min == b.rho_min
    for i, g in enumerate(setup_o[0]):
        fn, vn = oracle.newest_buffers(i, steps)
        for name in ("rho", vn, fn):
            x, y = keep["st"].dev[i].download(name), getattr(g, name)
            assert np.array_equal(x, y), (i + 1, name, int(np.count_nonzero(x != y)))
    for d in keep["st"].dev:
        d.close()


@pytest.mark.gpu
def test_wing_on_two_ranks_equals_single_device(gpu, wing, tmp_path):
    import test_partition_dist as tpd
    cfg, stl = wing
    steps = 60
    single, _, _ = case.run_case(cfg, case.HipStepper, steps=steps, setup=pp.setup_multilevel_domain(cfg, stl))
    tpd._launch("gpu_case", tmp_path, (1, 0, 0), steps, world=2)
    rows = json.load(open(os.path.join(tmp_path, "rows.json")))["rows"]
    assert len(rows) == len(single) == 3
    scale = max(abs(r.cd) for r in single)
    for got, want in zip(rows, single):          # forces: per-rank partial sums added in rank order (1e-6); the rest exact
        assert got[:3] == [want.step, want.u_lat, want.rho_min], (got, want)
        for a, b in zip(got[3:], [want.cd, want.cl, want.cs, want.cmy]):
            assert abs(a - b) <= 1e-6 * scale, (got, want)
    # result files: written once (rank 0), from fields gathered over both ranks
    from test_output_files import read_vtu
    from open_ludwig_amd import output
    res = os.path.join(tmp_path, "results")
    assert sorted(os.listdir(res)) == ["convergence.csv", "flow_000040.vtu", "forces.csv", "surface_000040.vtu"]
    assert len(open(os.path.join(res, "forces.csv")).read().splitlines()) == 4
    grids = pp.setup_multilevel_domain(cfg, stl)[0]
    d = read_vtu(os.path.join(res, "flow_000040.vtu"))
    assert d["n_cells"] == 512 * len(output.select_export_blocks([g.active_block_coords for g in grids]))
    assert np.isfinite(d["Velocity"]).all() and np.abs(d["Velocity"]).max() > 0 and "Density" not in d      # density: false in this case's config


# ---- BASELINE configs[4] on its OWN geometry: CASES/Wing_5_deg/model5deg.stl (63 196 triangles, kept as a data fixture) ----
# The shipped configuration is 5 levels at 1 100 cells per 14 m; BASELINE names the 3-level variant. Overrides, stated:
# num_levels 3, surface_resolution 200 (dx_fine = 0.07 m; 2 090 / 1 728 / 5 256 blocks = 4.65 M cells, 13.6 M cell updates per
# coarse step, 67 096 Bouzidi cells), ramp_steps 40 so that 24 coarse steps already load the wing. Everything else as shipped:
# symmetric half model (the STL holds both halves: the y < 0 half lies outside the domain), wall model, inlet turbulence 1 %,
# Bouzidi on the finest level, temporal interpolation. The reference holds no log or result file for this case, so its setup
# integers below pin this repository's restatement only (parity of the setup: unpinned); the stepping is HIP vs oracle.
REAL_OVERRIDES = {"basic": {"surface_resolution": 200, "num_levels": 3, "simulation": {"ramp_steps": 40}}}


@pytest.fixture(scope="module")
def wing_real():
    cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), REAL_OVERRIDES)
    cfg.diag_freq = 8
    return cfg, os.path.join(G, "wing5deg_model.stl")


def test_real_wing_setup(wing_real):
    cfg, stl = wing_real
    grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, stl)
    assert mesh.triangles.shape[0] == 63196
    assert rep.level_blocks == [2090, 1728, 5256] and rep.bouzidi_cells == [67096] and rep.flood_fill_filled == [143, 4619, 56967]
    assert (params.bx_max, params.by_max, params.bz_max) == (19, 11, 10) and params.mesh_offset[1] == 0.0
    fin = grids[-1]
    assert fin.obstacle[:, 0, :, fin.map_y == 1].any(), "the root section cuts the symmetry plane"
    near = (fin.wall_dist > 0) & (fin.wall_dist < 10)
    assert near.sum() > 30_000                                  # wall-model cells all around a thin, non-convex body
    assert float(grids[0].tau) == pytest.approx(0.5000086, abs=1e-7)


# The run length BASELINE states is the shipped one: steps 10 000, ramp_steps 2 000 (CASES/Wing_5_deg/config.yaml:78-79).
SHIPPED_RUN = {"basic": {"surface_resolution": 200, "num_levels": 3}}


@pytest.mark.gpu
def test_real_wing_hip_equals_oracle(gpu):
    """The first 200 coarse steps of the run AS SHIPPED (ramp over 2 000 steps: the inlet reaches 2.4 % of its speed) on HIP and on
    the CPU oracle, 2.7 G cell updates: every level's rho / u / f and every Cd / Cl / Cs / Cmy / rho_min row identical; the HIP
    stepping rate is printed for profiles/."""
    import copy
    import time
    from _steppers import OracleStepper
    from oracle import oracle
    oracle.set_num_threads(16)
    cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"), SHIPPED_RUN)
    assert (cfg.steps, cfg.ramp_steps) == (10000, 2000)
    cfg.diag_freq = 50
    stl = os.path.join(G, "wing5deg_model.stl")
    steps = 200
    setup_h = pp.setup_multilevel_domain(cfg, stl)
    setup_o = copy.deepcopy(setup_h)
    keep = {}

    def hip_factory(grids):
        keep["st"] = case.HipStepper(grids)
        keep["st"].close = lambda: None
        return keep["st"]

    hip, _, _ = case.run_case(cfg, hip_factory, steps=steps, setup=setup_h)
    ora, _, _ = case.run_case(cfg, OracleStepper, steps=steps, setup=setup_o)
    assert [r.step for r in hip] == [50, 100, 150, 200] == [r.step for r in ora]
    assert max(abs(r.cd) for r in ora) > 1e-5 and max(abs(r.cl) for r in ora) > 1e-6, "the start of the ramp must have loaded the wing"
    for a, b in zip(hip, ora):
        for name in ("cd", "cl", "cs", "cmy", "rho_min", "u_lat"):
            assert getattr(a, name) == getattr(b, name), (a.step, name, getattr(a, name), getattr(b, name))
    for i, g in enumerate(setup_o[0]):
        fn, vn = oracle.newest_buffers(i, steps)
        for name in ("rho", vn, fn):
            x, y = keep["st"].dev[i].download(name), getattr(g, name)
            assert np.isfinite(y).all()
            assert np.array_equal(x, y), (i + 1, name, int(np.count_nonzero(x != y)))
    # stepping rate of the device levels (state left as it is: 40 more coarse steps, timed)
    from open_ludwig_amd import execute_timestep_batch
    from open_ludwig_amd.preprocess import solver_params
    sp = solver_params(cfg, setup_h[2])
    dev = keep["st"].dev
    execute_timestep_batch(dev, steps + 1, 8, np.float32(cfg.u_lattice), sp)
    t0 = time.perf_counter()
    execute_timestep_batch(dev, steps + 9, 40, np.float32(cfg.u_lattice), sp)
    ms = (time.perf_counter() - t0) / 40 * 1e3
    work = sum(g.n_blocks * 512 * 2 ** i for i, g in enumerate(setup_h[0]))
    print(f"\nreal wing, 3 levels, {sum(g.n_blocks for g in setup_h[0]) * 512 / 1e6:.2f} M cells: {ms:.3f} ms per coarse step = "
          f"{work / ms / 1e3:.0f} M cell updates/s")
    for d in dev:
        d.close()


@pytest.mark.gpu
def test_real_wing_full_run_as_shipped(gpu):
    """BASELINE configs[4], "Cd/Cl/Cm convergence": the whole shipped run - 10 000 coarse steps, ramp over 2 000 - of the real wing
    (tools/run_wing.py, which also writes the series kept under profiles/). parity unpinned: the reference holds no wing log. What is
    asserted is that the run is a run: every row finite, rho_min bounded, the wing lifts and drags with the signs 5 degrees of
    incidence give, and the two halves of the last 2 000 steps agree within a quarter of the mean plus the scatter. (The series under
    profiles/ shows what that window is: at this resolution - 200 cells per 14 m, 3 levels instead of the shipped 1 100 and 5 - the
    loads are NOT converged after 10 000 steps; Cl is still falling.)"""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import run_wing
    a, summary = run_wing.run(diag_freq=100)
    print("\n" + json.dumps(summary))
    assert summary["rows"] == 100 and summary["all_finite"]
    assert 0.90 < summary["rho_min_over_run"] <= 1.0 + 1e-3
    tail = summary["last_2000_steps"]
    assert tail["Cd"]["mean"] > 0 and tail["Cl"]["mean"] > 0
    first, second = a[(a[:, 0] > 8000) & (a[:, 0] <= 9000)], a[a[:, 0] > 9000]
    for c in (3, 4, 6):      # Cd, Cl, Cmy: the two halves of the window agree within 25 % of the mean + the scatter
        assert abs(first[:, c].mean() - second[:, c].mean()) <= 0.25 * abs(a[a[:, 0] > 8000][:, c].mean()) + 2 * a[a[:, 0] > 8000][:, c].std()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_real_wing_on_ranks_equals_single_device(gpu, wing_real, tmp_path, world):
    """The REAL wing (its own STL, 3 levels, 4.65 M cells, per-level cuts) through run_case on 2 and 4 ranks sharing this box's GPU
    (gloo, host-staged messages): the finest level's rho and velocity and level 1's rho, gathered over the ranks, equal the
    single-device run bit for bit. The coefficients: per-rank Float32 partial sums added in rank order - another
    summation order of the same terms. At steps 8 ... 24 of a 40-step ramp the loads are a 1e-4 residue of pressure terms that
    cancel to three digits, so the rounding of the sums is 1e-6 of the TERMS and 5e-6 of the total: 2e-5 of the largest coefficient
    is asserted here, against 1e-6 where the loads are of the size of their terms (the sphere and synthetic-wing tests)."""
    import test_partition_dist as tpd
    cfg, stl = wing_real
    steps = 24
    keep = {}

    def hip_factory(grids):
        keep["st"] = case.HipStepper(grids)
        keep["st"].close = lambda: None
        return keep["st"]

    single, _, _ = case.run_case(cfg, hip_factory, steps=steps, setup=pp.setup_multilevel_domain(cfg, stl))
    want = {"rho2": keep["st"].dev[2].download("rho"), "vel2": keep["st"].dev[2].download("vel"), "rho0": keep["st"].dev[0].download("rho")}
    for d in keep["st"].dev:
        d.close()
    tpd._launch("gpu_case", tmp_path, (2, 0, 0), steps, world=world)
    rows = json.load(open(os.path.join(tmp_path, "rows.json")))["rows"]
    assert len(rows) == len(single) == 3
    scale = max(abs(r.cd) for r in single)
    assert scale > 1e-4
    for got, ref in zip(rows, single):
        assert got[:3] == [ref.step, ref.u_lat, ref.rho_min], (got, ref)
        for x, y in zip(got[3:], [ref.cd, ref.cl, ref.cs, ref.cmy]):
            assert abs(x - y) <= 2e-5 * scale, (got, ref)
    got = np.load(os.path.join(tmp_path, "fields.npz"))
    for name, arr in want.items():
        assert np.array_equal(got[name], arr), name
    stats = [json.load(open(os.path.join(tmp_path, f"stats{r}.json"))) for r in range(world)]
    for lvl in range(3):                 # every level cut on its own into near-equal parts (planar cuts at block granularity)
        owned = [s[lvl][0] for s in stats]
        assert sum(owned) == [2090, 1728, 5256][lvl] and max(owned) <= 1.15 * sum(owned) / world, owned


@pytest.mark.gpu
def test_wing_exactly_as_shipped_sets_up_and_steps(gpu):
    """CASES/Wing_5_deg/config.yaml with NO overrides - num_levels 5, surface_resolution 1100: 151 020 blocks, 77.3 M cells, 2.1 M Bouzidi
    cells, the finest level above the 32-bit address limit of a level (the WIDE instantiations). Native set-up, 24 coarse steps (24 G cell
    updates) through run_case: the set-up integers (pinned against the numpy restatement in profiles/r03_native_setup_vs_numpy.txt; the
    reference holds no wing log: parity unpinned), finite coefficients, a density that stays at rest level this early in the ramp.
    HIP = oracle at this size is a one-off (tests/oneoff_wing_shipped_oracle_check.py, profiles/r03_wing5deg_as_shipped_hip_equals_oracle.txt)."""
    cfg = pp.load_case_configuration(os.path.join(G, "wing5deg_config.yaml"))
    assert (cfg.num_levels, cfg.surface_resolution, cfg.steps, cfg.ramp_steps) == (5, 1100, 10000, 2000)
    cfg.diag_freq = 8
    setup = pp.setup_multilevel_domain(cfg, os.path.join(G, "wing5deg_model.stl"))
    rep = setup[3]
    assert rep.level_blocks == [5460, 2624, 8328, 28008, 106600] and rep.bouzidi_cells == [2109221]
    assert rep.flood_fill_filled == [890, 15530, 164099, 1488698, 12673840] and rep.near_wall_cells == [5146, 18661, 69950, 272616, 1073087]
    assert setup[0][-1].n_blocks * 512 * 27 * 4 > 2 ** 32, "the finest level's population array needs 64-bit offsets"
    rows, _, _ = case.run_case(cfg, case.HipStepper, steps=24, setup=setup)
    assert [r.step for r in rows] == [8, 16, 24]
    a = np.array([[r.rho_min, r.cd, r.cl, r.cmy] for r in rows])
    assert np.isfinite(a).all() and (a[:, 0] > 0.9999).all() and (a[:, 0] <= 1.0).all()
    assert np.abs(a[-1, 1:]).max() > 0, "the ramp has started: the coefficients are not identically zero"
