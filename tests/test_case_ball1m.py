"""External anchor: the reference's own run of CASES/ball1m at Re 266 667 (RESULTS_SPHERE_RE266K.txt, CUDA backend on an
RTX 3080), kept as data in tests/golden/sphere_re266k_*. YAML + STL go through this repo's pre-processing (row N1), the
HIP engine (rows A-I) and the surface forces (row N2).

  * setup integers must match the log exactly (CPU test),
  * the Cd / Cl / rho_min series of the first 2000 steps must match the log to its printed precision (4 decimals; the
    log is an FMA-contracting GPU run with Float32 atomics, so the last printed digit may differ by one or two),
  * HIP vs the CPU oracle on the same case: Cd within the north_star's 1e-5 relative (observed: identical).
"""
import json
import os
import sys

import numpy as np
import pytest

from open_ludwig_amd import case, preprocess as pp

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
RE266K = {"basic": {"surface_resolution": 25, "flow": {"velocity": 4.0}, "simulation": {"steps": 6000, "output_freq": 1000}}}


@pytest.fixture(scope="module")
def ball_setup():
    cfg = pp.load_case_configuration(os.path.join(G, "ball1m_config.yaml"), RE266K)
    return cfg, pp.setup_multilevel_domain(cfg, os.path.join(G, "ball1m.stl"))


def test_setup_matches_reference_log(ball_setup):
    cfg, (grids, mesh, params, rep) = ball_setup
    js = json.load(open(os.path.join(G, "sphere_re266k_setup.json")))
    assert mesh.triangles.shape[0] == 20480
    assert params.num_levels == 3 and [params.bx_max, params.by_max, params.bz_max] == js["level1_grid"]
    assert rep.level_blocks == js["level_blocks"]
    assert rep.halo_blocks_added == js["halo_blocks_added"]
    assert rep.flood_fill_filled == js["flood_fill_interior_voxels"]
    assert rep.bouzidi_cells == [js["bouzidi_boundary_cells_level3"]]
    assert round(100 * rep.sponge_fraction[0], 1) == round(100 * js["sponge_fraction_level1"], 1)
    assert round(rep.sponge_max[0], 3) == js["sponge_max_level1"]
    assert [f"{float(t):.6f}" for t in params.tau_levels] == [f"{t:.6f}" for t in js["tau_levels"]]
    assert [round(float(v), 2) for v in params.mesh_offset] == js["mesh_offset"]
    assert round(params.dx_fine, 6) == js["dx_fine"] and round(params.re_number) == js["reynolds"]
    assert round(params.rho_physical * params.velocity_scale ** 2, 2) == js["pressure_scale"]
    # near-wall counts in the log come from a racy counter (SURVEY section 4 caveat) - level 2's happens to be stable
    assert rep.near_wall_cells[1] == 1160
    fin = grids[-1]
    assert fin.bouzidi_enabled and fin.f_post_collision.shape[3] == 1728 and fin.bouzidi_q_map.dtype == np.float16
    q = fin.bouzidi_q_map.astype(np.float32)
    assert ((q >= 0) & (q <= 1)).all() and (q > 0).sum() > 5824


def _log_series(late=False):
    rows = [l.strip().split(",") for l in open(os.path.join(G, "sphere_re266k_log.csv")) if l[0].isdigit()]
    if late:
        rows += [l.strip().split(",") for l in open(os.path.join(G, "sphere_re266k_log_late.csv")) if l[0].isdigit()]
    return {int(r[0]): [float(v) for v in r[1:]] for r in rows}


@pytest.mark.gpu
def test_hip_reproduces_reference_cd_series(gpu, ball_setup):
    """The WHOLE run of the reference's log, all 30 rows to step 6000 (28.6 G cell updates, 2.8 s): the ramp (rows 200...2000) to the
    printed 4 decimals, and the developed phase (rows 2200...6000), where a contracting CUDA run and this one could have
    decorrelated but do not, within 8e-4 (observed <= 6e-4; profiles/r02_ball1m_re266k_6000_steps_vs_reference_log.txt). The log's
    final summary (Cd 0.447263, Cl -0.120969 at step 6000) is met to 5e-4 / 6e-4."""
    cfg, setup = ball_setup
    rows, _, _ = case.run_case(cfg, case.HipStepper, steps=6000, setup=pp.setup_multilevel_domain(cfg, os.path.join(G, "ball1m.stl")))
    log = _log_series(late=True)
    got = {r.step: r for r in rows}
    assert sorted(got) == sorted(log) and len(log) == 30
    for step, (u_lat, rho_min, cd, cl) in log.items():
        r = got[step]
        assert abs(r.u_lat - u_lat) <= 5.1e-5, step
        assert abs(r.rho_min - rho_min) <= 1.01e-4, (step, r.rho_min, rho_min)
        tol = 5e-4 if step == 200 else (2.01e-4 if step <= 2000 else 8e-4)
        assert abs(r.cd - cd) <= tol, (step, r.cd, cd)
        assert abs(r.cl - cl) <= tol, (step, r.cl, cl)
    assert abs(got[6000].cd - 0.447263) <= 8e-4 and abs(got[6000].cl + 0.120969) <= 8e-4


@pytest.mark.gpu
def test_oracle_itself_reproduces_reference_cd_series(gpu, ball_setup):
    """The oracle pinned DIRECTLY against the reference's own output, not through the HIP path: OracleStepper on ball1m at the
    log's settings for 600 coarse steps (2.9 G cell updates, ~1.5 min on the box's 16 cores - which is why this CPU-side
    test carries the gpu mark: it runs where those cores are; 1000 steps in round 2, shortened to keep the GPU suite inside the
    driver's limit: profiles/r02_* hold the 1000-step record), rows 200, 400, 600 of RESULTS_SPHERE_RE266K.txt to the printed
    digits. The same loop on the HIP path must give the SAME rows, bit for bit (wall model included: shared jl_math.h).
    Step 200: the oracle / HIP give Cd 0.0637, the log prints 0.0633. profiles/r02_step200_fma_contraction_experiment.txt
    shows why: the same oracle source built with -ffp-contract=fast (the log is a CUDA run, NVPTX fuses a*b+c) gives
    0.063311 at step 200 and the identical 0.145033 at step 400 (and 0.074350 against the kept forces.csv's 0.074373 for the
    4-level case, where the parity build gives 0.074838) - the difference is FMA contraction acting on forces of
    O(rho - 1) ~ 1e-5 in the first instants of the ramp, not a difference of algorithm."""
    from _steppers import OracleStepper
    from oracle import oracle
    oracle.set_num_threads(16)
    cfg, _ = ball_setup
    stl = os.path.join(G, "ball1m.stl")
    ora, _, _ = case.run_case(cfg, OracleStepper, steps=600, setup=pp.setup_multilevel_domain(cfg, stl))
    log = _log_series()
    got = {r.step: r for r in ora}
    assert sorted(got) == [200, 400, 600]
    for step, r in got.items():
        u_lat, rho_min, cd, cl = log[step]
        assert abs(r.u_lat - u_lat) <= 5.1e-5 and abs(r.rho_min - rho_min) <= 1.01e-4, (step, r)
        assert abs(r.cd - cd) <= (5e-4 if step == 200 else 2.01e-4), (step, r.cd, cd)
        assert abs(r.cl - cl) <= 2.01e-4, (step, r.cl, cl)
    assert abs(got[200].cd - 0.063654) < 2e-6                      # the contraction-off value (see the experiment file)
    hip, _, _ = case.run_case(cfg, case.HipStepper, steps=600, setup=pp.setup_multilevel_domain(cfg, stl))
    for a, b in zip(hip, ora):
        assert (a.step, a.u_lat, a.rho_min, a.cd, a.cl) == (b.step, b.u_lat, b.rho_min, b.cd, b.cl), (a, b)


# ---- BASELINE configs[0]: cube1m as a single-level ~64^3 case, "BGK only" (c_wale = 0, nu_sgs_background = 0) ----
CUBE = {"basic": {"num_levels": 1, "surface_resolution": 7},
        "advanced": {"boundary": {"method": "bounce_back"}, "high_re": {"wall_model": {"enabled": False}},
                     "numerics": {"c_wale": 0.0, "nu_sgs_background": 0.0}, "diagnostics": {"freq": 16}}}


def test_cube1m_setup_is_the_72x64x64_tunnel():
    """num_levels 1 + surface_resolution 7 gives a 72 x 64 x 64 level-1 tunnel (9.25 x 8.5 x 8.5 m domain)."""
    cfg = pp.load_case_configuration(os.path.join(G, "cube1m_config.yaml"), CUBE)
    grids, mesh, params, rep = pp.setup_multilevel_domain(cfg, os.path.join(G, "cube1m.stl"))
    assert mesh.triangles.shape[0] == 12 and params.num_levels == 1
    assert (params.nx_coarse, params.ny_coarse, params.nz_coarse) == (72, 64, 64) and rep.level_blocks == [576]
    assert not grids[0].bouzidi_enabled and int(grids[0].obstacle.sum()) == 576       # SAT shell (0.75 dx half-box) + 180 filled
    assert (grids[0].wall_dist == 100).all()


@pytest.mark.gpu
def test_cube1m_hip_equals_oracle(gpu):
    from _steppers import OracleStepper
    from oracle import oracle
    oracle.set_num_threads(16)
    cfg = pp.load_case_configuration(os.path.join(G, "cube1m_config.yaml"), CUBE)
    stl = os.path.join(G, "cube1m.stl")
    hip, _, _ = case.run_case(cfg, case.HipStepper, steps=48, setup=pp.setup_multilevel_domain(cfg, stl))
    ora, _, _ = case.run_case(cfg, OracleStepper, steps=48, setup=pp.setup_multilevel_domain(cfg, stl))
    assert [r.step for r in hip] == [16, 32, 48] == [r.step for r in ora]
    for a, b in zip(hip, ora):
        assert a.rho_min == b.rho_min and a.cd == b.cd and a.cl == b.cl, (a, b)      # bounce-back, no wall model: exact


@pytest.mark.gpu
def test_ball1m_on_two_ranks_equals_single_device(gpu, ball_setup, tmp_path):
    """Scope row N3 end to end: the same case through run_case on 2 ranks (cut through all three levels, wall model on,
    Bouzidi sphere split between the ranks). The fields are stepped bit-identically (tests/test_partition_dist.py); the
    diagnostics cross ranks as scalars only - rho_min by all-reduce MIN (exact), the nine force sums as per-rank partial sums
    added in rank order, so Cd / Cl equal the single-device rows up to the Float32 rounding of another summation
    order: 1e-6 of the largest coefficient is asserted."""
    import copy
    import json
    import test_partition_dist as tpd
    cfg, setup = ball_setup
    cfg = copy.copy(cfg)
    cfg.diag_freq = 20
    steps = 60
    single, _, _ = case.run_case(cfg, case.HipStepper, steps=steps, setup=setup)
    tpd._launch("gpu_case", tmp_path, (0, 0, 0), steps, world=2)
    rows = json.load(open(os.path.join(tmp_path, "rows.json")))["rows"]
    assert len(rows) == len(single) == 3
    scale = max(abs(r.cd) for r in single)
    for got, want in zip(rows, single):
        assert got[0] == want.step
        assert got[1:3] == [want.u_lat, want.rho_min], (got, want)
        assert abs(got[3] - want.cd) <= 1e-6 * scale and abs(got[4] - want.cl) <= 1e-6 * scale, (got, want)
    stats = [json.load(open(os.path.join(tmp_path, f"stats{r}.json"))) for r in range(2)]
    for lvl in range(3):
        assert sum(s[lvl][0] for s in stats) == setup[0][lvl].n_blocks
        assert all(s[lvl][1] > s[lvl][0] for s in stats), "every level was meant to be cut"


@pytest.mark.gpu
def test_ball1m_result_files(gpu, ball_setup, tmp_path):
    """Scope row N4 through run_case: convergence.csv / forces.csv rows at the diagnostics steps, flow + surface VTU at the
    output steps; the flow file holds the leaf blocks of all three levels and the density the diagnostics saw."""
    import copy
    from test_output_files import read_vtu
    from open_ludwig_amd import output
    cfg, setup = ball_setup
    cfg = copy.copy(cfg)
    cfg.diag_freq, cfg.output_freq = 16, 32
    rows, _, params = case.run_case(cfg, case.HipStepper, steps=48, setup=setup, out_dir=str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["convergence.csv", "flow_000032.vtu", "forces.csv", "surface_000032.vtu"]
    conv = open(tmp_path / "convergence.csv").read().splitlines()
    frc = open(tmp_path / "forces.csv").read().splitlines()
    assert conv[0] == output.CONVERGENCE_CSV_HEADER and frc[0] == output.FORCE_CSV_HEADER and len(conv) == len(frc) == 4
    for r, c, f in zip(rows, conv[1:], frc[1:]):
        cc, ff = c.split(","), f.split(",")
        assert int(cc[0]) == int(ff[0]) == r.step and cc[6] == "%.4f" % r.cd and cc[7] == "%.4f" % r.cl
        assert ff[11] == "%.6f" % r.cd and float(ff[1]) == pytest.approx(r.step * params.time_scale, rel=1e-6)
        assert np.float32(cc[4]) == np.float32(r.rho_min)
    grids = setup[0]
    d = read_vtu(str(tmp_path / "flow_000032.vtu"))
    sel = output.select_export_blocks([g.active_block_coords for g in grids])
    assert d["n_cells"] == 512 * len(sel) and set(np.unique(d["Level"])) == {1, 2, 3}
    assert int((d["Obstacle"] == 1).sum()) == sum(int(grids[l].obstacle[:, :, :, b].sum()) for l, b in sel)
    s = read_vtu(str(tmp_path / "surface_000032.vtu"))
    assert s["n_cells"] == 20480 and s["MappingQuality"].mean() > 0.9 and np.isfinite(s["Pressure_Pa"]).all()


# ---- second external anchor: CASES/ball1m as shipped (Re 9.87 M, N = 55), the run whose outputs the reference keeps ----
RE10M = {"basic": {"num_levels": 4}}      # the log of that run says "4 levels" (RESULTS_SPHERE_RE10M.txt:51)


@pytest.fixture(scope="module")
def ball_re10m_setup():
    cfg = pp.load_case_configuration(os.path.join(G, "ball1m_config.yaml"), RE10M)
    return cfg, pp.setup_multilevel_domain(cfg, os.path.join(G, "ball1m.stl"))


def test_re10m_setup_matches_reference_log(ball_re10m_setup):
    cfg, (grids, mesh, params, rep) = ball_re10m_setup
    js = json.load(open(os.path.join(G, "sphere_re10m_setup.json")))
    assert params.num_levels == 4 and [params.bx_max, params.by_max, params.bz_max] == js["level1_grid"]
    assert rep.level_blocks == js["level_blocks"] and rep.halo_blocks_added[1:] == js["halo_blocks_added"]
    assert rep.flood_fill_filled == js["flood_fill_interior_voxels"]
    assert rep.bouzidi_cells == [js["bouzidi_boundary_cells_level4"]]
    assert [round(100 * v, 1) for v in rep.sponge_fraction] == js["sponge_percent"]
    assert [round(v, 3) for v in rep.sponge_max] == js["sponge_max"]
    assert rep.near_wall_cells[1] == js["near_wall_cells_level2"]          # the other levels' counts come from a racy counter
    assert [f"{float(t):.6f}" for t in params.tau_levels] == js["tau_levels"]
    assert [round(float(v), 3) for v in params.mesh_offset] == js["mesh_offset"]
    assert round(params.dx_fine, 6) == js["dx_fine"]
    assert "%.6e" % (params.rho_physical * params.velocity_scale ** 2) == "2.981378e+07"
    assert round(sum(g.n_blocks for g in grids) * 512 / 1e6, 2) == js["total_cells_millions"]


@pytest.mark.gpu
def test_hip_reproduces_reference_forces_csv_re10m(gpu, ball_re10m_setup, tmp_path):
    """The reference's kept forces.csv (CUDA run, FMA contraction, Float32 atomics) against this engine + forces + CSV
    writer, steps 400...4000: drag force to 5e-4 relative (observed <= 3e-4, mostly ~5e-5), its viscous part to 1e-3, Cd to
    1e-4 absolute during the ramp and 1.5e-4 after it, side / lift coefficients to 1e-4 / 2e-4; physical time and inlet speed columns
    exactly. At this Reynolds number (tau_fine 0.500001) the two runs do separate in the end: 3.5e-4 at step 4200, 2.6e-3 at 4800,
    1e-1 at 5800 (profiles/r02_ball1m_re10m_6000_steps_vs_reference_forces_csv.txt) - the chaotic divergence SURVEY section 4 predicted."""
    import copy
    cfg, setup = ball_re10m_setup
    cfg = copy.copy(cfg)
    cfg.output_freq = 10 ** 9                         # no VTU here: the flow file of 3.9 M cells is ~100 MB
    last = 4000
    rows, _, params = case.run_case(cfg, case.HipStepper, steps=last, setup=setup, out_dir=str(tmp_path))
    ref = {int(l.split(",")[0]): l.strip().split(",") for l in open(os.path.join(G, "sphere_re10m_forces.csv")) if l[0].isdigit()}
    mine = {int(l.split(",")[0]): l.strip().split(",") for l in open(tmp_path / "forces.csv") if l[0].isdigit()}
    assert sorted(mine) == list(range(200, last + 1, 200))
    for s in range(400, last + 1, 200):
        a, b = mine[s], ref[s]
        assert a[1] == b[1] and a[2] == b[2], "Time_s / U_inlet columns"
        fx, fxr = float(a[3]), float(b[3])
        assert abs(fx - fxr) <= 5e-4 * abs(fxr), (s, fx, fxr)
        assert abs(float(a[7]) - float(b[7])) <= 1e-3 * abs(float(b[7])), (s, "Fx_v")
        tol = 1e-4 if s <= 2000 else 2e-4
        assert abs(float(a[11]) - float(b[11])) <= (1e-4 if s <= 2000 else 1.5e-4), (s, "Cd", a[11], b[11])
        for col in (12, 13):                           # Cl, Cs
            assert abs(float(a[col]) - float(b[col])) <= tol, (s, col, a[col], b[col])
    conv = {int(l.split(",")[0]): l.strip().split(",") for l in open(os.path.join(G, "sphere_re10m_convergence.csv")) if l[0].isdigit()}
    minec = {int(l.split(",")[0]): l.strip().split(",") for l in open(tmp_path / "convergence.csv") if l[0].isdigit()}
    for s in range(400, last + 1, 200):
        assert minec[s][2] == conv[s][2] and minec[s][3] == conv[s][3], "Time_phys_s / U_inlet_lat text"
        assert abs(float(minec[s][4]) - float(conv[s][4])) <= (2e-6 if s <= 2000 else 2e-5), ("rho_min", s, minec[s][4], conv[s][4])
        ctol = 1.5e-4 if s <= 2000 else 2.5e-4
        assert abs(float(minec[s][6]) - float(conv[s][6])) <= ctol and abs(float(minec[s][7]) - float(conv[s][7])) <= ctol, (s, minec[s], conv[s])


# ---- third log: same mesh at Re 986 667 (velocity 14.8 m/s) ----
@pytest.mark.gpu
def test_hip_reproduces_reference_cd_series_re1m(gpu):
    """RESULTS_SPHERE_RE1M.txt: the 3-level mesh of the Re 266 k run at 3.7x the speed (tau_fine 0.500002). Cd / Cl / rho_min
    of steps 200...2000 (the ramp) to the log's 4 printed decimals (+-2 in the last digit; the log is a CUDA run), 2200...6000
    (the first two shedding cycles) within 1e-3; observed 6.7e-4. Past step ~7200 the two runs decorrelate - the whole 12 000-step
    table is profiles/r02_ball1m_re1m_12000_steps_vs_reference_log.txt."""
    cfg = pp.load_case_configuration(os.path.join(G, "ball1m_config.yaml"),
                                     {"basic": {"surface_resolution": 25, "num_levels": 3, "flow": {"velocity": 14.8}}})
    setup = pp.setup_multilevel_domain(cfg, os.path.join(G, "ball1m.stl"))
    assert [f"{float(t):.6f}" for t in setup[2].tau_levels] == ["0.500009", "0.500005", "0.500002"]      # log line 103
    assert "%.2f" % np.float32(setup[2].rho_physical * setup[2].velocity_scale ** 2) == "298137.78"      # log line 162 (Float32)
    cfg.diag_freq = 200
    rows, _, _ = case.run_case(cfg, case.HipStepper, steps=6000, setup=setup)
    log = {}
    for name in ("sphere_re1m_log.csv", "sphere_re1m_log_late.csv"):
        log.update({int(l.split(",")[0]): [float(v) for v in l.split(",")[1:]] for l in open(os.path.join(G, name)) if l[0].isdigit()})
    log = {s: v for s, v in log.items() if s <= 6000}
    got = {r.step: r for r in rows}
    assert sorted(got) == sorted(log) and len(log) == 30
    for step, (u_lat, rho_min, cd, cl) in log.items():
        r = got[step]
        assert abs(r.u_lat - u_lat) <= 5.1e-5 and abs(r.rho_min - rho_min) <= 1.01e-4, step
        # ramp (<= 2000): last printed digit; after it the wake goes unsteady and the CUDA run's rounding shows (observed 2.7e-4)
        tol = 5e-4 if step == 200 else (2.01e-4 if step <= 2000 else 1e-3)
        assert abs(r.cd - cd) <= tol, (step, r.cd, cd)
        assert abs(r.cl - cl) <= tol, (step, r.cl, cl)


def test_exported_cell_counts_match_reference_logs(ball_setup, ball_re10m_setup):
    """'VTK Export END (1.42M cells' (RESULTS_SPHERE_RE266K.txt:167, RE1M:181) and '(3.46M cells' (RE10M:202): the leaf-block
    rule of the flow export (row N4) on the two meshes."""
    from open_ludwig_amd import output
    for (cfg, setup), want in ((ball_setup, "1.42"), (ball_re10m_setup, "3.46")):
        sel = output.select_export_blocks([g.active_block_coords for g in setup[0]])
        assert "%.2f" % (len(sel) * 512 / 1e6) == want


@pytest.mark.gpu
def test_device_stress_mapping_equals_host_mapping(gpu, ball_setup):
    """ludwig_map_surface_stresses (one thread per triangle on the device) against the numpy restatement of
    map_stresses_kernel! on the downloaded fields of the same state: all four per-triangle arrays bit-identical, hence the
    same Cd / Cl; also with a search radius too small to reach fluid for some triangles, and on the other velocity buffer."""
    from open_ludwig_amd import forces
    cfg, (grids, mesh, params, rep) = ball_setup
    sp = pp.solver_params(cfg, params)
    st = case.HipStepper(grids)
    st.batch(1, 120, np.float32(0.02), sp)
    fin = len(grids) - 1
    rho, vel, vel_t = st.field(fin, "rho"), st.field(fin, "vel"), st.field(fin, "vel_temp")
    for radius, vname, v in ((5, "vel", vel), (0, "vel", vel), (1, "vel", vel), (2, "vel_temp", vel_t)):
        want = forces.map_surface_stresses(mesh, rho, v, grids[fin].obstacle, grids[fin].block_pointer, grids[fin].dx, grids[fin].tau, params, radius)
        got = forces.map_surface_stresses_device(mesh, st.dev[fin], grids[fin].dx, grids[fin].tau, params, radius, vname)
        for name, a, b in zip(("p", "tau_x", "tau_y", "tau_z"), got, want):
            assert np.array_equal(a, b), (radius, vname, name, int((a != b).sum()))
        if radius == 5:
            assert np.count_nonzero(got[0]) == mesh.centers.shape[0] and np.abs(got[1]).max() > 0
        if radius == 0:
            assert not got[0].any(), "every triangle centre lies in a voxelised (solid) cell: radius 0 maps nothing"
        if radius == 1:
            assert 0 < np.count_nonzero(got[0] == 0) < mesh.centers.shape[0], "radius 1 reaches fluid for some triangles only"
    fr_dev = case._aerodynamics(st, grids, mesh, params, False)
    fr_host = forces.compute_aerodynamics(mesh, grids[fin], rho, vel, params, False)
    assert fr_dev.Cd == fr_host.Cd and fr_dev.Cl == fr_host.Cl and fr_dev.Fx_viscous == fr_host.Fx_viscous
    st.close()


@pytest.mark.gpu
def test_hip_matches_oracle_on_ball1m_re10m(gpu, ball_re10m_setup):
    """The 4-level case (tau_fine 0.500001, sponge reaching level 2, 28 400 Bouzidi cells, wall model): 32 coarse steps
    (650 M cell updates) on HIP and on the CPU oracle; every level's rho / vel / f and the Cd row identical."""
    import copy
    from _steppers import OracleStepper
    from oracle import oracle
    oracle.set_num_threads(16)
    cfg, _ = ball_re10m_setup
    cfg = copy.copy(cfg)
    cfg.diag_freq, cfg.output_freq, steps = 32, 10 ** 9, 32
    stl = os.path.join(G, "ball1m.stl")
    setup_h = pp.setup_multilevel_domain(cfg, stl)
    setup_o = copy.deepcopy(setup_h)
    keep = {}

    def hip_factory(grids):
        keep["st"] = case.HipStepper(grids)
        keep["st"].close = lambda: None              # keep the device levels for the field comparison below
        return keep["st"]

    hip, _, _ = case.run_case(cfg, hip_factory, steps=steps, setup=setup_h)
    ora, _, _ = case.run_case(cfg, OracleStepper, steps=steps, setup=setup_o)
    assert len(hip) == len(ora) == 1
    assert hip[0].cd == ora[0].cd and hip[0].rho_min == ora[0].rho_min
    for i, g in enumerate(setup_o[0]):
        fn, vn = oracle.newest_buffers(i, steps)
        for name in ("rho", vn, fn):
            a, b = keep["st"].dev[i].download(name), getattr(g, name)
            assert np.array_equal(a, b), (i + 1, name, int(np.count_nonzero(a != b)))
    for d in keep["st"].dev:
        d.close()
