"""HIP path vs CPU oracle on identical inputs, through the C ABI (libludwig_hip.so).

Bar (BASELINE.json north_star): rho/u within 1e-5 relative, FP32. What we actually assert is stronger: the HIP
kernels keep the reference's operation order with contraction off, so every field is BIT-IDENTICAL to the oracle -
including the wall-model force since round 2: its pow / log are open_ludwig_amd/csrc/jl_math.h on both sides (the
shared source is checked against glibc in tests/test_jl_math.py).
"""
import numpy as np
import pytest

from open_ludwig_amd import adapt, cases, execute_timestep_batch
from oracle import oracle

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5   # north_star tolerance for floating point fields


def run_both(grids, params, steps, u, batch=None):
    dev = [adapt(g, 0) for g in grids]
    t = 1
    batch = batch or steps
    while t <= steps:
        n = min(batch, steps - t + 1)
        execute_timestep_batch(dev, t, n, np.float32(u), params)
        oracle.execute_timestep_batch(grids, t, n, np.float32(u), params)
        t += n
    return dev


def post_collision_blocks(g):
    """blocks that hold a Bouzidi cell or a cell adjacent to one: where f_post_collision has a reader"""
    m = np.zeros(g.n_blocks, dtype=bool)
    cb = g.bouzidi_cell_block.astype(np.int64) - 1
    xyz = [a.astype(np.int64) - 1 for a in (g.bouzidi_cell_x, g.bouzidi_cell_y, g.bouzidi_cell_z)]
    nt = np.asarray(g.neighbor_table)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                o = [np.where(c + d < 0, -1, np.where(c + d > 7, 1, 0)) for c, d in zip(xyz, (dx, dy, dz))]
                nb = nt[cb, (o[0] + 1) + 3 * (o[1] + 1) + 9 * (o[2] + 1)]
                m[nb[nb > 0] - 1] = True
    return m


def post_collision_rows(g):
    """bool [8,8,8,nb]: the x-rows (8 cells) that hold a cell f_post_collision is READ at by a link with q > 0 - the Bouzidi cell itself
    and the cell one step behind it along the link (src/bouzidi_kernel.jl:44-77). The library stores f_post_collision there and
    nowhere else (include/ludwig_hip.h: store_post_collision_everywhere)."""
    m = np.zeros((8, 8, 8, g.n_blocks), dtype=bool)
    cb = g.bouzidi_cell_block.astype(np.int64) - 1
    x, y, z = (a.astype(np.int64) - 1 for a in (g.bouzidi_cell_x, g.bouzidi_cell_y, g.bouzidi_cell_z))
    nt = np.asarray(g.neighbor_table)
    for k in range(27):
        q = g.bouzidi_q_map[x, y, z, cb, k].astype(np.float32)
        sel = q > 0
        if not sel.any():
            continue
        c = (k % 3 - 1, (k // 3) % 3 - 1, k // 9 - 1)
        b0, p0 = cb[sel], [x[sel], y[sel], z[sel]]
        m[:, p0[1], p0[2], b0] = True
        n = [p - d for p, d in zip(p0, c)]                     # x + c_opp(k)
        o = [np.where(v < 0, -1, np.where(v > 7, 1, 0)) for v in n]
        nb = np.where((o[0] == 0) & (o[1] == 0) & (o[2] == 0), b0 + 1, nt[b0, (o[0] + 1) + 3 * (o[1] + 1) + 9 * (o[2] + 1)])
        ok = nb > 0
        m[:, (n[1] % 8)[ok], (n[2] % 8)[ok], nb[ok] - 1] = True
    return m


def compare(grids, dev, steps, exact=True):
    for i, (g, d) in enumerate(zip(grids, dev)):
        fn, vn = oracle.newest_buffers(i, steps)
        names = [fn, vn, "rho"]
        if g.n_boundary_cells > 0:
            names.append("f_post_collision")
        if g.f_old.size > 27 and i < len(grids) - 1:
            names += ["f_old", "rho_old", "vel_old"]
        for name in names:
            a, b = d.download(name), getattr(g, name)
            assert np.isfinite(b).all(), f"oracle produced non-finite {name}"
            if name == "f_post_collision" and not getattr(g, "force_post_collision", False):
                # written only where it has a reader: the x-rows of the Bouzidi cells and of the cells behind their links (include/ludwig_hip.h)
                m = post_collision_rows(g)
                assert 0 < m.sum()
                a, b = a[m], b[m]
            if exact:
                bad = np.argwhere(a != b)
                assert bad.size == 0, (f"level {i + 1} {name}: {bad.shape[0]} elements differ, first at {bad[0]}, "
                                       f"hip {a[tuple(bad[0])]!r} oracle {b[tuple(bad[0])]!r}")
            else:
                err = np.abs(a.astype(np.float64) - b).max() / np.abs(b).max()
                assert err <= REL_TOL, f"level {i + 1} {name}: max rel err {err:.3e} > {REL_TOL}"
    for d in dev:
        d.close()


@pytest.mark.parametrize("nb,steps", [((4, 4, 4), 20), ((8, 8, 8), 6), ((2, 3, 5), 7)])
def test_periodic_box_bit_exact(gpu, nb, steps):
    """C1-style workload: periodic Taylor-Green box; all blocks take the all-neighbours kernel."""
    grids, params = cases.periodic_box(nb)
    dev = run_both(grids, params, steps, 0.0)
    assert dev[0].info().n_fast_blocks == grids[0].n_blocks
    compare(grids, dev, steps)


@pytest.mark.parametrize("nb", [(4, 4, 4), (8, 2, 3), (12, 2, 2), (5, 3, 2), (3, 2, 2)])
def test_periodic_box_rough_state_bit_exact(gpu, nb):
    """Non-smooth random state: every lane sees distinct values, so a wrong cross-lane move (DPP / shuffle / LDS
    column exchange of the x-run kernel) cannot hide. Widths 4, 8, 12 = whole x-runs; 5 = a run + a lone block (both faces
    from global memory), 3 = a short run sharing its workgroup with an idle wave."""
    grids, params = cases.periodic_box(nb)
    cases.init_perturbed(grids[0], 5)
    dev = run_both(grids, params, 3, 0.0)
    info = dev[0].info()
    linked_per_row = nb[0] - (1 if nb[0] % 4 == 1 else 0)          # blocks that hand a face column to a neighbouring wave
    assert info.n_xrun_blocks == linked_per_row * nb[1] * nb[2]
    compare(grids, dev, 3)


@pytest.mark.parametrize("order_name", ["block_planes", "pxcd_4x1_zxy", "prr_4x1_xyz", "pxcd_2x2_xyz", "cols_t44"])
def test_result_does_not_depend_on_launch_order(gpu, order_name):
    from open_ludwig_amd import order as order_mod
    grids, params = cases.periodic_box((8, 3, 2))
    cases.init_perturbed(grids[0], 9)
    dev = [adapt(g, 0) for g in grids]
    dev[0].set_order(order_mod.build(order_name, np.asarray(grids[0].active_block_coords)))
    execute_timestep_batch(dev, 1, 4, np.float32(0.0), params)
    oracle.execute_timestep_batch(grids, 1, 4, np.float32(0.0), params)
    compare(grids, dev, 4)


@pytest.mark.parametrize("merge", [None, "0", "1"])
def test_tunnel_interior_uses_x_runs(gpu, monkeypatch, merge):
    """A tunnel long enough that interior all-neighbour blocks form x-runs: the x-run kernel then runs with obstacle,
    sponge, Bouzidi (POST) and wall-model (WALL) work inside it. merge "0": all-neighbour and edge blocks in separate
    launches (what large levels do), "1" / default for this size: one launch, everything in the GENERAL instantiation."""
    if merge is None:
        monkeypatch.delenv("LUDWIG_MERGE_CLASSES", raising=False)
    else:
        monkeypatch.setenv("LUDWIG_MERGE_CLASSES", merge)
    grids, params = cases.tunnel_with_sphere((8, 4, 4), levels=1, wall_model=False)
    dev = run_both(grids, params, 3, 0.05)
    assert dev[0].info().n_xrun_blocks >= 16
    compare(grids, dev, 3)
    grids, params = cases.tunnel_with_sphere((8, 4, 4), levels=1, wall_model=True, tau=0.5003)
    dev = run_both(grids, params, 3, 0.05)
    assert dev[0].info().n_xrun_blocks >= 16
    compare(grids, dev, 3)


def test_periodic_box_bgk_only(gpu):
    """BASELINE configs[0] "BGK only": c_wale = 0 and nu_sgs_background = 0 (regularisation cannot be disabled, F9)."""
    grids, params = cases.periodic_box((4, 4, 4))
    params.c_wale = 0.0
    params.nu_sgs_bg = 0.0
    dev = run_both(grids, params, 10, 0.0)
    compare(grids, dev, 10)


@pytest.mark.parametrize("levels", [1, 2, 3])
@pytest.mark.parametrize("temporal", [True, False])
def test_tunnel_sphere_bit_exact(gpu, levels, temporal):
    """Inlet/outlet/mirror edges, obstacle bounce, sponge (+f blending), inlet noise, Bouzidi on the finest level,
    coarse->fine interpolation with and without temporal blending. No wall model -> bit-exact."""
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=levels, wall_model=False, temporal=temporal)
    steps = 3
    dev = run_both(grids, params, steps, 0.05)
    compare(grids, dev, steps)


def test_nested_levels_with_separate_launches_per_block_class(gpu, monkeypatch):
    """Same 3-level case with LUDWIG_MERGE_CLASSES=0: all-neighbour blocks in the plain x-run instantiation, edge blocks in
    the GENERAL one - the split large levels use."""
    monkeypatch.setenv("LUDWIG_MERGE_CLASSES", "0")
    grids, params = cases.tunnel_with_sphere((8, 4, 4), levels=3, wall_model=False, temporal=True)
    dev = run_both(grids, params, 3, 0.05)
    compare(grids, dev, 3)


def test_tunnel_symmetric_no_blend(gpu):
    grids, params = cases.tunnel_with_sphere((5, 3, 4), levels=2, symmetric=True, sponge_blend=False, inlet_turbulence=0.0)
    dev = run_both(grids, params, 3, 0.04)
    compare(grids, dev, 3)


def test_native_batch_driver_equals_python_recursion(gpu):
    """ludwig_execute_timestep_batch (row H inside the library) vs the call-by-call mirror of solver_control.jl."""
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=3)
    a = [adapt(g, 0) for g in grids]
    b = [adapt(g, 0) for g in grids]
    execute_timestep_batch(a, 1, 3, np.float32(0.05), params, native=True)
    execute_timestep_batch(b, 1, 3, np.float32(0.05), params, native=False)
    for i in range(3):
        for name in ("f", "f_temp", "vel", "vel_temp", "rho", "f_old", "rho_old", "vel_old"):
            if i == 2 and name.endswith("_old"):
                continue
            assert np.array_equal(a[i].download(name), b[i].download(name)), (i, name)
    for d in a + b:
        d.close()


def test_tunnel_batches_match_single_batch(gpu):
    """execute_timestep_batch! called in batches of 2 (async_depth) must give what one long batch gives."""
    grids, params = cases.tunnel_with_sphere((5, 3, 3), levels=2)
    dev = run_both(grids, params, 5, 0.05, batch=2)
    compare(grids, dev, 5)


@pytest.mark.parametrize("levels", [1, 2])
def test_tunnel_wall_model(gpu, levels):
    """Wall-model force: x^(1/7) and log come from the shared jl_math.h on both sides -> bit-identical like the rest."""
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=levels, wall_model=True, tau=0.5003)
    near = sum(int(((g.wall_dist > 0) & (g.wall_dist < 10)).sum()) for g in grids)
    assert near > 1000
    steps = 3
    dev = run_both(grids, params, steps, 0.05)
    compare(grids, dev, steps)


def test_saved_old_state_survives_steps_without_a_new_save(gpu):
    """copy_to_old! is served without copying f / vel (the step never writes its input buffers). The saved state must
    still read back as the reference's copy would - right after the save, after the step that follows it, and after a
    further step that overwrites the buffer it was aliased to (the library then gives it its own storage first)."""
    from open_ludwig_amd.physics import perform_timestep_v2
    grids, params = cases.tunnel_with_sphere((4, 2, 2), levels=1, wall_model=False, temporal=True)
    g = grids[0]
    cases.init_perturbed(g, seed=3)
    d = adapt(g, 0)
    saved_f, saved_v, saved_r = g.f.copy(), g.vel.copy(), g.rho.copy()      # t_sub = 2 is even: input buffers are f / vel
    d.copy_to_old(2)
    for name, want in (("f_old", saved_f), ("vel_old", saved_v), ("rho_old", saved_r)):
        assert np.array_equal(d.download(name), want), name
    perform_timestep_v2(d, None, np.float32(0.5), np.float32(0.04), params, 2, np.float32(0.0))
    assert np.array_equal(d.download("f_old"), saved_f) and np.array_equal(d.download("f"), saved_f)
    perform_timestep_v2(d, None, np.float32(0.5), np.float32(0.04), params, 3, np.float32(0.0))     # writes f: the aliased buffer
    assert not np.array_equal(d.download("f"), saved_f)
    for name, want in (("f_old", saved_f), ("vel_old", saved_v), ("rho_old", saved_r)):
        assert np.array_equal(d.download(name), want), name
    # an upload into the aliased buffer must not change the saved state either
    d.copy_to_old(4)
    now_f = d.download("f")
    d.upload("f", np.zeros_like(now_f))
    assert np.array_equal(d.download("f_old"), now_f) and not d.download("f").any()
    d.close()


def test_post_collision_store_modes(gpu, monkeypatch):
    """Default: f_post_collision is written only in the x-rows its only reader, src/bouzidi_kernel.jl:44-77, looks at (elsewhere the
    store is dead); LUDWIG_POST_ROWS=0: in whole blocks that hold or touch a Bouzidi cell (round 2); store_post_collision_everywhere
    restores the reference's full array. Either way the populations are the same."""
    results = []
    for mode in ("rows", "blocks", "everywhere"):
        if mode == "blocks":
            monkeypatch.setenv("LUDWIG_POST_ROWS", "0")
        else:
            monkeypatch.delenv("LUDWIG_POST_ROWS", raising=False)
        grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=2, wall_model=False, temporal=True)
        for g in grids:
            g.force_post_collision = mode == "everywhere"
        dev = run_both(grids, params, 3, 0.05)
        fin, d = grids[-1], dev[-1]
        mb, mr = post_collision_blocks(fin), post_collision_rows(fin)
        assert 0 < mb.sum() < fin.n_blocks and 0 < mr.sum() < mr[:, :, :, mb].size and not mr[:, :, :, ~mb].any()
        fp = d.download("f_post_collision")
        assert np.array_equal(fp[mr], fin.f_post_collision[mr])
        if mode == "everywhere":
            assert np.array_equal(fp, fin.f_post_collision)
        elif mode == "blocks":
            assert np.array_equal(fp[:, :, :, mb], fin.f_post_collision[:, :, :, mb])
            assert not fp[:, :, :, ~mb].any(), "blocks without a reader keep their initial zeros"
        else:
            assert not fp[~mr].any(), "rows without a reader keep their initial zeros"
        results.append(d.download("f"))
        compare(grids, dev, 3)
    assert np.array_equal(results[0], results[1]) and np.array_equal(results[0], results[2])


def test_negative_q_min_threshold_makes_every_direction_a_link(gpu):
    """`q > q_min && q <= 1` (src/bouzidi_kernel.jl:44) with a NEGATIVE threshold also takes the directions whose q is 0: every listed
    cell then reads f_post_collision at all its neighbours. The row-granular store only covers links with q > 0, so the library
    stores whole blocks for such a step (and refuses a correction whose threshold contradicts the store). Same bits as the oracle."""
    import dataclasses
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=2, wall_model=False, temporal=True)
    params = dataclasses.replace(params, q_min_threshold=-1.0)
    dev = run_both(grids, params, 3, 0.05)
    compare(grids, dev, 3)
    # stream-collide told q_min >= 0, correction asked with q_min < 0: an error, not wrong numbers
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=2, wall_model=False, temporal=True)
    dev = run_both(grids, params, 1, 0.05)
    from open_ludwig_amd import _lib
    lib = _lib.load()
    assert lib.ludwig_bouzidi_correction(dev[-1].handle, 1, -1.0) == -5          # LUDWIG_ERR_STATE
    for d in dev:
        d.close()


def test_interface_values_computed_ahead_are_dropped_when_the_parent_changes(gpu, monkeypatch):
    """The first sub-step of a pair also produces the interface values of the second one (same parent buffers, weight 0.5).
    If anything writes the parent in between, they must be recomputed: the sequence with an upload into the parent between
    the two child sub-steps gives the same child state as with the look-ahead switched off."""
    from open_ludwig_amd.physics import perform_timestep_v2
    results = []
    for ahead in (True, False):
        if ahead:
            monkeypatch.delenv("LUDWIG_NO_IFACE_AHEAD", raising=False)
        else:
            monkeypatch.setenv("LUDWIG_NO_IFACE_AHEAD", "1")
        grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=2, wall_model=False, temporal=True)
        for g in grids:
            cases.init_perturbed(g, seed=11 + g.level_id)
        dev = [adapt(g, 0) for g in grids]
        u = np.float32(0.05)
        dev[0].copy_to_old(2)
        perform_timestep_v2(dev[0], None, np.float32(0.5), u, params, 2, np.float32(0.0))                 # parent: writes f_temp
        perform_timestep_v2(dev[1], dev[0], dev[0].tau, u, params, 4, np.float32(0.0))                    # child, first of the pair
        newer = dev[0].download("f_temp")
        dev[0].upload("f_temp", (newer * np.float32(1.01)).astype(np.float32))                            # the parent changes
        perform_timestep_v2(dev[1], dev[0], dev[0].tau, u, params, 5, np.float32(0.5))                    # child, second of the pair
        results.append((dev[1].download("f"), dev[1].download("rho")))
        for d in dev:
            d.close()
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])


# ---- lazy rho: the step leaves `rho` unwritten when nobody reads it before the next step, and reproduces it on demand ----

def test_lazy_rho_is_what_the_step_would_have_stored(gpu, monkeypatch):
    """Single level (no child reads rho): the stores are elided, a download replays them. Checked against the oracle after
    1, 2 and 5 steps, with edge blocks, obstacle, sponge and Bouzidi in play, and against the eager library bit for bit."""
    from open_ludwig_amd.physics import perform_timestep_v2
    grids, params = cases.tunnel_with_sphere((6, 3, 3), levels=1, wall_model=True, tau=0.5003)
    dev = [adapt(g, 0) for g in grids]
    t = 1
    for n in (1, 1, 3):
        execute_timestep_batch(dev, t, n, np.float32(0.05), params)
        oracle.execute_timestep_batch(grids, t, n, np.float32(0.05), params)
        t += n
        assert np.array_equal(dev[0].download("rho"), grids[0].rho), f"after step {t - 1}"
    # an upload into the step's INPUT buffer must not change the answer: rho is produced before the upload lands
    execute_timestep_batch(dev, t, 1, np.float32(0.05), params)
    oracle.execute_timestep_batch(grids, t, 1, np.float32(0.05), params)
    f_in_name = "f" if t % 2 == 0 else "f_temp"
    dev[0].upload(f_in_name, np.zeros_like(grids[0].f))
    assert np.array_equal(dev[0].download("rho"), grids[0].rho)
    dev[0].close()
    # the same run with the stores forced on
    monkeypatch.setenv("LUDWIG_EAGER_RHO", "1")
    import subprocess, sys, os
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from open_ludwig_amd import adapt, cases, execute_timestep_batch;"
            "g, p = cases.tunnel_with_sphere((6, 3, 3), levels=1, wall_model=True, tau=0.5003); d = adapt(g[0], 0);"
            "execute_timestep_batch([d], 1, 6, np.float32(0.05), p); np.save(sys.argv[1], d.download('rho'))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rho.npy")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, LUDWIG_EAGER_RHO="1"))
        assert np.array_equal(np.load(out), grids[0].rho)


def test_lazy_rho_across_part_launches(gpu):
    """Multi-GPU style stepping (boundary part, then interior part): part launches always store rho (only whole-level launches
    elide it), so it can be read at any point - between the two part launches of a step it holds the new values where the step
    has run and the previous ones elsewhere, like the reference's array would. A whole-level step followed by part launches:
    the elided rho is produced before the first part launch reuses its inputs."""
    import ctypes as C
    from open_ludwig_amd import _lib
    lib = _lib.load()
    grids, params = cases.periodic_box((4, 2, 2))
    cases.init_perturbed(grids[0], 3)
    g = grids[0]
    g.comm_boundary = (np.asarray(g.map_x) == 1).astype(np.uint8)          # pretend the bx = 1 slab touches a peer
    d = adapt(g, 0)
    fl = params.to_c()
    _lib.check(lib.ludwig_stream_collide(d.handle, None, 1, 0.0, 0.5, 0.0, C.byref(fl), _lib.PART_ALL))       # rho elided
    oracle.execute_timestep_batch(grids, 1, 1, np.float32(0.0), params)
    rho1 = g.rho.copy()
    for part in (_lib.PART_BOUNDARY, _lib.PART_INTERIOR):
        _lib.check(lib.ludwig_stream_collide(d.handle, None, 2, 0.0, 0.5, 0.0, C.byref(fl), part))
    oracle.execute_timestep_batch(grids, 2, 1, np.float32(0.0), params)
    assert np.array_equal(d.download("rho"), g.rho)
    rho2 = g.rho.copy()
    for part in (_lib.PART_INTERIOR, _lib.PART_BOUNDARY):
        _lib.check(lib.ludwig_stream_collide(d.handle, None, 3, 0.0, 0.5, 0.0, C.byref(fl), part))
    _lib.check(lib.ludwig_stream_collide(d.handle, None, 4, 0.0, 0.5, 0.0, C.byref(fl), _lib.PART_INTERIOR))
    oracle.execute_timestep_batch(grids, 3, 1, np.float32(0.0), params)
    rho3 = g.rho.copy()
    oracle.execute_timestep_batch(grids, 4, 1, np.float32(0.0), params)
    mid = d.download("rho")                                                # between the two part launches of step 4
    slab = np.asarray(g.map_x) == 1
    assert np.array_equal(mid[..., slab], rho3[..., slab]) and np.array_equal(mid[..., ~slab], g.rho[..., ~slab])
    assert not np.array_equal(rho1, rho2) and not np.array_equal(rho3, g.rho)
    _lib.check(lib.ludwig_stream_collide(d.handle, None, 4, 0.0, 0.5, 0.0, C.byref(fl), _lib.PART_BOUNDARY))
    assert np.array_equal(d.download("rho"), g.rho)
    d.close()


def test_writes_through_a_raw_field_pointer_are_honoured(gpu):
    """ludwig_level_field_ptr hands out a writable device pointer. The library cannot see writes through it, so the level
    gives up the shortcuts that depend on seeing every write (aliased saved state, interface values computed one sub-step
    ahead, elided rho). Here the caller overwrites part of the PARENT's newest populations between the child's two sub-steps,
    through the pointer: the child's second sub-step must interpolate from the modified parent - checked against the same
    sequence with the modification made by ludwig_level_upload, which the library does see."""
    import ctypes as C
    from open_ludwig_amd.physics import perform_timestep_v2
    hip = None
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            hip = C.CDLL(name)
            break
        except OSError:
            continue
    assert hip is not None, "HIP runtime not loadable through ctypes"
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    u, t = np.float32(0.05), 1                                       # parent step 1 (odd): in = f_temp, out = f

    def first_half(raw_pointer):
        grids, params = cases.tunnel_with_sphere((5, 3, 3), levels=2)
        dev = [adapt(g, 0) for g in grids]
        ptr = dev[0].field_ptr("f")[0] if raw_pointer else None      # handed out BEFORE the steps: no call in between
        dev[0].copy_to_old(t)
        perform_timestep_v2(dev[0], None, np.float32(0.5), u, params, t, np.float32(0.0))
        perform_timestep_v2(dev[1], dev[0], dev[0].tau, u, params, 2 * t, np.float32(0.0))
        dev[0].synchronize()
        return grids, params, dev, ptr

    grids, params, dev, ptr = first_half(True)
    f_new = dev[0].download("f")
    order = dev[0].block_order()                                     # raw pointers show the library's own block order
    assert sorted(order) == list(range(grids[0].n_blocks)) and not np.array_equal(order, np.arange(grids[0].n_blocks))
    inv = np.argsort(order)                                          # internal position -> reference block
    patch_ref = f_new[..., 5] * np.float32(1.01)                     # (8,8,8,nb) in the reference order
    # the device array is block-major, [internal block][27][512] (ludwig_level_field_layout): population 5 of every block
    K, block_stride, comp_stride = dev[0].field_layout("f")
    assert (K, block_stride, comp_stride) == (27, 27 * 512, 512)
    nb0 = grids[0].n_blocks
    raw = np.empty(nb0 * block_stride, dtype=np.float32)
    assert hip.hipMemcpy(raw.ctypes.data, C.c_void_p(ptr), raw.nbytes, 2) == 0          # device -> host
    raw3 = raw.reshape(nb0, 27, 512)
    assert np.array_equal(raw3[:, 7, :], f_new[..., 7].reshape(512, nb0, order="F").T[inv])      # the layout is as documented
    raw3[:, 5, :] = patch_ref.reshape(512, nb0, order="F").T[inv]
    assert hip.hipMemcpy(C.c_void_p(ptr), raw.ctypes.data, raw.nbytes, 1) == 0          # host -> device
    perform_timestep_v2(dev[1], dev[0], dev[0].tau, u, params, 2 * t + 1, np.float32(0.5))
    got, got_old = dev[1].download("f"), dev[0].download("f_old")

    _, _, devb, _ = first_half(False)
    fb = devb[0].download("f")
    assert np.array_equal(fb, f_new)
    fb[..., 5] = patch_ref
    devb[0].upload("f", fb)
    perform_timestep_v2(devb[1], devb[0], devb[0].tau, u, params, 2 * t + 1, np.float32(0.5))
    assert np.array_equal(got, devb[1].download("f"))
    assert np.array_equal(got_old, devb[0].download("f_old"))      # the saved state is a real copy, untouched by the write

    _, _, devc, _ = first_half(False)                                # and the write did matter
    perform_timestep_v2(devc[1], devc[0], devc[0].tau, u, params, 2 * t + 1, np.float32(0.5))
    assert not np.array_equal(devc[1].download("f"), got)
    for d in dev + devb + devc:
        d.close()


def test_reference_behaviour_switches_give_the_same_bits(gpu, tmp_path):
    """The three places where the library deliberately does something else than the reference's arrays suggest - its own block
    order in device memory, one HIP stream per level inside a batch, the elided rho store - each have a switch back
    (LUDWIG_REFERENCE_BLOCK_ORDER, LUDWIG_BATCH_SERIAL, LUDWIG_EAGER_RHO). With all three set, in a fresh process, a 3-level
    wall-model tunnel gives exactly the fields of the default mode (and both equal the oracle's)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\\n"
            "from open_ludwig_amd import adapt, cases, execute_timestep_batch\\n"
            "g, p = cases.tunnel_with_sphere((6, 4, 4), levels=3, wall_model=True, tau=0.5003)\\n"
            "d = [adapt(x, 0) for x in g]\\n"
            "assert np.array_equal(d[0].block_order(), np.arange(g[0].n_blocks))\\n"
            "execute_timestep_batch(d, 1, 5, np.float32(0.05), p)\\n"
            "np.savez(sys.argv[1], **{f'{n}{i}': x.download(n) for i, x in enumerate(d) for n in ('f', 'f_temp', 'vel', 'vel_temp', 'rho')})\\n" % root)
    out = str(tmp_path / "ref_mode.npz")
    env = dict(os.environ, LUDWIG_REFERENCE_BLOCK_ORDER="1", LUDWIG_BATCH_SERIAL="1", LUDWIG_EAGER_RHO="1")
    subprocess.run([sys.executable, "-c", code.replace("\\n", "\n"), out], check=True, env=env)
    ref = np.load(out)
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=3, wall_model=True, tau=0.5003)
    dev = [adapt(g, 0) for g in grids]
    assert not np.array_equal(dev[0].block_order(), np.arange(grids[0].n_blocks))
    execute_timestep_batch(dev, 1, 5, np.float32(0.05), params)
    oracle.execute_timestep_batch(grids, 1, 5, np.float32(0.05), params)
    for i, (d, g) in enumerate(zip(dev, grids)):
        for n in ("f", "f_temp", "vel", "vel_temp", "rho"):
            a = d.download(n)
            assert np.array_equal(a, ref[f"{n}{i}"]), (i, n)
        fn, vn = oracle.newest_buffers(i, 5)
        for n in (fn, vn, "rho"):
            assert np.array_equal(d.download(n), getattr(g, n)), (i, n)
        d.close()


@pytest.mark.parametrize("hoist", ["0", "1"])
def test_interface_pass_ahead_of_the_wait_gives_the_same_bits(gpu, tmp_path, hoist):
    """Level streams: a level with a parent AND children may run its own interface pass before it waits for its children to have
    read its buffers (recursive_step, LUDWIG_IFACE_HOIST; by default a size rule decides). Forced on and forced off, in fresh
    processes, over two batches: the fields of all three levels equal the oracle's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from open_ludwig_amd import adapt, cases, execute_timestep_batch\n"
            "g, p = cases.tunnel_with_sphere((6, 4, 4), levels=3, wall_model=True, tau=0.5003)\n"
            "d = [adapt(x, 0) for x in g]\n"
            "execute_timestep_batch(d, 1, 3, np.float32(0.05), p)\n"
            "execute_timestep_batch(d, 4, 4, np.float32(0.05), p)\n"
            "np.savez(sys.argv[1], **{f'{n}{i}': x.download(n) for i, x in enumerate(d) for n in ('f', 'f_temp', 'vel', 'vel_temp', 'rho')})\n" % root)
    out = str(tmp_path / "hoist.npz")
    # round 3 computes a level's interface values on its PARENT's stream (no child-side pass left to hoist): LUDWIG_CHILD_SIDE_IFACE and
    # LUDWIG_RHO_OLD_COPY keep round 2's placement - the path this test is about - alive and compared against the default's oracle-equal bits
    subprocess.run([sys.executable, "-c", code.replace("\\n", "\n"), out], check=True,
                   env=dict(os.environ, LUDWIG_IFACE_HOIST=hoist, LUDWIG_CHILD_SIDE_IFACE="1", LUDWIG_RHO_OLD_COPY="1"))
    got = np.load(out)
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=3, wall_model=True, tau=0.5003)
    oracle.execute_timestep_batch(grids, 1, 7, np.float32(0.05), params)
    for i, g in enumerate(grids):
        fn, vn = oracle.newest_buffers(i, 7)
        for n in (fn, vn, "rho"):
            assert np.array_equal(got[f"{n}{i}"], getattr(g, n)), (i, n)


def test_stepping_stream_with_reserved_compute_units(gpu):
    """ludwig_stream_create: a stream whose kernels leave some compute units alone (for the halo exchange of the multi-GPU
    schedule). Stepping on it gives the bits of the default stream; more than half the device reserved is refused."""
    import ctypes as C
    from open_ludwig_amd import _lib
    lib = _lib.load()
    grids, params = cases.periodic_box((4, 3, 2))
    cases.init_perturbed(grids[0], 11)
    d = adapt(grids[0], 0)
    st = C.c_void_p()
    _lib.check(lib.ludwig_stream_create(0, 8, C.byref(st)))
    assert st.value
    d.set_stream(st.value)
    execute_timestep_batch([d], 1, 4, np.float32(0.0), params)
    oracle.execute_timestep_batch(grids, 1, 4, np.float32(0.0), params)
    fn, vn = oracle.newest_buffers(0, 4)
    for n in (fn, vn, "rho"):
        assert np.array_equal(d.download(n), getattr(grids[0], n)), n
    d.set_stream(None)
    d.close()
    _lib.check(lib.ludwig_stream_destroy(0, st))
    bad = C.c_void_p()
    assert lib.ludwig_stream_create(0, 100000, C.byref(bad)) == -1 and bad.value is None
    plain = C.c_void_p()
    _lib.check(lib.ludwig_stream_create(0, 0, C.byref(plain)))            # 0 reserved: an ordinary stream
    _lib.check(lib.ludwig_stream_destroy(0, plain))


@pytest.mark.parametrize("wide", [False, True])
def test_block_major_storage_is_invisible_at_the_abi(gpu, monkeypatch, wide):
    """The device arrays are block-major (ludwig_level_field_layout); everything that crosses the ABI keeps the reference's
    [8,8,8,n_blocks,K]. A 3-level wall-model tunnel with Bouzidi cells (every array, the q map, the parents' interpolation reads,
    upload / download) equals the oracle bit for bit, with 32-bit per-lane offsets and with the 64-bit addresses levels of 4 GiB and
    more use (forced here: LUDWIG_WIDE_ADDR); halo pack / unpack, given element offsets of the reference layout, address the same cells."""
    import ctypes as C
    import torch
    from open_ludwig_amd import _lib
    if wide:
        monkeypatch.setenv("LUDWIG_WIDE_ADDR", "1")
    grids, params = cases.tunnel_with_sphere((6, 4, 4), levels=3, wall_model=True, tau=0.5003)
    dev = run_both(grids, params, 4, 0.05, batch=3)
    for d in dev:
        assert d.field_layout("vel") == (3, 3 * 512, 512) and d.field_layout("rho") == (1, 512, 512)
        assert d.field_ptr("vel")[1] == 3 * 4 * 512 * d.n_blocks
    # pack / unpack take element offsets of the REFERENCE layout: pick population 5 of 100 cells scattered over level 2
    d, g = dev[1], grids[1]
    fn, _ = oracle.newest_buffers(1, 4)
    rng = np.random.default_rng(3)
    off = np.sort(rng.choice(g.n_blocks * 512, 100, replace=False)) + 5 * g.n_blocks * 512
    idx = torch.as_tensor(off, dtype=torch.int64, device="cuda")
    buf = torch.empty(100, dtype=torch.float32, device="cuda")
    lib = _lib.load()
    _lib.check(lib.ludwig_halo_pack(d.handle, _lib.FIELD_NAMES[fn], C.c_void_p(idx.data_ptr()), 100, C.c_void_p(buf.data_ptr()), None))
    d.synchronize()
    assert np.array_equal(buf.cpu().numpy(), getattr(g, fn).reshape(-1, order="F")[off])
    buf.mul_(2.0)
    torch.cuda.synchronize()
    _lib.check(lib.ludwig_halo_unpack(d.handle, _lib.FIELD_NAMES[fn], C.c_void_p(idx.data_ptr()), 100, C.c_void_p(buf.data_ptr()), None))
    d.synchronize()
    want = getattr(g, fn).copy(order="F")
    want.reshape(-1, order="F")[off] *= np.float32(2.0)
    assert np.array_equal(d.download(fn), want)
    d.upload(fn, getattr(g, fn))
    compare(grids, dev, 4)


def test_rho_min_propagates_nan_like_the_reference(gpu):
    """compute_flow_stats takes minimum(rho[.!obstacle]) (src/diagnostics.jl:71), and Julia's minimum propagates NaN: a diverged run
    must not print a finite rho_min. NaN in a fluid cell -> NaN; NaN in an obstacle cell is not counted."""
    grids, params = cases.tunnel_with_sphere((5, 3, 3), levels=1)
    g = grids[0]
    d = adapt(g, 0)
    rho = np.asfortranarray(1.0 + 0.01 * np.random.default_rng(1).random(g.rho.shape, dtype=np.float32))
    d.upload("rho", rho)
    assert d.rho_min() == float(rho[~g.obstacle].min())
    fluid = np.argwhere(~g.obstacle)[17]
    solid = np.argwhere(g.obstacle)[3]
    bad = rho.copy(order="F")
    bad[tuple(solid)] = np.nan
    d.upload("rho", bad)
    assert d.rho_min() == float(rho[~g.obstacle].min())
    bad[tuple(fluid)] = np.nan
    d.upload("rho", bad)
    assert np.isnan(d.rho_min())
    d.close()
