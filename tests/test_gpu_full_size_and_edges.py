"""(a) BASELINE full sizes: 256^3 (configs[1]) HIP vs the CPU oracle bit for bit (the oracle steps 16.8 M cells in
about half a second per step on the box's cores), 64^3 for 100 steps (SURVEY section 7 minimum slice), and 256^3 / 512^3
(configs[3] on one GPU) through size-independent properties: mass / momentum conservation on the periodic box,
independence of the launch order, uniform state = fixed point. (b) edge cases of the C ABI: empty level, single block,
wrong sizes, invalid orders, missing storage."""
import ctypes as C

import numpy as np
import pytest

from open_ludwig_amd import _lib, adapt, cases, execute_timestep_batch, order as order_mod
from open_ludwig_amd.blocks import BlockLevel, build_neighbor_table

pytestmark = pytest.mark.gpu


def test_256_cubed_bit_exact_vs_oracle(gpu):
    """BASELINE configs[1] at its own size: HIP == oracle on rho, u and all 27 populations after 3 steps (odd: newest state
    in f / vel). The start is the bench's Taylor-Green field roughened by a deterministic per-cell perturbation so that
    every lane of every wave carries a distinct value."""
    grids, params = cases.periodic_box((32, 32, 32))
    g = grids[0]
    rng = np.random.default_rng(256)
    for k in range(27):                                            # in place, one population at a time (1.8 GB array)
        g.f[..., k] *= (1.0 + 0.01 * rng.standard_normal(g.rho.shape, dtype=np.float32))
    g.f_temp[...] = g.f
    steps = 3
    d = adapt(g, 0)
    assert d.info().n_xrun_blocks == g.n_blocks
    execute_timestep_batch([d], 1, steps, np.float32(0.0), params)
    from oracle import oracle
    oracle.execute_timestep_batch(grids, 1, steps, np.float32(0.0), params)
    fn, vn = oracle.newest_buffers(0, steps)
    for name in ("rho", vn, fn):
        a, b = d.download(name), getattr(g, name)
        assert np.isfinite(b).all()
        assert np.array_equal(a, b), f"256^3 {name}: {np.count_nonzero(a != b)} elements differ from the oracle"
        del a
    assert g.rho.std() > 0
    d.close()


def test_64_cubed_100_steps_bit_exact_vs_oracle(gpu):
    """SURVEY section 7 minimum slice / 8d C1: 64^3 periodic Taylor-Green box, 100 steps, every field identical."""
    grids, params = cases.periodic_box((8, 8, 8))
    g = grids[0]
    d = adapt(g, 0)
    from oracle import oracle
    t = 1
    for n in (1, 32, 33, 34):                                      # uneven batches: the A/B parity follows t, not the batch
        execute_timestep_batch([d], t, n, np.float32(0.0), params)
        oracle.execute_timestep_batch(grids, t, n, np.float32(0.0), params)
        t += n
    steps = t - 1
    assert steps == 100
    fn, vn = oracle.newest_buffers(0, steps)
    for name in ("rho", vn, fn):
        assert np.array_equal(d.download(name), getattr(g, name)), name
    u = getattr(g, vn)
    assert 0.005 < np.abs(u).max() < 0.03                           # the vortex decays but is still there
    d.close()


def test_512_cubed_single_gpu_properties(gpu):
    """BASELINE configs[3] on ONE GPU (134 M cells, 262 144 blocks, 29 GB of f + f_temp): 32-bit byte offsets, work lists and
    launch order at 8 x the headline size. Checked through properties: finite, mass and momentum conserved, and the
    64^3-periodic start makes the field 8-fold periodic in every direction - the step must keep that symmetry exactly,
    block (bx,by,bz) == block (bx+8k, by+8l, bz+8m) bit for bit - which compares 512 independently computed copies."""
    nb = 64
    small, params = cases.periodic_box((8, 8, 8))
    s = small[0]
    coords = cases.full_box_coords(nb, nb, nb)
    table = build_neighbor_table(coords, nb, nb, nb, (True, True, True))
    big = BlockLevel(1, coords, table, 1.0, 1.0, s.tau, enable_temporal_interpolation=False)   # zero pages, never touched
    d = adapt(big, 0)
    del big
    assert d.info().n_xrun_blocks == nb ** 3
    # tile the 64^3 state: block (bx,by,bz) of the big box takes block (bx%8, by%8, bz%8) of the small one
    c = np.asarray(coords) - 1
    src = ((c[:, 0] % 8) * 8 + (c[:, 1] % 8)) * 8 + (c[:, 2] % 8)
    for name in ("f", "f_temp", "vel", "vel_temp", "rho"):
        d.upload(name, np.asfortranarray(getattr(s, name)[:, :, :, src]))
    steps = 4
    execute_timestep_batch([d], 1, steps, np.float32(0.0), params)
    dsmall = adapt(s, 0)
    execute_timestep_batch([dsmall], 1, steps, np.float32(0.0), params)
    from oracle import oracle
    oracle.execute_timestep_batch(small, 1, steps, np.float32(0.0), params)      # and the 64^3 box is the oracle's
    for name in ("rho", "vel_temp", "f_temp"):
        a = d.download(name)
        b = dsmall.download(name)
        assert np.array_equal(b, getattr(s, name))
        assert np.isfinite(a).all()
        for blk in range(0, nb ** 3, 4096):                        # chunked: no second 14.5 GB temporary
            sl = slice(blk, blk + 4096)
            assert np.array_equal(a[:, :, :, sl], b[:, :, :, src[sl]]), f"512^3 {name}: the 8-fold periodic copies differ"
        del a, b
    d.close(); dsmall.close()


def test_256_cubed_conservation_and_order_independence(gpu):
    nb = (32, 32, 32)
    grids, params = cases.periodic_box(nb)
    g = grids[0]
    coords = np.asarray(g.active_block_coords)
    rho0 = g.rho.astype(np.float64).sum()
    mom0 = [(g.rho.astype(np.float64) * g.vel[..., c]).sum() for c in range(3)]
    a = adapt(g, 0)
    b = adapt(g, 0)
    b.set_order(order_mod.build("block_planes", coords))           # block order: no wave has its x neighbour next to it
    assert a.info().n_xrun_blocks == g.n_blocks and b.info().n_xrun_blocks == 0
    del grids
    steps = 6
    execute_timestep_batch([a], 1, steps, np.float32(0.0), params)
    execute_timestep_batch([b], 1, steps, np.float32(0.0), params)
    ra, rb = a.download("rho"), b.download("rho")
    va, vb = a.download("vel_temp"), b.download("vel_temp")       # even number of steps -> newest in vel_temp
    assert np.array_equal(ra, rb) and np.array_equal(va, vb), "x-run kernel and per-wave kernel disagree at 256^3"
    assert np.isfinite(ra).all() and ra.std() > 0
    # periodic box, no force, no sponge: sum(rho) and sum(rho u) are invariants of collide-and-stream
    assert abs(ra.astype(np.float64).sum() / rho0 - 1.0) < 2e-7
    for c in range(3):
        m = (ra.astype(np.float64) * va[..., c]).sum()
        assert abs(m - mom0[c]) < 1e-6 * ra.size * 0.03            # |u| <= 0.03: absolute scale of the momentum sums
    # checksum of the distributions of the two runs
    fa, fb = a.download("f_temp"), b.download("f_temp")
    assert np.array_equal(fa, fb)
    a.close(); b.close()


def test_uniform_state_is_fixed_point_at_full_size(gpu):
    grids, params = cases.periodic_box((32, 32, 32), init=False)
    g = grids[0]
    cases.set_state(g, np.float32(1.0), np.float32(0.04), np.float32(-0.01), np.float32(0.02))
    f0 = g.f[:, :, :, ::997, :].copy()
    d = adapt(g, 0)
    del grids
    execute_timestep_batch([d], 1, 3, np.float32(0.0), params)
    f3 = d.download("f")[:, :, :, ::997, :]
    assert np.abs(f3 - f0).max() < 3e-7
    d.close()


def test_empty_level_and_single_block(gpu):
    lib = _lib.load()
    empty = BlockLevel(1, [], np.zeros((0, 27), np.int32, order="F"), 1.0, 1.0, 0.6)
    d = adapt(empty, 0)
    from open_ludwig_amd.physics import SolverParams
    p = SolverParams(domain_nx=8, domain_ny=8, domain_nz=8)
    execute_timestep_batch([d], 1, 2, np.float32(0.0), p)          # reference returns early on n_blocks == 0
    assert d.info().n_blocks == 0
    d.close()
    # one lonely block: every neighbour missing -> inlet / outlet / mirror chain on all six sides
    one = BlockLevel(1, [(1, 1, 1)], build_neighbor_table([(1, 1, 1)], 1, 1, 1), 1.0, 1.0, 0.6, enable_temporal_interpolation=False)
    cases.init_perturbed(one, 1)
    from oracle import oracle
    dev = adapt(one, 0)
    execute_timestep_batch([dev], 1, 3, np.float32(0.05), p)
    oracle.execute_timestep_batch([one], 1, 3, np.float32(0.05), p)
    assert np.array_equal(dev.download("f"), one.f) and np.array_equal(dev.download("rho"), one.rho)
    dev.close()


def test_abi_rejects_bad_arguments(gpu):
    lib = _lib.load()
    grids, params = cases.periodic_box((2, 2, 2))
    d = adapt(grids[0], 0)
    bad = np.zeros(10, np.float32)
    assert lib.ludwig_level_upload(d.handle, _lib.F, bad.ctypes.data, bad.nbytes) == -1
    assert b"expected" in lib.ludwig_last_error()
    assert lib.ludwig_level_download(d.handle, _lib.F_POST, bad.ctypes.data, bad.nbytes) == -5      # not allocated
    assert lib.ludwig_level_download(d.handle, _lib.F_OLD, bad.ctypes.data, bad.nbytes) == -5
    assert lib.ludwig_level_upload(d.handle, 99, bad.ctypes.data, bad.nbytes) == -5
    items = np.arange(8 * 8, dtype=np.int32)
    items[3] = items[2]                                                                                # duplicate
    assert lib.ludwig_level_set_order(d.handle, 0, items.ctypes.data, items.size) == -1
    short = np.arange(8, dtype=np.int32)
    assert lib.ludwig_level_set_order(d.handle, 0, short.ctypes.data, short.size) == -1
    fl = params.to_c()
    assert lib.ludwig_stream_collide(d.handle, None, -1, 0.0, 0.5, 0.0, C.byref(fl), 0) == -1         # negative t_sub
    assert lib.ludwig_stream_collide(d.handle, None, 1, 0.0, 0.5, 0.0, C.byref(fl), 7) == -1          # bad part
    # a valid call still works after the rejected ones
    execute_timestep_batch([d], 1, 1, np.float32(0.0), params)
    assert np.isfinite(d.download("rho")).all()
    d.close()
    with pytest.raises(RuntimeError):
        d.handle
