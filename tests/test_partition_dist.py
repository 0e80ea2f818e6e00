"""N > 1 path over torch.distributed: 2 processes, gloo. CPU variant runs everywhere; the GPU variant puts both ranks
on the one MI355X of the test box and stages messages through the host (RCCL needs one device per rank; the driver
measures the real RCCL path at round end)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from open_ludwig_amd import cases
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(mode, outdir, nbg, steps, world=2, overlap=1):
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"), mode, str(outdir),
           str(nbg[0]), str(nbg[1]), str(nbg[2]), str(steps), str(overlap)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]


def _check(outdir, nbg, steps, world):
    grids, params = cases.periodic_box(nbg)
    oracle.execute_timestep_batch(grids, 1, steps, np.float32(0.0), params)
    fn, vn = oracle.newest_buffers(0, steps)
    seen = 0
    for r in range(world):
        d = np.load(os.path.join(outdir, f"rank{r}.npz"))
        for name in (fn, vn, "rho"):
            assert np.array_equal(d[name], getattr(grids[0], name)[:, :, :, d["l2g"]]), f"rank {r} {name}"
        seen += d["l2g"].size
    assert seen == grids[0].n_blocks


def test_two_ranks_gloo_cpu(tmp_path):
    nbg, steps = (4, 2, 2), 3
    _launch("cpu", tmp_path, nbg, steps, world=2)
    _check(tmp_path, nbg, steps, 2)


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_force_sums_gloo_cpu(tmp_path, world):
    """SURVEY 8e diagnostics over gloo: per-rank stresses of the rank's own triangles, nine partial Float32 sums per rank, one
    all-gather, added in rank order; rho_min by all-reduce MIN. Against the single-domain evaluation of the same fields:
    coefficients within 1e-6 of the largest one (another summation order), coverage and rho_min exact, same on every rank."""
    import json
    _launch("cpu_forces", tmp_path, (0, 0, 0), 0, world=world)
    res = [json.load(open(os.path.join(tmp_path, f"forces{r}.json"))) for r in range(world)]
    assert all(r["dist"] == res[0]["dist"] for r in res), "every rank must hold the same totals"
    assert all(r["n_mine"] > 100 for r in res), "every rank was meant to own part of the surface"
    d, s = res[0]["dist"], res[0]["single"]
    scale = max(abs(v) for v in s[:4])
    assert scale > 1e-3
    for a, b in zip(d[:4], s[:4]):
        assert abs(a - b) <= 1e-6 * scale, (d, s)
    for a, b in zip(d[4:6], s[4:6]):
        assert abs(a - b) <= 2e-6 * max(abs(s[4]), abs(s[5])), (d, s)
    assert d[6] == s[6] and d[7] == s[7]


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [1, 0])
def test_two_ranks_one_gpu_hip_path(tmp_path, gpu, overlap):
    nbg, steps = (8, 4, 4), 4
    _launch("gpu", tmp_path, nbg, steps, world=2, overlap=overlap)
    _check(tmp_path, nbg, steps, 2)


@pytest.mark.gpu
def test_four_ranks_one_gpu_hip_path(tmp_path, gpu):
    nbg, steps = (4, 4, 4), 3
    _launch("gpu", tmp_path, nbg, steps, world=4)
    _check(tmp_path, nbg, steps, 4)


@pytest.mark.gpu
def test_two_ranks_tunnel_with_bouzidi_across_the_cut(tmp_path, gpu):
    """Non-periodic tunnel (inlet/outlet/mirror edges, sponge, obstacle) cut through the sphere: Bouzidi cells on both
    ranks, f_post_collision exchanged between collision and correction. Owned blocks must equal the single-domain oracle."""
    nbg, steps = (8, 4, 4), 3
    _launch("gpu_tunnel", tmp_path, nbg, steps, world=2)
    grids, params = cases.tunnel_with_sphere(nbg, levels=1, wall_model=False, temporal=False)
    oracle.execute_timestep_batch(grids, 1, steps, np.float32(0.05), params)
    fn, vn = oracle.newest_buffers(0, steps)
    both_have_cells, used_post_halo = [], []
    for r in range(2):
        d = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        for name in (fn, vn, "rho"):
            assert np.array_equal(d[name], getattr(grids[0], name)[:, :, :, d["l2g"]]), f"rank {r} {name}"
        both_have_cells.append(int(d["nbc"][0]) > 0)
        used_post_halo.append(bool(d["nbc"][1]))
    assert all(both_have_cells), "the cut was meant to pass through the Bouzidi cells"
    assert any(used_post_halo), "no f_post_collision element crossed the cut: the test does not exercise the exchange"


def _check_multilevel(outdir, nbg, steps, levels, wall, world, rtol=0.0):
    grids, params = cases.tunnel_with_sphere(nbg, levels=levels, wall_model=wall, temporal=True)
    oracle.execute_timestep_batch(grids, 1, steps, np.float32(0.05), params)
    cut_levels = 0
    for i, g in enumerate(grids):
        fn, vn = oracle.newest_buffers(i, steps)
        seen, ghosts = 0, 0
        for r in range(world):
            d = np.load(os.path.join(outdir, f"rank{r}.npz"))
            l2g = d[f"l2g{i}"]
            for got, name in ((d[f"f{i}"], fn), (d[f"vel{i}"], vn), (d[f"rho{i}"], "rho")):
                want = getattr(g, name)[:, :, :, l2g]
                if rtol == 0.0:
                    assert np.array_equal(got, want), f"level {i + 1} rank {r} {name}"
                else:
                    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * 1e-2, err_msg=f"level {i + 1} rank {r} {name}")
            seen += l2g.size
            ghosts += int(d[f"stats{i}"][1] - d[f"stats{i}"][0])
        assert seen == g.n_blocks
        cut_levels += ghosts > 0
    return cut_levels


@pytest.mark.gpu
@pytest.mark.parametrize("levels", [2, 3])
def test_two_ranks_nested_levels_cut_through_the_refinement(tmp_path, gpu, levels):
    """Scope row N3: nested levels on 2 ranks, the cut passing through every level. The fine levels interpolate from
    parent cells that live in parent GHOST blocks (f_k, rho, u and their saved old copies), Bouzidi cells sit on both sides.
    Owned blocks of every level must equal the single-domain oracle bit for bit."""
    nbg, steps = (8, 4, 4), 3
    _launch("gpu_multilevel", tmp_path, nbg, steps, world=2, overlap=levels)
    assert _check_multilevel(tmp_path, nbg, steps, levels, False, 2) == levels, "every level was meant to be cut"


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4])
def test_nested_levels_every_level_cut_on_its_own(tmp_path, gpu, world):
    """partition.level_owners: each of the 3 levels is bisected separately into `world` equal parts, so the cuts do not
    line up and fine blocks interpolate from parent blocks that live on other ranks (kept as extra ghost copies; on some
    ranks a level consists of such copies only). Owned blocks of every level must still equal the single-domain oracle
    bit for bit, and every rank must own its share of every level."""
    nbg, steps, levels = (8, 4, 4), 3, 3
    _launch("gpu_multilevel", tmp_path, nbg, steps, world=world, overlap=levels | 16)
    _check_multilevel(tmp_path, nbg, steps, levels, False, world)
    grids, _ = cases.tunnel_with_sphere(nbg, levels=levels, wall_model=False, temporal=True)
    for i, g in enumerate(grids):
        owned = [int(np.load(os.path.join(tmp_path, f"rank{r}.npz"))[f"l2g{i}"].size) for r in range(world)]
        assert sum(owned) == g.n_blocks and min(owned) >= 0.6 * g.n_blocks / world, (i + 1, owned)


@pytest.mark.gpu
def test_nested_levels_with_levels_living_on_single_ranks(tmp_path, gpu):
    """Level 1 entirely on rank 0, level 3 entirely on rank 1, level 2 bisected: rank 1 holds level 1 only as ghost copies
    (parent data for its level-2 blocks; n_owned < 0 in the ABI: nothing to step, halo unpack only), rank 0 has no copy of
    level 3 at all. Still bit-identical to the single-domain oracle."""
    nbg, steps, levels = (8, 4, 4), 3, 3
    _launch("gpu_multilevel", tmp_path, nbg, steps, world=2, overlap=levels | 32)
    _check_multilevel(tmp_path, nbg, steps, levels, False, 2)
    d0, d1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    assert d1["l2g0"].size == 0 and d0["l2g2"].size == 0 and d0["l2g1"].size > 0 and d1["l2g1"].size > 0
    assert d1["stats0"][1] > 0, "rank 1 keeps ghost copies of level 1"


@pytest.mark.gpu
def test_four_ranks_nested_levels_with_wall_model(tmp_path, gpu):
    nbg, steps = (8, 4, 4), 2
    _launch("gpu_multilevel", tmp_path, nbg, steps, world=4, overlap=2 | 8)
    _check_multilevel(tmp_path, nbg, steps, 2, True, 4)          # wall model: shared jl_math.h, identical too
