"""RCCL itself on a one-GPU box: rank 0 of a brick decomposition whose bricks hold identical data talks to itself over the
`nccl` backend (tests/_rccl_loopback_worker.py). The CPU test checks the construction the GPU test rests on - that reading the
peers' messages from rank 0's own brick reproduces the one-brick periodic run - with the oracle and plain array copies."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from open_ludwig_amd import cases
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 2)])
def test_symmetric_brick_plan_reproduces_the_one_brick_box_cpu(grid):
    import _rccl_loopback_worker as w
    nb, steps = 2, 4
    view, plan, params = w.symmetric_brick_plan(grid, nb, upload_only=False)
    assert len(plan.peers) == int(np.prod(grid)) - 1 and 0 not in plan.peers
    lvl = view.level
    for p in plan.peers:                    # one message per peer and field, equal length both ways
        for name in ("f", "vel"):
            assert plan.send[p][name].size == plan.recv[p][name].size
        assert plan.send[p]["f"].size > 0          # edge and corner peers: populations only, the velocity stencil has faces only

    def exchange(fields):
        for group, name in fields.items():
            a = getattr(lvl, name)
            flat = a.reshape(-1, order="F").copy()
            for p in plan.peers:
                flat[plan.recv[p][group]] = flat[plan.send[p][group]]
            a[...] = flat.reshape(a.shape, order="F")

    exchange({"f": "f", "vel": "vel"})
    exchange({"f": "f_temp", "vel": "vel_temp"})
    for t in range(1, steps + 1):
        oracle.execute_timestep_batch([lvl], t, 1, np.float32(0.0), params)
        exchange({"f": "f_temp", "vel": "vel_temp"} if t % 2 == 0 else {"f": "f", "vel": "vel"})
    grids, params1 = cases.periodic_box((nb, nb, nb))
    oracle.execute_timestep_batch(grids, 1, steps, np.float32(0.0), params1)
    fn, vn = oracle.newest_buffers(0, steps)
    pos = {tuple(c): i for i, c in enumerate(grids[0].active_block_coords)}
    sel = np.array([pos[tuple(c)] for c in np.asarray(lvl.active_block_coords)[: view.n_owned]])
    for n in (fn, vn, "rho"):
        assert np.array_equal(getattr(lvl, n)[:, :, :, : view.n_owned], getattr(grids[0], n)[:, :, :, sel]), n
    assert getattr(lvl, vn)[:, :, :, : view.n_owned].std() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("grid,nb,steps,transport", [("2x1x1", 4, 7, "native"), ("2x2x2", 4, 8, "native"), ("1x2x2", "8,4,2", 6, "native"),
                                                      ("2x2x2", 4, 8, "torch"), ("2x2x2", 4, 7, "native+spheres"), ("2x1x1", 4, 6, "torch+spheres")])
def test_halo_exchange_over_rccl_loopback(gpu, tmp_path, grid, nb, steps, transport):
    """The production exchange with all 1 / 7 peers wired to rank 0: fields identical to the single-device run of the one-brick box.
    native: ludwig_step_distributed - octet pack, one ncclSend / ncclRecv per peer and field in one group called FROM THE LIBRARY on
    its own high-priority stream, unpack, interior blocks stepping under it (messages to rank 0 itself forced through RCCL).
    torch: the same schedule driven from Python over torch.distributed's nccl backend (round 2's path, kept for the rehearsals).
    +spheres: a body on every brick corner, so Bouzidi cells sit on both sides of every cut and their q < 1/2 links read
    f_post_collision across it: the f_post halo between collision and correction (group 2 of the plan) travels too."""
    spheres = transport.endswith("+spheres")
    transport = transport.split("+")[0]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", LOOPBACK_TRANSPORT=transport)
    out = tmp_path / "rep.json"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_loopback_worker.py"), grid, str(nb), str(steps), str(out)] + (["spheres"] if spheres else []),
                         capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    rep = json.load(open(out))
    assert rep["backend"] == "nccl" and rep["collectives_ok"] and rep["transport"] == transport
    assert rep["peers"] == {"2x1x1": 1, "2x2x2": 7, "1x2x2": 3}[grid]      # 1x2x2 with 8 x 4 x 2 blocks: the shape of bench.py's 8-rank bricks
    assert rep["moved"] and all(rep["identical"].values()), rep
    if spheres:
        assert rep["bouzidi_cells"] > 100 and rep["f_post_halo_elements"] > 0, rep


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [True, False])
def test_native_exchange_with_device_copies(gpu, overlap):
    """The library's halo plan without any communicator: one process, every peer of the symmetric brick plan is this rank itself, so
    ludwig_halo_exchange moves the messages with device copies - octet pack / unpack, the plan's stream and events and
    ludwig_step_distributed's schedule (interior, wait, boundary, exchange left in flight) against the single-device run."""
    import torch
    import _rccl_loopback_worker as w
    from open_ludwig_amd import adapt, partition
    from open_ludwig_amd.physics import stream_collide
    nb, steps, grid = 4, 7, (2, 2, 2)
    view, plan, params = w.symmetric_brick_plan(grid, nb)
    runner = partition.DistributedLevelRunner(view, plan, params, 0, overlap=overlap, transport="native", wire_rank={p: 0 for p in plan.peers})
    assert runner.comm is None
    runner.exchange_now("f", "vel")
    runner.exchange_now("f_temp", "vel_temp")
    runner.ex.timing = True
    for t in range(1, steps + 1):
        runner.step(t)
    runner.synchronize()
    assert len(runner.ex.exchange_ms()) == steps
    fn, vn = ("f_temp", "vel_temp") if steps % 2 == 0 else ("f", "vel")
    got = {n: runner.level.download(n)[:, :, :, : view.n_owned] for n in (fn, vn, "rho")}
    grids, params1 = cases.periodic_box((nb, nb, nb), upload_only=True)
    pos = {tuple(c): i for i, c in enumerate(grids[0].active_block_coords)}
    sel = np.array([pos[tuple(c)] for c in np.asarray(view.level.active_block_coords)[: view.n_owned]])
    single = adapt(grids[0], 0)
    for t in range(1, steps + 1):
        stream_collide(single, None, np.float32(0.5), np.float32(0.0), params1, t)
    torch.cuda.synchronize()
    for n in (fn, vn, "rho"):
        assert np.array_equal(got[n], single.download(n)[:, :, :, sel]), n
    assert got[vn].std() > 0
    single.close()
    runner.close()


@pytest.mark.gpu
@pytest.mark.parametrize("grid,steps,transport", [("2x1x1", 4, "native"), ("4x1x1", 3, "native"), ("2x1x1", 4, "native-overlap"), ("2x1x1", 4, "torch")])
def test_nested_levels_over_rccl_loopback(gpu, tmp_path, grid, steps, transport):
    """The multi-GPU schedule of NESTED levels (partition.MultiLevelRunner: per-level exchanges posted and joined around the part
    launches, both levels' same-level ghosts, temporal blend) over RCCL on one GPU: two levels on a periodic
    box of identical bricks in a row, the refined region a slab around every cut plane so that every cut runs through level 2
    (_rccl_loopback_worker.nested_symmetric_bricks); rank 0 plays its brick with every peer wired to itself. Both levels' f / u / rho of
    the owned blocks equal the single-device run of the one-brick box, bit for bit. native: every exchange is ludwig_halo_exchange
    (ncclSend / ncclRecv from the library); torch: batch_isend_irecv from Python."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", LOOPBACK_TRANSPORT=transport)
    # native: levels this small exchange ON their own stream (ludwig_halo_plan_in_stream; partition.MultiLevelRunner's rule);
    # native-overlap: the rule switched off - every exchange on its plan's stream under the level's interior part
    if transport == "native-overlap":
        transport = env["LOOPBACK_TRANSPORT"] = "native"
        env["LUDWIG_HALO_IN_STREAM_BELOW"] = "0"
    out = tmp_path / "rep.json"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_loopback_worker.py"), grid, "4", str(steps), str(out), "nested"],
                         capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    rep = json.load(open(out))
    assert rep["backend"] == "nccl" and rep["transport"] == transport and rep["nested"]
    if transport == "native":
        assert rep["exchange_in_stream"] == [env.get("LUDWIG_HALO_IN_STREAM_BELOW") != "0"] * 2
    assert rep["owned"] == [64, 256] and all(b > 0 for b in rep["halo_bytes_per_level"])
    # what this construction cannot reach: parent-data ghosts. An interface stencil crosses a cut only if the refined region ends AT the
    # cut, and there the reference's edge chain (global coordinates, no wrap) gives the one-brick box an inlet where the G-brick box
    # interpolates - the bricks stop being identical. Parent data across ranks stays verified over gloo (tests/test_partition_dist.py).
    assert not rep["parent_data_in_level1_halo"]
    assert rep["moved1"] and rep["moved2"] and all(rep["identical"].values()), rep
