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
@pytest.mark.parametrize("grid,nb,steps,pad", [("2x1x1", 4, 7, None), ("2x2x2", 4, 8, None), ("1x2x2", "8,4,2", 6, None)])
def test_halo_exchange_over_rccl_loopback(gpu, tmp_path, grid, nb, steps, pad):
    """The production exchange (pack -> grouped isend/irecv on RCCL -> unpack on the comm stream, interior blocks stepping under
    it) with all 1 / 7 peers wired to rank 0: fields identical to the single-device run of the one-brick box."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = tmp_path / "rep.json"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_loopback_worker.py"), grid, str(nb), str(steps), str(out)],
                         capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    rep = json.load(open(out))
    assert rep["backend"] == "nccl" and rep["collectives_ok"]
    assert rep["peers"] == {"2x1x1": 1, "2x2x2": 7, "1x2x2": 3}[grid]      # 1x2x2 with 8 x 4 x 2 blocks: the shape of bench.py's 8-rank bricks
    assert rep["moved"] and all(rep["identical"].values()), rep
