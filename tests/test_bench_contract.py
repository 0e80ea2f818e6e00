"""bench.py's contract with the driver: one JSON line on stdout with the agreed keys; no GPU, no number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this container has a GPU")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert res.returncode != 0 and res.stdout.strip() == "" and "needs a GPU" in res.stderr


@pytest.mark.gpu
def test_bench_prints_exactly_one_json_line_with_the_contract_keys(gpu):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "64", "--steps", "4", "--warmup", "2", "--cpu-seconds", "1",
                          "--cpu-size", "32", "--preheat-ms", "5"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[:500]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "MLUPS" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "64^3" in d["config"]["workload"] and "model" not in d["config"] and d["config"]["state_finite"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None and "64" in r["traffic_source"] or r["traffic"] is None      # traffic.json is for 256^3 only
    assert abs(r["achieved"] - 216 * 64 ** 3 / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "MLUPS" and c["cores"] >= 1 and c["value"] > 0 and "32^3" in c["sample"]
    assert abs(d["value"] - 64 ** 3 * 4 / (d["ms_per_step"] * 4 * 1e-3) / 1e6) / d["value"] < 1e-2


def _run_bench(*argv, env=None, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                          env=dict(os.environ, **(env or {})))


def test_bench_starts_its_own_ranks_from_a_plain_call():
    """`python3 bench.py --gpus 2` with no WORLD_SIZE in the environment (the form the driver uses for N = 1) starts two ranks as a
    child torch.distributed.run. --plan-only stops after the rendezvous and the halo plan - no GPU needed - and rank 0's ONE JSON line
    carries the backend's own world size; both scaling modes."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for mode, size, box, brick in (("weak", "32", [32, 32, 64], [4, 4, 4]), ("strong", "64", [64, 64, 64], [8, 8, 4])):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", mode, "--size", size, "--plan-only"],
                             capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [l for l in res.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, res.stdout[:500]
        d = json.loads(lines[0])
        assert d["value"] is None and d["plan_only"] is True and d["n_gpus"] == 2 and d["scaling"] == mode
        assert d["comm"]["backend_world_size"] == 2 and d["comm"]["rank_grid"] == [1, 1, 2] and d["comm"]["blocks_per_rank"] == brick
        assert d["config"]["global_box_cells"] == box and mode.upper() in d["config"]["workload"]
        assert d["comm"]["halo_bytes_per_rank_per_step"][0] == d["comm"]["halo_bytes_per_rank_per_step"][1] > 0


def test_self_launched_ranks_fail_loudly_without_gpus():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this container has a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "32", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode != 0 and res.stdout.strip() == "" and "needs a GPU" in res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode,size", [("weak", "64"), ("strong", "64")])
def test_self_launched_two_rank_rehearsal_on_one_gpu(gpu, mode, size):
    """The whole N > 1 bench path from a plain `python3 bench.py --gpus 2`: two ranks on this box's one GPU (gloo + host-staged
    messages: RCCL refuses two ranks on a device), ONE JSON line, the backend reports world size 2, the state stays finite."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LUDWIG_BENCH_FORCE_DEVICE"] = "0"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", mode, "--size", size, "--steps", "4", "--warmup", "2",
                          "--preheat-ms", "5"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == mode and d["comm"]["backend_world_size"] == 2 and d["config"]["state_finite"]
    cells = 2 * 64 ** 3 if mode == "weak" else 64 ** 3
    assert d["config"]["global_cells"] == cells
    assert abs(d["value"] - cells * 4 / (d["ms_per_step"] * 4 * 1e-3) / 1e6) / d["value"] < 1e-2
