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
