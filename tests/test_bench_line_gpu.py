"""The driver's contract for bench.py at N = 1: one JSON line with the metric BASELINE.json names, the roofline and
cpu_baseline objects, and figures that agree with each other (the driver's own flags: --steps K --warmup W)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_gpu_line(cuda):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--no-epoch", "--no-variants"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert base["metric"].startswith(d["metric"]) and d["unit"] == "edges/s"      # BASELINE.json: "... edges/sec + epoch time, ..."
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    e = d["config"]["edges_per_gpu"]
    assert abs(d["value"] - e / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (d["ms_per_step"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    n, c = d["config"]["nodes_per_gpu"], d["config"]["channels"]
    assert r["algorithmic_bytes"] == e * (4 * c + 8) + n * (8 * c + 8)
    assert 0.2 < r["frac"] < 0.7 and r["traffic"] is None or r["traffic"] > r["algorithmic_bytes"]
    # the dominant launch fits inside the step; its floors were measured in this run
    assert 0 < r["kernel_ms_batched_events"] < d["ms_per_step"] and r["kernel_ms"] >= r["kernel_ms_batched_events"] * 0.999
    assert 0 < r["gather_floor_ms"] < r["kernel_ms_batched_events"] * 1.05
    assert 0 < r["normalize_floor_ms"] < r["normalize_kernel_ms"] * 1.2
    assert r["step_over_floors"] > 1.0
    # where the split rows were finalized, and the other form beside it
    assert r["finalize_workgroups_in_main"] > 0 and r["finalize_kernel_ms"] < 0.002
    ab = r["finalize_as_a_launch"]
    assert ab["finalize_kernel_ms"] > 0.002 and ab["ms_per_step"] > 0
    # steady figure and cold figure
    assert d["preheat"]["steps"] > 0 and d["cold_start"]["ms_per_step"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
    assert d["value"] > 100 * cb["value"]
    assert d["max_abs_err_vs_oracle"] < 1e-5
