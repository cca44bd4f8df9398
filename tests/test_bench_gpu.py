"""GPU: bench.py honours its output contract -- exactly one JSON line on stdout with the required fields, the
`roofline` and `cpu_baseline` objects, and sane values."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "1", "--cpu-frames", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "latents/s" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] - 2.0 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]  # 2 latents per step
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 1e6
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["config"]["finite_outputs"] is True


@pytest.mark.gpu
def test_bench_with_two_task_streams():
    """Default scheduling: the K steps are dealt to two tasks in flight (one HIP stream each); the roofline figures
    come from the one-task-at-a-time pass that follows the timed region."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["steps"] == 2 and d["config"]["task_streams"] == 2 and d["config"]["finite_outputs"] is True
    assert abs(d["value"] - 2.0 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    rf = d["roofline"]
    assert rf["launches"] == 2 * 48 and 0.05 < rf["frac"] < 1.0 and "one task in flight" in rf["measured_in"]
