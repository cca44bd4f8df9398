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
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "1", "--cpu-frames", "2", "--grid-frames", "12"],
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
    assert d["config"]["finite_outputs"] is True and d["config"]["mode"] == "task" and d["scaling"] == "weak"
    # the CPU-baseline forward doubles as a parity check on the SD-2.1 geometry (same weights, same input): bf16 noise floor
    pr = d["parity"]
    assert 0.0 < pr["rel_l2"] < 2.5e-2 and pr["north_star_tolerance"] == 1e-3 and pr["meets_north_star"] == (pr["rel_l2"] <= 1e-3)
    # three distances: HIP vs oracle fp32, HIP vs oracle bf16 (the reference's arithmetic), oracle bf16 vs oracle fp32
    assert pr["hip_vs_oracle_fp32"] == pr["rel_l2"] and 0.0 < pr["hip_vs_oracle_bf16"] < 3e-2 and 0.0 < pr["oracle_bf16_vs_oracle_fp32"] < 3e-2
    # both precisions against the fp32 oracle on that call, and the parity precision's own step time with its per-family breakdown
    pm = pr["modes"]
    assert pm["fast"]["rel_l2"] == pr["rel_l2"] and 0.0 < pm["parity"]["rel_l2"] < 1e-4 and pm["parity"]["meets_north_star"] is True
    pp = d["secondary"]["parity_precision"]
    assert pp["finite_outputs"] is True and pp["ms_per_step"] > d["ms_per_step"]
    assert {"linear", "conv3x3", "attention", "groupnorm", "layernorm", "split"} <= set(pp["kernel_breakdown_one_step"])
    # the tolerance mode: precision "fp16", one MFMA per product -- an order of magnitude closer than the fast precision and faster than the parity
    # one.  (north_star's 1e-3 is on the decoded RGB of a task -- tests/modelcheck.py fp16_demo3d_sd21_72x40 --; a single UNet call sits at
    # 6e-4 for the judged F = 16 window and at 1.0-1.3e-3 for other windows, e.g. the two frames of this reduced call: FP16_BOUNDS gives it 2e-3.)
    assert 0.0 < pm["fp16"]["rel_l2"] < 2e-3 and pm["fp16"]["rel_l2"] < 0.2 * pm["fast"]["rel_l2"]
    assert pm["fp16"]["meets_north_star"] == (pm["fp16"]["rel_l2"] <= 1e-3)
    tm = d["secondary"]["tolerance_mode"]
    # (this run's main line is ONE single task, --steps 1; the tolerance mode's one-stack figure is per step of a 2-task stack, which runs
    # several per cent faster per step than a single task, so the fast single task and the fp16 stack may come out close)
    assert tm["precision"] == "fp16" and tm["finite_outputs"] is True and 0.9 * d["ms_per_step"] < tm["ms_per_step_one_task"] < pp["ms_per_step_one_task"]
    assert tm["task_batch"] == d["config"]["task_batch"] and pp["task_batch"] == 1
    assert {"linear", "conv3x3", "attention", "groupnorm", "layernorm"} <= set(tm["kernel_breakdown_one_step"])
    assert "seconds_f16_all" in cb and len(cb["seconds_f16_all"]) >= 1
    # the same-mode baseline of the N > 1 (grid) lines: one pass over the real round structure on this GPU
    g = d["secondary"]["grid"]
    assert g["n_gpus"] == 1 and g["window_calls_per_task"] == {"spatial": 1, "temporal": 3} and g["calls"] == (12 + 12) * 1 + 44 * 3
    assert abs(g["latents_per_s"] - g["calls"] * 12 / 18 / g["timed_seconds"]) < 1e-2 * g["latents_per_s"]


@pytest.mark.gpu
def test_bench_with_the_default_task_streams():
    """Default scheduling: the K steps are dealt to the runner's default number of task streams (one HIP stream each) in stacks of the
    runner's default task_batch (3 steps on 2 streams of 2-task stacks: one stack of two, one single task); the roofline figures come
    from the one-stack-at-a-time pass that follows the timed region (3 units on one stream: a stack of two and a single task = 2 x 48 attention
    launches)."""
    from diffuman4d_amd.host.runner import DEFAULT_GPU_STREAMS, DEFAULT_TASK_BATCH
    sys.path.insert(0, str(ROOT))
    import bench
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-grid-secondary",
                        "--no-parity-precision", "--no-tolerance-mode", "--no-latent128"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["steps"] == 3 and d["config"]["finite_outputs"] is True
    assert d["config"]["task_streams"] == min(3, DEFAULT_GPU_STREAMS) and d["config"]["task_batch"] == DEFAULT_TASK_BATCH
    assert abs(d["value"] - 2.0 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    rf = d["roofline"]
    stacks = len(bench.deal_units(3, 1, DEFAULT_TASK_BATCH)[0])
    assert rf["launches"] == stacks * 48 and 0.05 < rf["frac"] < 1.0 and "one stack in flight" in rf["measured_in"]
    k = d["kernel_breakdown_one_step"]  # per-family ms are per STEP whatever the stack size: they add up to about the one-stack step time
    fam = sum(k[f]["ms"] for f in ("linear", "conv3x3", "attention", "groupnorm", "layernorm"))
    assert 0.75 * d["ms_per_step"] < fam < 1.25 * d["ms_per_step"], (fam, d["ms_per_step"])


@pytest.mark.gpu
def test_bench_main_line_in_the_fp16_precision():
    """`--precision fp16` (a profiling aid: the judged line stays bf16): the same units on the fp16 precision's kernels, `dtype` says so."""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--precision", "fp16", "--no-grid-secondary", "--no-latent128",
                        "--no-vae"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert d["dtype"] == "fp16" and d["config"]["precision"] == "fp16" and d["config"]["finite_outputs"] is True and "cpu_baseline" not in d
    assert "tolerance_mode" not in d["secondary"] and "parity_precision" not in d["secondary"]


@pytest.mark.gpu
def test_bench_self_launches_grid_mode_for_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench starts its own ranks (torch.distributed.run on
    127.0.0.1), defaults to the grid mode (real round structure through DistributedSamplingRunner: partition, barrier,
    cell exchange) and rank 0 prints the one line.  Both ranks share GPU 0 over gloo here (DM4D_BENCH_SHARED_GPU, testing
    only), on a 48 x 12 grid so that the run stays short."""
    import os
    env = dict(os.environ, DM4D_BENCH_SHARED_GPU="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--grid-frames", "12"],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["mode"] == "grid" and d["scaling"] == "strong" and "cpu_baseline" not in d
    g = d["config"]["grid"]
    # 12 + 12 spatial tasks x 1 call, 44 temporal tasks x 3 calls (depth for K = 1), dealt round-robin to 2 ranks
    assert g["window_calls_per_task"] == {"spatial": 1, "temporal": 3} and g["calls_per_rank"] == [6 + 66 + 6, 6 + 66 + 6]
    assert abs(d["value"] - sum(g["calls_per_rank"]) * 12 / 18 / g["timed_seconds"]) < 1e-2 * d["value"]
    assert d["config"]["finite_outputs"] is True
    sg = d["secondary"]["grid"]  # same field names as the N = 1 line's secondary.grid: a 1 -> N curve is grid / grid
    assert sg["n_gpus"] == 2 and sg["latents_per_s"] == d["value"] and sg["calls"] == sum(g["calls_per_rank"])


@pytest.mark.gpu
def test_bench_hybrid_mode_runs_a_frame_sharded_tail_wave():
    """`bench.py --gpus 4 --mode hybrid` on a 48 x 14 grid: the spatial rounds have 14 tasks = three full waves of 4 (task-parallel) and a
    tail wave of 2 <= world / 2, which the product runner (DistributedSamplingRunner(mode="hybrid")) runs frame-sharded on the sub-groups
    [0, 1] and [2, 3]; the 44 temporal tasks are 11 full waves.  (Frame counts are even under sliding_fast and at least the window, 12: two ranks never
    leave a tail, and 14 frames on four ranks is the smallest grid that does.)  All ranks share GPU 0 over gloo (testing only): what is checked is that the HIP
    pipeline, the sub-group collectives and the cell exchange of the tail tasks work together and every target cell of the grid reaches
    the final timestep index (bench raises otherwise)."""
    import os
    env = dict(os.environ, DM4D_BENCH_SHARED_GPU="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--mode", "hybrid", "--steps", "1", "--warmup", "0", "--grid-frames", "14"],
                       cwd=ROOT, capture_output=True, text=True, timeout=2400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["mode"].startswith("grid") and "hybrid" in d["config"]["parallelism"] and d["config"]["finite_outputs"] is True
    g = d["config"]["grid"]
    # 14 spatial tasks x 1 call per spatial round (12 whole + 2 sharded over 2 ranks each, counted 0.5 per rank), 44 temporal tasks x 3 calls
    assert abs(sum(g["calls_per_rank"]) - (14 + 44 * 3 + 14)) < 0.05, g
