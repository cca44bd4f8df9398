"""CPU: bench.py's grid mode (the real round structure through the product's runner on a latents-only pipeline adapter)
with a stand-in for the HIP pipeline: single process and gloo world 2; plus the launch plumbing that needs no device."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


class FakeDevicePipeline:
    """What LatentGridPipeline needs of Diffuman4DPipeline: .device, upload_plan, window_call (adds 1 to the window's targets)."""
    device = torch.device("cpu")

    def __init__(self):
        self.copies_seen = []

    def upload_plan(self, plan, guidance_scale, shard=None, copies=1, rows_per_task=0):
        assert copies == 1 or rows_per_task > 0
        self.copies_seen.append(copies)
        win = [np.concatenate([w + k * rows_per_task for k in range(copies)]) for w in plan.windows]  # stacked tasks: rows k * n ...
        cond = [np.concatenate([c] * copies) for c in plan.is_cond]
        return dict(win=[torch.from_numpy(w) for w in win], cond=[torch.from_numpy(c) for c in cond],
                    calls=len(plan.windows), cfg=2 if guidance_scale > 1 else 1)

    def window_call(self, lat3, pv3, pl3, sk3, cm3, tb, i, h, w, domains, gs, use_cfg, vpred, shard=None):
        assert lat3.shape[1:] == (h * w, 4) and pv3.shape[0] == lat3.shape[0] and len(domains) == tb["cfg"]
        lat3[tb["win"][i][~tb["cond"][i]]] += 1.0


@pytest.fixture()
def small_latents(monkeypatch):
    monkeypatch.setattr(bench, "LAT_H", 2)
    monkeypatch.setattr(bench, "LAT_W", 2)


def test_grid_depth_keeps_the_call_mix():
    assert bench.grid_depth(20) == {"spatial": 4, "temporal": 14}
    assert bench.grid_depth(4) == {"spatial": 1, "temporal": 3}
    assert bench.grid_depth(1) == {"spatial": 1, "temporal": 3}
    assert bench.grid_depth(10 ** 4) == {"spatial": 22, "temporal": 75}  # never deeper than the real sweeps
    d = bench.grid_depth(20)
    assert abs((300 * d["spatial"]) / (44 * d["temporal"]) - 2.0) < 0.1  # 6600 : 3300 in the full run


def test_grid_pass_single_process(small_latents):
    calls = bench.run_grid_pass(FakeDevicePipeline(), {"spatial": 2, "temporal": 5}, 12, 1, 0, 2)
    assert calls == (12 + 12) * 2 + 44 * 5


def test_grid_pass_in_task_stacks(small_latents):
    """runner.task_batch through the adapter: the tasks of a round in stacks of at most three sharing their window calls; the count is in
    task-calls (a stacked call counts once per task), so it equals the unstacked pass's."""
    fake = FakeDevicePipeline()
    calls = bench.run_grid_pass(fake, {"spatial": 2, "temporal": 5}, 12, 1, 0, 2, task_batch=3)
    assert calls == (12 + 12) * 2 + 44 * 5
    assert sorted(set(fake.copies_seen)) == [2, 3] and fake.copies_seen.count(3) == 4 + 14 + 4 and fake.copies_seen.count(2) == 1


def test_deal_units():
    assert bench.deal_units(20, 2, 2) == [[2] * 5, [2] * 5]
    assert bench.deal_units(20, 2, 3) == [[3, 3, 2, 2], [3, 3, 2, 2]]
    assert bench.deal_units(20, 3, 1) == [[1] * 7, [1] * 7, [1] * 6]
    assert bench.deal_units(5, 2, 2) == [[2, 1], [2]]
    assert bench.deal_units(1, 3, 4) == [[1], [], []]
    assert bench.deal_units(0, 2, 2) == [[], []]
    for count in range(0, 40):
        for streams in (1, 2, 3):
            for batch in (1, 2, 3, 4):
                d = bench.deal_units(count, streams, batch)
                assert sum(map(sum, d)) == count and all(0 < z <= batch for st in d for z in st)


@pytest.mark.parametrize("streams,batch", [(1, 1), (3, 1), (2, 2), (2, 3), (1, 4)])
def test_unit_runner_runs_every_unit_once(small_latents, streams, batch):
    """bench.UnitRunner (the timed region of the task mode): K units dealt to `streams` worker threads in stacks of at most `batch` -- every
    unit's three window calls run exactly once whatever K, and `prepare` has built (and run once) every task state before the timing."""
    fake = FakeDevicePipeline()
    fake.dtype = torch.float32
    ur = bench.UnitRunner(fake, torch.device("cpu"), streams, batch)
    ur.prepare(20, 5)
    built = set(ur.sets)
    assert built == {(si, z) for c in (20, 5) for si, sizes in enumerate(bench.deal_units(c, streams, batch)) for z in sizes}
    total = lambda: sum(float(t[d]["lat"].float().sum()) for t in ur.sets.values() for d in ("spatial", "temporal"))  # noqa: E731
    base = total()
    ur.run(0, 5)
    ur.run(5, 20)
    assert set(ur.sets) == built  # nothing is built inside a timed region
    sizes = ur.run_single_stream(25, 7)
    assert sum(sizes) == 7 and max(sizes) <= batch
    # the fake adds 1 to every element of the 12 target rows of a window: 3 calls x 12 rows x (2 x 2 x 4) elements per unit
    assert total() - base == pytest.approx((5 + 20 + 7) * 3 * 12 * 16)


def _worker(rank, world, port, outdir, task_batch=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bench.LAT_H, bench.LAT_W = 2, 2
        fake = FakeDevicePipeline()
        calls = bench.run_grid_pass(fake, {"spatial": 1, "temporal": 3}, 12, world, rank, 2, task_batch=task_batch)
        assert task_batch == 1 or max(fake.copies_seen) == task_batch
        torch.save(calls, f"{outdir}/r{rank}.pt")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("task_batch", [1, 2])
def test_grid_pass_two_ranks_gloo(task_batch):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, 29400 + os.getpid() % 500 + task_batch, d, task_batch), nprocs=2, join=True)
        per_rank = [torch.load(f"{d}/r{r}.pt") for r in range(2)]
    # every window call of the job ran exactly once; an even split unless the runner measured different task rates on a
    # loaded test machine and re-dealt a round (DistributedSamplingRunner.balance)
    assert sum(per_rank) == 2 * (6 + 22 * 3 + 6) and min(per_rank) >= 6 + 6


def test_self_launch_command(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(bench.parse())
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_config5_switches_the_workload_and_nothing_else(monkeypatch):
    """--config5 = BASELINE.json configs[4]: 225 frames, stride 1 (36 steps per latent, one latent per 3-call unit), CPU baseline
    off; attention is the bf16 kernel (the fp8 kernel of rounds 2-3 never beat it and was removed in round 4); the default line
    keeps sliding_fast."""
    for name in ("N_FRAMES", "STRIDE", "STEPS_PER_LATENT", "LATENTS_PER_UNIT"):
        monkeypatch.setattr(bench, name, getattr(bench, name))  # restored after the test
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    bench.apply_workload_flags(a)
    assert (bench.N_FRAMES, bench.STRIDE, bench.STEPS_PER_LATENT, bench.LATENTS_PER_UNIT) == (150, 2, 18, 2.0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--attention", "fp8"])
    with pytest.raises(SystemExit):
        bench.parse()  # the flag is gone
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config5"])
    a = bench.parse()
    bench.apply_workload_flags(a)
    assert a.no_cpu_baseline
    assert (bench.N_FRAMES, bench.STRIDE, bench.STEPS_PER_LATENT, bench.LATENTS_PER_UNIT) == (225, 1, 36, 1.0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config5", "--gpus", "2"])
    with pytest.raises(SystemExit):
        bench.apply_workload_flags(bench.parse())
