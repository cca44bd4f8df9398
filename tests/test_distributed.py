"""CPU, gloo, world_size 2: the one-process-per-GPU runner (task sharding + grid exchange at round
boundaries) reproduces the single-process result cell for cell."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffuman4d_amd.host.dataset import SyntheticSpaTemDataset
from diffuman4d_amd.host.runner import DistributedSamplingRunner
from diffuman4d_amd.host.sampler import SlidingIterativeSampler
from stubs import ShardStubPipeline, StubPipeline

KW = dict(spa_label_range=[0, 20, 1], tem_label_range=[0, 12, 1], input_spa_labels=[1, 9], window_size=6,
          sliding_stride=2, alternation_rounds=3, bidirectional=False)


# more ranks than frames: a rank idles through both spatial rounds and still has to take part in the exchanges
KW_FEW_FRAMES = dict(spa_label_range=[0, 6, 1], tem_label_range=[0, 2, 1], input_spa_labels=[1], window_size=2,
                     sliding_stride=1, alternation_rounds=3, bidirectional=False)


def make_sampler(kw=None, stub=StubPipeline):
    ds = SyntheticSpaTemDataset(height=16, width=16, num_cameras=48)
    s = SlidingIterativeSampler(ds, [stub()], "/tmp/unused", **(kw or KW))
    s.result_writer = None  # (a None ctor argument selects the default JPEG writer)
    return s


def grid_state(s, cells=None):
    out = {}
    for c in s.spa_labels:
        for f in s.tem_labels:
            if cells is None or (c, f) in cells:
                l = s.latents[c][f]
                out[(c, f)] = (s.timestep_indices[c][f], None if l is None else l.clone())
    return out


def _worker(rank, world, port, outdir, kw=None, slow_rank=None, slow_delay=0.18, mode="task"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = make_sampler(kw, ShardStubPipeline if mode != "task" else StubPipeline)
        if slow_rank is not None:  # one GPU three times slower than the others
            import time
            pipe, delay = s.pipelines[0], (slow_delay if rank == slow_rank else 0.06)
            inner = pipe.sliding_iterative_denoise

            def slowed(**kwargs):
                time.sleep(delay)
                return inner(**kwargs)
            pipe.sliding_iterative_denoise = slowed
        runner = DistributedSamplingRunner(s, gpu_streams=1, prefetch_depth=1, mode=mode)
        runner.inference()
        last = len(s.all_tasks) - 1
        owned = set(runner._owned_after(last, rank))
        torch.save({"state": grid_state(s, owned), "n_calls": len(s.pipelines[0].calls),
                    "sharded": [(c["domain"], c["sharded"], c["noise_seed"]) for c in s.pipelines[0].calls if c.get("sharded")],
                    "tails": [[(t["domain_label"], ranks) for t, ranks in runner.tail_of(ri)] for ri in range(len(s.all_tasks))],
                    "deal": [[len(runner.tasks_of(ri, q)) for q in range(world)] for ri in range(len(s.all_tasks))]},
                   f"{outdir}/rank{rank}.pt")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,kw,steps", [(2, KW, 9), (3, KW_FEW_FRAMES, 6)], ids=["world2", "world3_more_ranks_than_frames"])
def test_ranks_match_single_process(world, kw, steps):
    ref = make_sampler(kw)
    for tasks in ref.all_tasks:
        for t in tasks:
            ref.execute_one_task(t)
    ref_state = grid_state(ref)
    total_calls = sum(len(t) for t in ref.all_tasks)
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + world
        mp.spawn(_worker, args=(world, port, d, kw), nprocs=world, join=True)
        merged, calls = {}, 0
        for r in range(world):
            blob = torch.load(f"{d}/rank{r}.pt")
            calls += blob["n_calls"]
            for k, v in blob["state"].items():
                assert k not in merged, f"cell {k} owned by two ranks"
                merged[k] = v
    assert calls == total_calls  # every task ran exactly once
    target_cells = {(c, f) for c in ref.target_spa_labels for f in ref.tem_labels}
    assert target_cells == set(merged)
    for cell in target_cells:
        idx, lat = merged[cell]
        ridx, rlat = ref_state[cell]
        assert idx == ridx == steps
        assert torch.equal(lat, rlat)


@pytest.mark.timeout(300)
def test_slow_rank_gets_fewer_tasks_and_the_grid_is_unchanged():
    """The reference's pipelines drain one queue, so a slow GPU takes fewer tasks (sampling_runner.py:27-33); here the rates
    measured in a round re-deal the next one.  Same grid as the single-process run, cell for cell."""
    ref = make_sampler(KW)
    for tasks in ref.all_tasks:
        for t in tasks:
            ref.execute_one_task(t)
    ref_state = grid_state(ref)
    world = 2
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + 7
        mp.spawn(_worker, args=(world, port, d, KW, 1), nprocs=world, join=True)
        blobs = [torch.load(f"{d}/rank{r}.pt") for r in range(world)]
    assert blobs[0]["deal"] == blobs[1]["deal"]  # replicated bookkeeping: both ranks computed the same deals
    deal = blobs[0]["deal"]
    assert deal[0][0] == deal[0][1]            # first round: nothing measured yet, round-robin
    assert deal[1][0] > deal[1][1] and deal[2][0] > deal[2][1], deal  # the slow rank 1 gets fewer tasks afterwards
    assert sum(b["n_calls"] for b in blobs) == sum(len(t) for t in ref.all_tasks)
    merged = {}
    for b in blobs:
        merged.update(b["state"])
    for cell in {(c, f) for c in ref.target_spa_labels for f in ref.tem_labels}:
        assert merged[cell][0] == ref_state[cell][0] and torch.equal(merged[cell][1], ref_state[cell][1])


@pytest.mark.timeout(300)
def test_very_slow_rank_may_get_no_task_and_still_exchanges():
    """World 3, one rank an order of magnitude slower: its share of the later rounds shrinks to (almost) nothing; ranks without
    a task still take part in the exchanges and the grid equals the single-process run."""
    ref = make_sampler(KW)
    for tasks in ref.all_tasks:
        for t in tasks:
            ref.execute_one_task(t)
    ref_state = grid_state(ref)
    world = 3
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + 11
        mp.spawn(_worker, args=(world, port, d, KW, 2, 0.6), nprocs=world, join=True)
        blobs = [torch.load(f"{d}/rank{r}.pt") for r in range(world)]
    deal = blobs[0]["deal"]
    assert all(b["deal"] == deal for b in blobs)
    assert deal[1][2] < deal[1][0] and deal[2][2] < deal[2][0] and deal[2][2] <= 2, deal
    assert sum(b["n_calls"] for b in blobs) == sum(len(t) for t in ref.all_tasks)
    merged = {}
    for b in blobs:
        merged.update(b["state"])
    for cell in {(c, f) for c in ref.target_spa_labels for f in ref.tem_labels}:
        assert merged[cell][0] == ref_state[cell][0] and torch.equal(merged[cell][1], ref_state[cell][1])


def test_weighted_deal():
    deal = DistributedSamplingRunner.weighted_deal
    assert deal(7, [1.0, 1.0, 1.0]) == [[0, 3, 6], [1, 4], [2, 5]]  # equal rates: the round-robin deal [r::world]
    d = deal(12, [1.0, 0.5])
    assert sorted(d[0] + d[1]) == list(range(12)) and len(d[0]) == 8 and len(d[1]) == 4
    assert deal(0, [1.0, 2.0]) == [[], []]


def test_partition_is_a_disjoint_cover():
    s = make_sampler()
    for ri, tasks in enumerate(s.all_tasks):
        parts = [s.partition(ri, r, 3) for r in range(3)]
        flat = [t["domain_label"] for p in parts for t in p]
        assert sorted(flat) == sorted(t["domain_label"] for t in tasks)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


# hybrid: 8 cameras (6 targets) x 8 frames on 4 ranks -- the spatial rounds are two full waves (8 tasks), the temporal round is one
# full wave + 2 tasks, which run frame-sharded on 2 sub-groups of 2 ranks (the 44-camera round on 8 GPUs in small)
KW_HYBRID = dict(spa_label_range=[0, 8, 1], tem_label_range=[0, 8, 1], input_spa_labels=[1, 5], window_size=4, sliding_stride=2,
                 alternation_rounds=3, bidirectional=False)
# fewer tasks than ranks in every round: 2 frames / 6 target cameras on 4 ranks
KW_HYBRID_FEW = dict(spa_label_range=[0, 8, 1], tem_label_range=[0, 2, 1], input_spa_labels=[1, 5], window_size=2, sliding_stride=1,
                     alternation_rounds=3, bidirectional=False)


def _run_and_merge(world, kw, mode, port_off):
    ref = make_sampler(kw)
    for tasks in ref.all_tasks:
        for t in tasks:
            ref.execute_one_task(t)
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + (os.getpid() % 2000) + port_off
        mp.spawn(_worker, args=(world, port, d, kw, None, 0.18, mode), nprocs=world, join=True)
        blobs = [torch.load(f"{d}/rank{r}.pt") for r in range(world)]
    merged = {}
    for b in blobs:
        for k, v in b["state"].items():
            assert k not in merged, f"cell {k} owned by two ranks"
            merged[k] = v
    ref_state = grid_state(ref)
    cells = {(c, f) for c in ref.target_spa_labels for f in ref.tem_labels}
    assert cells == set(merged)
    for cell in cells:
        assert merged[cell][0] == ref_state[cell][0] and torch.equal(merged[cell][1], ref_state[cell][1]), cell
    return ref, blobs


@pytest.mark.timeout(300)
def test_hybrid_mode_shards_the_tail_wave_and_matches_single_process():
    """runner.mode = hybrid on 4 ranks: full waves task-parallel, the 2-task tail of the temporal round frame-sharded on 2 sub-groups of
    2 ranks (real gloo all-gathers inside the stub) -- same grid as one process, cell for cell; every group member was handed the
    same noise seed; the replicated bookkeeping agrees on who ran what."""
    ref, blobs = _run_and_merge(4, KW_HYBRID, "hybrid", 21)
    tails = blobs[0]["tails"]
    assert all(b["tails"] == tails for b in blobs)
    assert tails[0] == [] and tails[2] == []                    # 8 spatial tasks = two full waves
    assert [ranks for _, ranks in tails[1]] == [[0, 1], [2, 3]]   # 6 temporal tasks = one wave + 2 tasks on 2 groups of 2
    n_tasks = sum(len(t) for t in ref.all_tasks)
    assert sum(b["n_calls"] for b in blobs) == n_tasks + 2        # the 2 tail tasks ran on 2 ranks each
    for grp in ([0, 1], [2, 3]):
        seeds = [blobs[r]["sharded"] for r in grp]
        assert seeds[0] == seeds[1] and len(seeds[0]) == 1 and seeds[0][0][:2] == ("temporal", 2)
    assert blobs[0]["sharded"] != blobs[2]["sharded"]            # different tasks, different seeds


@pytest.mark.timeout(300)
def test_hybrid_mode_with_fewer_tasks_than_ranks():
    """2 frames on 4 ranks: the spatial rounds have 2 tasks (both sharded on 2 ranks each instead of leaving 2 ranks idle), the temporal
    round 6 = one wave + 2 sharded."""
    _, blobs = _run_and_merge(4, KW_HYBRID_FEW, "hybrid", 23)
    tails = blobs[0]["tails"]
    assert [len(t) for t in tails] == [2, 2, 2] and all(len(ranks) == 2 for rnd in tails for _, ranks in rnd)


@pytest.mark.timeout(300)
def test_frame_shard_mode_runs_every_task_on_all_ranks():
    """runner.mode = frame-shard on 2 ranks: every task on both ranks, the grid replicated, nothing to exchange."""
    ref, blobs = _run_and_merge(2, KW_HYBRID, "frame-shard", 25)
    n_tasks = sum(len(t) for t in ref.all_tasks)
    assert [b["n_calls"] for b in blobs] == [n_tasks, n_tasks]
    assert all(len(b["sharded"]) == n_tasks for b in blobs) and blobs[0]["sharded"] == blobs[1]["sharded"]


def test_hybrid_split_of_the_judged_grid():
    """The 48 x 150 job on 8 ranks (BASELINE.json configs[2..3]): spatial rounds 150 = 18 waves + 6 (> world / 2: stays task-parallel),
    temporal round 44 = 5 waves + 4 -> 4 sub-groups of 2 ranks (24-frame windows split 12 + 12).  Pure bookkeeping, no process group."""
    kw = dict(spa_label_range=[0, 48, 1], tem_label_range=[0, 150, 1], input_spa_labels=[1, 13, 25, 37], window_size=12,
              sliding_stride=2, alternation_rounds=3, bidirectional=False)
    s = make_sampler(kw)
    r = DistributedSamplingRunner.__new__(DistributedSamplingRunner)
    r.sampler, r.world, r.mode, r._assign = s, 8, "hybrid", {}
    main, tail, width = r._split_round(0)
    assert (len(main), len(tail), width) == (150, 0, 1)
    main, tail, width = r._split_round(1)
    assert (len(main), len(tail), width) == (40, 4, 2)
    assert [ranks for _, ranks in r.tail_of(1)] == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert sorted(t["domain_label"] for q in range(8) for t in r.tasks_of(1, q)) == sorted(t["domain_label"] for t in main)
    r.mode = "frame-shard"
    assert r._split_round(1)[2] == 8 and len(r._split_round(1)[1]) == 44
    with pytest.raises(ValueError, match="Unsupported runner mode"):
        DistributedSamplingRunner(s, mode="bogus")


def test_frame_sharding_modes_refuse_the_parity_precision():
    """precision 'parity' has no frame-sharded attention: the runner says so when it is built, not inside the first tail task."""
    kw = dict(spa_label_range=[0, 8, 1], tem_label_range=[0, 4, 1], input_spa_labels=[1, 5], window_size=4, sliding_stride=2,
              alternation_rounds=3, bidirectional=False)
    s = make_sampler(kw)
    s.pipelines[0].parity = True
    for mode in ("hybrid", "frame-shard"):
        with pytest.raises(ValueError, match="precision 'parity' does not support"):
            DistributedSamplingRunner(s, mode=mode)

