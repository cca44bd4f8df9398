"""CPU: the 3-stage task pipeline of the runner (load || denoise || save) leaves the grid, the
timestep bookkeeping, the pipeline call sequence and the set of written samples exactly as the reference's serial
load -> denoise -> save order does, and surfaces worker errors."""
import threading

import pytest

from diffuman4d_amd.host.dataset import SyntheticSpaTemDataset
from diffuman4d_amd.host.runner import SamplingRunner, run_round_pipelined
from diffuman4d_amd.host.sampler import SlidingIterativeSampler
from stubs import StackStubPipeline, StubPipeline


def make(writer, **kw):
    ds = SyntheticSpaTemDataset(height=16, width=16, num_cameras=48)
    args = dict(window_size=6, sliding_stride=2, bidirectional=False, alternation_rounds=3, spa_label_range=[0, 20, 1],
                tem_label_range=[0, 12, 1], input_spa_labels=[1, 9], result_writer=writer)
    args.update(kw)
    return SlidingIterativeSampler(ds, [StubPipeline()], "/tmp/unused", **args)


def run(depth, writers=1, gpu_streams=1, task_batch=1, pipe=None):
    written, threads = [], set()

    def writer(sample, output_dir=None):
        threads.add(threading.current_thread().name)
        written.append((sample["alt"], sample["domain"], sample["domain_label"], sample["timestep_indices"].tolist()))

    s = make(writer)
    if pipe is not None:
        s.pipelines[0] = pipe
    for tasks in s.all_tasks:
        run_round_pipelined(s, tasks, 0, depth, writers, gpu_streams, task_batch=task_batch)
    grid = {c: {f: (s.timestep_indices[c][f], float(s.latents[c][f].flatten()[0])) for f in s.tem_labels} for c in s.spa_labels}
    return grid, s.pipelines[0].calls, written, threads


@pytest.mark.parametrize("depth,writers", [(1, 1), (3, 1), (4, 3)])
def test_pipelined_rounds_equal_serial_rounds(depth, writers):
    g0, calls0, w0, t0 = run(0)
    g1, calls1, w1, t1 = run(depth, writers)
    assert g1 == g0
    assert calls1 == calls0          # same tasks denoised in the same order, same latents-is-None pattern
    if writers == 1:
        assert w1 == w0              # every sample written, in task order, with the same final indices
    else:
        assert sorted(w1) == sorted(w0)
    assert t0 == {"MainThread"} and all(n.startswith("dm4d-writer") for n in t1)


@pytest.mark.parametrize("streams,depth", [(2, 1), (3, 2), (2, 0)])
def test_concurrent_gpu_streams_leave_the_same_grid(streams, depth):
    """Tasks of a round touch disjoint target cells: denoising several of them concurrently (one worker thread and
    HIP stream each) must leave the grid, the set of pipeline calls and the written samples of the serial order."""
    g0, calls0, w0, _ = run(0)
    g1, calls1, w1, _ = run(depth, 1, streams)
    assert g1 == g0
    assert sorted(map(repr, calls1)) == sorted(map(repr, calls0))  # same calls; their start order is not defined
    assert w1 == w0  # results are still collected and written in task order


@pytest.mark.parametrize("batch,streams,depth", [(3, 1, 0), (3, 2, 1), (2, 3, 2), (5, 1, 1), (64, 2, 0)])
def test_task_stacks_leave_the_same_grid(batch, streams, depth):
    """runner.task_batch: consecutive tasks of a round handed to the pipeline as one stack (shared window calls) -- same grid, same
    per-task pipeline calls, every sample written in task order; stacks are as even as the round allows (12 spatial tasks by 5 -> 4 + 4 + 4,
    18 temporal tasks by 5 -> 5 + 5 + 4 + 4) and never cross a round."""
    g0, calls0, w0, _ = run(0)
    pipe = StackStubPipeline()
    g1, calls1, w1, _ = run(depth, 1, streams, batch, pipe)
    assert g1 == g0 and w1 == w0
    assert sorted(map(repr, calls1)) == sorted(map(repr, calls0))
    per_round = [12, 18, 12]  # frames, target cameras, frames
    want = []
    for n in per_round:
        k = -(-n // batch)
        want += sorted([n // k + (1 if i < n % k else 0) for i in range(k)])
    got = pipe.stacks + [1] * (sum(per_round) - sum(pipe.stacks))  # a stack of one goes through the plain call
    assert sorted(got) == sorted(want), (pipe.stacks, want)


def test_a_pipeline_without_the_stack_entry_runs_its_tasks_one_by_one():
    g0, calls0, w0, _ = run(0)
    g1, calls1, w1, _ = run(1, 1, 2, 3)  # StubPipeline has no sliding_iterative_denoise_stack
    assert g1 == g0 and w1 == w0 and sorted(map(repr, calls1)) == sorted(map(repr, calls0))


def test_runner_uses_the_pipeline_and_checks_nothing_without_a_writer():
    s = make(None)
    s.result_writer = None
    SamplingRunner(s, prefetch_depth=2).inference()
    assert all(s.timestep_indices[c][f] == 9 for c in s.target_spa_labels for f in s.tem_labels)


def test_loader_and_writer_errors_are_raised_on_the_caller():
    def bad_writer(sample, output_dir=None):
        raise OSError("disk full")

    s = make(bad_writer)
    with pytest.raises(OSError, match="disk full"):
        run_round_pipelined(s, s.all_tasks[0], 0, 1)

    s = make(lambda *a, **k: None)
    orig = s.load_sample
    n = []

    def flaky(**task):
        n.append(1)
        if len(n) == 3:
            raise RuntimeError("decode failed")
        return orig(**task)

    s.load_sample = flaky
    with pytest.raises(RuntimeError, match="decode failed"):
        run_round_pipelined(s, s.all_tasks[0], 0, 1)
    assert len(s.pipelines[0].calls) <= 2  # nothing is denoised past the failure
