"""Task execution across GPUs.

``SamplingRunner`` keeps the reference's contract (``/root/reference/src/samplers/sampling_runner.py``:
one Python thread per pipeline draining a per-round queue, barrier between alternation rounds).

``DistributedSamplingRunner`` is the MI355X-first form: ONE PROCESS PER GPU (torchrun / RCCL), the
round's tasks sharded round-robin over ranks, the latent grid resident in HBM, and -- instead of the
reference's implicit transpose through host RAM under a lock (sliding_iterative_sampler.py:142-147,
181-185) -- one explicit exchange at every round boundary in which each rank sends every peer exactly
the grid cells that peer's next-round tasks read.  No collective on the data path of a
task.  Works with backend "nccl" (= RCCL over xGMI) and "gloo" (CPU tests).
"""
from __future__ import annotations

import threading
from queue import Empty, Queue
from threading import Thread
from typing import Dict, List, Tuple

import torch

from .results import check_sampling_results
from .sampler import SlidingIterativeSampler


_tls = threading.local()


# Defaults of the GPU stage (profiles/r05_task_batch_streams.log): stacks of tasks of a round in flight per GPU, and tasks per stack of
# shared window calls.  Two streams of 2-task stacks are the fastest resident steady state (+2.4 % over three single tasks, three streams
# of stacks +1.9 %), but end to end two streams leave the GPU to ONE stack whenever the other is in a host-side phase (image upload,
# result packing): 23.8 against 24.5 latents/s on a 48 x 64 grid.  Three streams of stacks keep both.
DEFAULT_GPU_STREAMS = 3
DEFAULT_TASK_BATCH = 2


def _denoise_group(sampler: SlidingIterativeSampler, group: List[dict], pipe_idx: int) -> List[dict]:
    """One task, or a stack of tasks sharing their window calls (task_batch > 1)."""
    if len(group) == 1:
        return [sampler.denoise(group[0], pipe_idx=pipe_idx)]
    return sampler.denoise_stack(group, pipe_idx=pipe_idx)


def _denoise_on_own_stream(sampler: SlidingIterativeSampler, group: List[dict], pipe_idx: int) -> List[dict]:
    """GPU-stage worker: every worker thread owns one HIP stream (created on first use), so the kernels of concurrently
    denoised tasks interleave on the device."""
    dev = sampler.pipelines[pipe_idx].device
    if not (torch.cuda.is_available() and getattr(dev, "type", "cpu") == "cuda"):
        return _denoise_group(sampler, group, pipe_idx)
    torch.cuda.set_device(dev)
    st = getattr(_tls, "stream", None)
    if st is None or st.device != torch.device(dev):
        st = _tls.stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        return _denoise_group(sampler, group, pipe_idx)


def run_round_pipelined(sampler: SlidingIterativeSampler, tasks: List[dict], pipe_idx: int = 0, depth: int = 1,
                        writers: int = 1, gpu_streams: int = 1, writer_pool=None, task_batch: int = 1) -> None:
    """Execute the tasks of ONE alternation round on one pipeline as a 3-stage software pipeline:

        loader pool: load_sample(task i+1 .. i+depth)  ||  denoise(task i) on the GPU  ||  writer pool: save(task < i)

    ``gpu_streams`` > 1 denoises that many tasks of the round concurrently, each on its own HIP stream (one worker
    thread per stream, all sharing the pipeline's weights).  One task's kernels leave CUs idle -- partial last rounds
    of workgroups, the 120-460-workgroup launches of the two deepest UNet levels, the drain/fill at every one of the
    ~1000 kernel boundaries of a window call -- and a second task's kernels fill them: +7 % denoised latents/s with 2
    streams, +8 % with 3 (profiles/r01_task_streams.log).  Every task computes exactly what it computes alone, so the
    grid is bitwise the same as with one stream.

    ``task_batch`` > 1 hands every GPU worker that many consecutive tasks of the round at once; they run through SHARED window
    calls (sampler.denoise_stack -> pipeline.sliding_iterative_denoise_stack: the tasks' tensors stacked along the frame axis,
    every call carrying task_batch x F frames).  Each task still computes exactly what it computes alone (bitwise, GPU test
    `modelcheck task_stack_*`); what is gained is the GEMM / convolution grids of the two deepest UNet levels (120-460
    workgroups for one task) and half of the launches: 74.4-74.7 ms per 2 spatial + 1 temporal window calls with 2 streams of
    2-task stacks against 76.2-76.4 with 3 streams of single tasks on the same box (profiles/r05_task_batch_streams.log).

    The reference runs load -> denoise -> save serially per worker thread (sliding_iterative_sampler.py:201-204), so
    the GPU idles during the host-side decode/resize of 3N images and the JPEG writes (measured on
    the 576x320 synthetic task: load 7.0 s, denoise 1.8 s, save 1.4 s).  Tasks of a round read and write disjoint
    target cells of the grid (spatial: one frame each; temporal: one target camera each; the shared input-camera
    cells are conditioning rows whose grid value is never consumed), so loading task i+k before task i has
    written back is equivalent to the serial order; rounds are never overlapped.  Samples are denoised in task
    order whatever order the loads finish in.
    ``depth`` = tasks loaded ahead, each on its own thread (host memory: one task's tensors each); 0 = serial.
    ``writer_pool`` (imgwrite.WriterPool): samples that carry a ``_package`` (sampler.device_results) are encoded by its writer
    PROCESSES instead of the writer threads -- JPEG / WebP encoding off the interpreter lock the launch threads need."""
    task_batch = max(1, int(task_batch))
    if (depth <= 0 and gpu_streams <= 1 and task_batch <= 1) or len(tasks) <= 1:
        for t in tasks:
            sampler.execute_one_task(t, pipe_idx=pipe_idx)
        return
    from concurrent.futures import ThreadPoolExecutor
    gpu_streams = max(1, gpu_streams)
    if task_batch > 1:  # stacks as even as the round allows: 48 tasks by 3 -> 16 stacks of 3; 44 -> 14 of 3 + 1 of 2; 5 -> 3 + 2
        n_groups = -(-len(tasks) // task_batch)
        sizes = [len(tasks) // n_groups + (1 if g < len(tasks) % n_groups else 0) for g in range(n_groups)]
    else:
        sizes = [1] * len(tasks)
    depth = max(depth, gpu_streams * max(sizes))  # every GPU stream needs its loaded samples to start on

    pin = torch.cuda.is_available() and getattr(sampler.pipelines[pipe_idx].device, "type", "cpu") == "cuda"

    def load(**task):
        sample = sampler.load_sample(**task)
        if pin:  # page-locked staging on the loader thread: the pipeline's H2D copies become plain DMA
            for k, v in sample.items():
                if torch.is_tensor(v) and v.device.type == "cpu" and v.numel() > 1 << 16:
                    sample[k] = v.pin_memory()
        return sample

    loaders = ThreadPoolExecutor(max_workers=depth, thread_name_prefix="dm4d-loader")
    savers = ThreadPoolExecutor(max_workers=max(1, writers), thread_name_prefix="dm4d-writer") \
        if sampler.result_writer is not None else None
    gpu = ThreadPoolExecutor(max_workers=gpu_streams, thread_name_prefix="dm4d-gpu") if gpu_streams > 1 else None
    pending: List = []   # load futures, in task order
    running: List = []   # denoise futures (gpu_streams > 1), in task order
    saves: List = []
    nxt = 0

    def hand_to_writers(sample: dict) -> None:
        nonlocal saves
        if savers is None:
            return
        saves = [f for f in saves if not (f.done() and f.exception() is None)]
        for f in saves:
            if f.done():
                f.result()  # surface a writer error as soon as it is known
        while len(saves) > depth + max(writers, len(getattr(writer_pool, "_procs", ()))):  # bound what is held for writing
            saves.pop(0).result()
        if writer_pool is not None and sample.get("_package") is not None:
            saves.append(writer_pool.submit(sample.pop("_package")))
        else:
            saves.append(savers.submit(sampler.result_writer, sample, output_dir=sampler.output_dir))

    def next_group() -> List[dict]:
        """The next stack's loaded samples, in task order (a loader error is re-raised here, on the caller's thread)."""
        nonlocal nxt
        group = []
        for _ in range(sizes.pop(0)):
            group.append(pending.pop(0).result())
            if nxt < len(tasks):
                pending.append(loaders.submit(load, **tasks[nxt]))
                nxt += 1
        return group

    try:
        while nxt < len(tasks) and len(pending) < depth:
            pending.append(loaders.submit(load, **tasks[nxt]))
            nxt += 1
        while pending or running:
            if gpu is None:
                done = _denoise_group(sampler, next_group(), pipe_idx)
            else:
                while pending and len(running) < gpu_streams:
                    running.append(gpu.submit(_denoise_on_own_stream, sampler, next_group(), pipe_idx))
                done = running.pop(0).result()  # task order; a worker error surfaces here
            for sample in done:
                hand_to_writers(sample)
        for f in saves:
            f.result()
    finally:
        for f in pending + running:
            f.cancel()
        if gpu is not None:
            gpu.shutdown(wait=True)
        loaders.shutdown(wait=True)
        if savers is not None:
            savers.shutdown(wait=True)


def _make_writer_pool(sampler, writer_processes: int):
    """Writer processes make sense only for packaged results (sampler.device_results on a HIP device)."""
    if writer_processes <= 0 or sampler.result_writer is None or not getattr(sampler, "device_results", False):
        return None
    from .imgwrite import WriterPool
    return WriterPool(writer_processes)


class SamplingRunner:
    def __init__(self, sampler: SlidingIterativeSampler, prefetch_depth: int = 2, writers: int = 2, gpu_streams: int = DEFAULT_GPU_STREAMS,
                 writer_processes: int = 0, task_batch: int = DEFAULT_TASK_BATCH):
        self.sampler = sampler
        self.prefetch_depth, self.writers, self.gpu_streams = prefetch_depth, writers, gpu_streams
        self.writer_processes = writer_processes
        self.task_batch = task_batch  # tasks of a round per stack of shared window calls (run_round_pipelined)

    def prepare_task_queues(self):
        self.task_queues = []
        for tasks in self.sampler.all_tasks:
            q = Queue()
            for t in tasks:
                q.put(t)
            self.task_queues.append(q)

    def parallel_execute_tasks(self, task_queue: Queue):
        errors: List[BaseException] = []

        def _worker(q: Queue, pipe_idx: int):
            while True:
                try:
                    task = q.get_nowait()
                except Empty:
                    break
                try:
                    self.sampler.execute_one_task(task, pipe_idx=pipe_idx)
                except BaseException as e:  # surface worker failures instead of ending the round short
                    errors.append(e)
                    break

        threads = [Thread(target=_worker, args=(task_queue, i)) for i in range(len(self.sampler.pipelines))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def inference(self):
        s = self.sampler
        if len(s.pipelines) > 1:
            self.prepare_task_queues()
            for q in self.task_queues:
                self.parallel_execute_tasks(q)
            if s.result_writer is not None and not check_sampling_results(s.spa_labels, s.tem_labels, s.output_dir):
                raise ValueError("Sampling failed.")
        else:
            pool = _make_writer_pool(s, self.writer_processes)
            try:
                for tasks in s.all_tasks:
                    run_round_pipelined(s, tasks, 0, self.prefetch_depth, self.writers, self.gpu_streams, pool, self.task_batch)
            finally:
                if pool is not None:
                    pool.shutdown()
            if s.result_writer is not None and not check_sampling_results(s.spa_labels, s.tem_labels, s.output_dir):
                raise ValueError("Sampling failed.")


class DistributedSamplingRunner:
    """One process per GPU.  Every rank builds the same sampler (same task lists); rank r executes
    ``sampler.partition(round, r, world)`` with its single pipeline, then the grid is re-partitioned.

    Grid bookkeeping is replicated, data is not: every rank simulates, for ALL ranks, which round's value of which cell
    each rank holds (``_holds``) and which rank wrote a cell last (``_written``).  That state is a pure function of the
    task lists, so the ranks agree on every transfer without negotiating, and a rank that ran no task in a round (more
    GPUs than frames or target cameras) still takes part in the exchange: it is sent every cell its next tasks read,
    whoever holds them.

    Load balance.  The reference's pipelines drain ONE queue (sampling_runner.py:27-33), so a slower GPU simply takes fewer
    tasks.  Across processes the tasks of a round have to be known before the round (the exchange ships each rank the cells
    its next tasks read), so the balance is applied between rounds instead of inside them: every rank measures its task rate,
    the rates are all-gathered at the round boundary and `balance=True` deals the next round's tasks in proportion to them
    (`weighted_deal`: a pure function of the gathered numbers, so the replicated bookkeeping stays in agreement).  Rates within
    10 % of each other are treated as equal, which reproduces the round-robin deal exactly.

    Modes (``runner.mode`` of inference.py; the scheduling seam is the reference's runner, sampling_runner.py:26-62):
      * ``task`` (default): every task runs on one rank, as above.
      * ``frame-shard``: every task runs on ALL ranks together -- each window call split over them by frames, K/V all-gathered inside
        the 3-D attention layers (parallel.FrameShard; BASELINE.json configs[3]).  The grid stays replicated, the
        exchange has nothing to send.  The latency mode: one task finishes world times sooner, throughput is lower.
      * ``hybrid``: full waves of a round run task-parallel (a sub-group's tail starts when ITS ranks are done with their main
        waves: the sub-groups exist before the first round); when the LAST wave would leave at least half of the ranks idle
        (r = tasks mod world, 0 < r <= world / 2 -- the 44-camera temporal round on 8 GPUs: 5 waves + 4 tasks) those r tasks run
        frame-sharded on r sub-groups of P = world / r ranks (P a power of two dividing the window's frame count) instead of on r
        single ranks beside world - r idle ones.  Also what a round with fewer tasks than ranks gets.
    Ranks of a group pass identical arguments (same sample, a task-derived noise seed), the group leader alone writes results."""

    MODES = ("task", "frame-shard", "hybrid")

    def __init__(self, sampler: SlidingIterativeSampler, group=None, prefetch_depth: int = 2, writers: int = 2,
                 gpu_streams: int = DEFAULT_GPU_STREAMS, balance: bool = True, writer_processes: int = 0, mode: str = "task",
                 task_batch: int = DEFAULT_TASK_BATCH):
        import torch.distributed as dist
        if mode not in self.MODES:
            raise ValueError(f"Unsupported runner mode: {mode}. Supported modes are {', '.join(self.MODES)}.")
        self.dist = dist
        self.mode = mode
        if mode != "task" and group is not None and dist.is_initialized() and dist.get_world_size(group) != dist.get_world_size():
            # the sub-groups of the frame-sharding modes are made with dist.new_group, which is collective over the DEFAULT group: ranks
            # outside `group` would never make the call and the first tail wave would hang
            raise ValueError(f"runner mode '{mode}' needs the runner's group to be the whole world (new_group is collective over it); "
                             "use mode='task' on a sub-group, or one runner over all ranks")
        self._subgroups: Dict[int, list] = {}  # P -> [(ranks, process group)] in sub-group order
        self.sampler = sampler
        self.prefetch_depth, self.writers, self.gpu_streams = prefetch_depth, writers, gpu_streams
        self.writer_processes = writer_processes
        self.task_batch = task_batch  # stacks of tasks on this rank (the frame-sharded tail of a round runs task by task)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._written: Dict[Tuple[str, str], Tuple[int, int]] = {}  # cell -> (round of the last write, sending rank)
        self._holds: List[Dict[Tuple[str, str], int]] = [dict() for _ in range(self.world)]  # rank -> {cell: round held}
        self._proto = None  # (shape, dtype) of a grid cell, agreed once
        self.balance = balance
        self._assign: Dict[int, List[List[dict]]] = {}  # round -> tasks per rank (absent: the sampler's round-robin deal)
        self.rates: List[float] = [1.0] * self.world     # relative task rates measured in the last round that ran tasks

    MIN_ROUND_SECONDS = 0.25  # rounds shorter than this do not update a rank's rate

    @staticmethod
    def weighted_deal(n_tasks: int, rates: List[float]) -> List[List[int]]:
        """Task indices per rank: task k goes to the rank that would finish it first, (assigned + 1) / rate minimal, ties to
        the lowest rank.  Equal rates give the round-robin deal [r::world]."""
        world = len(rates)
        out: List[List[int]] = [[] for _ in range(world)]
        for k in range(n_tasks):
            r = min(range(world), key=lambda q: ((len(out[q]) + 1) / rates[q], q))
            out[r].append(k)
        return out

    # -- which tasks of a round run alone on a rank ("main") and which on a group of ranks ("tail") ------------------------
    def _window_frames(self, domain: str) -> int:
        """Frames of one window call: the quantity a shard group divides (spatial: inputs + window, temporal: 2 x window)."""
        s = self.sampler
        return s.sweep.window_size + (len(s.input_spa_labels) if domain == "spatial" else s.sweep.window_size)

    def _split_round(self, round_index: int) -> Tuple[List[dict], List[dict], int]:
        """(main tasks, tail tasks, ranks per tail task) -- a pure function of the task lists, the world size and the mode."""
        tasks = self.sampler.all_tasks[round_index]
        if self.mode == "task" or self.world == 1 or not tasks:
            return tasks, [], 1
        frames = self._window_frames(tasks[0]["domain"])
        if self.mode == "frame-shard":
            if frames % self.world != 0:
                raise ValueError(f"runner mode 'frame-shard': a window of {frames} frames cannot be split over {self.world} ranks")
            return [], tasks, self.world
        r = len(tasks) % self.world
        if r == 0 or 2 * r > self.world:
            return tasks, [], 1
        width = 1
        while width * 2 <= self.world // r and self.world % (width * 2) == 0 and frames % (width * 2) == 0:
            width *= 2
        if width == 1:
            return tasks, [], 1
        return tasks[:len(tasks) - r], tasks[len(tasks) - r:], width

    def _group_ranks(self, width: int, g: int) -> List[int]:
        return list(range(g * width, (g + 1) * width))

    def tail_of(self, round_index: int) -> List[Tuple[dict, List[int]]]:
        """(task, ranks that run it together) for the round's tail tasks; tail task k goes to sub-group k mod (world / width)."""
        _, tail, width = self._split_round(round_index)
        n_groups = self.world // width
        return [(t, self._group_ranks(width, k % n_groups)) for k, t in enumerate(tail)]

    def _subgroup(self, ranks: List[int]):
        """The process group of a set of consecutive ranks.  Groups of one width are created together, by every rank, in sub-group
        order the first time that width is needed (new_group is collective over the default group)."""
        width = len(ranks)
        if width == self.world:
            return self.group
        if width not in self._subgroups:
            base = self.dist.get_process_group_ranks(self.group) if self.group is not None else list(range(self.world))
            self._subgroups[width] = [(self._group_ranks(width, g), self.dist.new_group([base[r] for r in self._group_ranks(width, g)]))
                                      for g in range(self.world // width)]
        return next(grp for rk, grp in self._subgroups[width] if rk == ranks)

    def tasks_of(self, round_index: int, rank: int) -> List[dict]:
        """The tasks `rank` runs ALONE in the round (its share of the main tasks)."""
        a = self._assign.get(round_index)
        if a is not None:
            return a[rank]
        main = self._split_round(round_index)[0]
        if len(main) == len(self.sampler.all_tasks[round_index]):
            return self.sampler.partition(round_index, rank, self.world)
        return main[rank::self.world]

    def _plan_next_round(self, round_index: int, seconds: float, n_done: int) -> None:
        """Gather (seconds, tasks) of the round that just ran and deal round_index + 1 accordingly."""
        nxt = round_index + 1
        if not self.balance or self.world == 1 or nxt >= len(self.sampler.all_tasks):
            return
        stats = [None] * self.world
        self.dist.all_gather_object(stats, (float(seconds), int(n_done)), group=self.group)
        ok = [n > 0 and t >= self.MIN_ROUND_SECONDS for t, n in stats]  # a round of milliseconds measures the scheduler, not the GPU
        measured = [n / t for (t, n), good in zip(stats, ok) if good]
        if measured:
            mean = sum(measured) / len(measured)
            # ranks without a (long enough) round keep their last rate; 10 % steps: noise does not reshuffle the deal
            self.rates = [max(0.1, round((n / t) / mean, 1)) if good else self.rates[q]
                          for q, ((t, n), good) in enumerate(zip(stats, ok))]
        tasks = self._split_round(nxt)[0]  # the tasks that run on single ranks; a tail (hybrid / frame-shard) is dealt to groups
        self._assign[nxt] = [[tasks[k] for k in idx] for idx in self.weighted_deal(len(tasks), self.rates)]

    def _input_camera_of(self, target_label: str) -> str:
        ds = self.sampler.dataset
        inputs = [int(c) for c in self.sampler.input_spa_labels]
        near = getattr(ds, "nearest_input_camera", None)
        return f"{near(int(target_label), inputs):02d}" if near is not None else None

    # cells a task reads from / writes to the grid: (spa_label, tem_label)
    def _task_cells(self, task: dict) -> List[Tuple[str, str]]:
        s = self.sampler
        if task["domain"] == "spatial":
            return [(c, task["domain_label"]) for c in s.spa_labels]
        near = self._input_camera_of(task["domain_label"])
        cams = ([near] if near is not None else list(s.input_spa_labels)) + [task["domain_label"]]
        return [(c, t) for c in cams for t in s.tem_labels]

    def _owned_after(self, round_index: int, rank: int) -> List[Tuple[str, str]]:
        """TARGET cells whose authoritative copy lives on `rank` after `round_index`: the targets its tasks denoised.
        (Input-camera rows are re-encoded by every task that conditions on them, so their grid value is never read back
        as data -- any rank's copy will do.)"""
        s = self.sampler
        cells = []
        mine = self.tasks_of(round_index, rank) + [t for t, ranks in self.tail_of(round_index) if ranks[0] == rank]  # group leaders own
        for t in mine:
            if t["domain"] == "spatial":
                cells += [(c, t["domain_label"]) for c in s.target_spa_labels]
            else:
                cells += [(t["domain_label"], f) for f in s.tem_labels]
        return cells

    def _record_round(self, round_index: int) -> None:
        """Replay what every rank wrote in `round_index` into the replicated bookkeeping (lowest writer rank sends)."""
        for r in range(self.world):
            for t in self.tasks_of(round_index, r):
                for cell in self._task_cells(t):
                    w = self._written.get(cell)
                    if w is None or w[0] < round_index:
                        self._written[cell] = (round_index, r)
                    self._holds[r][cell] = round_index
        for t, ranks in self.tail_of(round_index):  # every rank of the group ends the task with the same cells; the leader sends
            for cell in self._task_cells(t):
                w = self._written.get(cell)
                if w is None or w[0] < round_index:
                    self._written[cell] = (round_index, ranks[0])
                for r in ranks:
                    self._holds[r][cell] = round_index

    def _cell_proto(self):
        """(shape, dtype) of a grid cell.  Ranks that hold no cell yet learn it from the others (one small
        all_gather_object, first exchange only)."""
        if self._proto is None:
            mine = next((l for d in self.sampler.latents.values() for l in d.values() if l is not None), None)
            info = None if mine is None else (tuple(mine.shape), str(mine.dtype).replace("torch.", ""))
            infos = [None] * self.world
            self.dist.all_gather_object(infos, info, group=self.group)
            info = next((i for i in infos if i is not None), None)
            if info is None:
                raise RuntimeError("grid exchange: no rank holds a latent cell")
            self._proto = (info[0], getattr(torch, info[1]))
        return self._proto

    def _exchange_device(self) -> torch.device:
        backend = self.dist.get_backend(self.group)
        dev = getattr(self.sampler.pipelines[0], "device", torch.device("cpu"))
        return torch.device(dev) if backend == "nccl" else torch.device("cpu")

    def exchange(self, round_index: int):
        """After round `round_index`: every rank receives, from the rank that wrote it last, each cell its round+1 tasks
        read and whose latest value it does not hold."""
        s, dist = self.sampler, self.dist
        self._record_round(round_index)
        if round_index + 1 >= len(s.all_tasks) or self.world == 1:
            return
        send_cells: Dict[int, list] = {q: [] for q in range(self.world)}
        recv_cells: Dict[int, list] = {q: [] for q in range(self.world)}
        for q in range(self.world):
            need = set()
            for t in self.tasks_of(round_index + 1, q) + [t for t, ranks in self.tail_of(round_index + 1) if q in ranks]:
                need.update(self._task_cells(t))
            for cell in sorted(need):  # deterministic cell order on both sides of every pair
                w = self._written.get(cell)
                if w is None or self._holds[q].get(cell) == w[0]:
                    continue  # never written yet, or q already holds the latest value
                src = w[1]
                if src == self.rank:
                    send_cells[q].append(cell)
                if q == self.rank:
                    recv_cells[src].append(cell)
                self._holds[q][cell] = w[0]
        shape, dtype = self._cell_proto()
        xdev = self._exchange_device()
        ops_, recv_bufs = [], {}
        for q in range(self.world):
            if send_cells[q]:
                # one stacked tensor per peer.  Under RCCL the cells already live on the exchange device (`.to` is a no-op); under gloo
                # with HIP pipelines the copies are device-to-host and BLOCKING: a non_blocking D2H copy returns before the bytes have
                # landed, and torch.stack / isend would read them too early
                buf = torch.stack([s.latents[c][t].to(xdev) for c, t in send_cells[q]])
                idx = torch.tensor([s.timestep_indices[c][t] for c, t in send_cells[q]], dtype=torch.int64, device=xdev)
                ops_ += [dist.P2POp(dist.isend, buf, q, self.group), dist.P2POp(dist.isend, idx, q, self.group)]
            if recv_cells[q]:
                buf = torch.empty((len(recv_cells[q]),) + tuple(shape), dtype=dtype, device=xdev)
                idx = torch.empty(len(recv_cells[q]), dtype=torch.int64, device=xdev)
                recv_bufs[q] = (buf, idx)
                ops_ += [dist.P2POp(dist.irecv, buf, q, self.group), dist.P2POp(dist.irecv, idx, q, self.group)]
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()
        cell_dev = getattr(s.pipelines[0], "device", xdev)
        for q, (buf, idx) in recv_bufs.items():
            buf = buf.to(cell_dev)
            idx = idx.cpu()
            for k, (c, t) in enumerate(recv_cells[q]):
                s.latents[c][t] = buf[k]
                s.timestep_indices[c][t] = int(idx[k])

    def _run_sharded(self, task: dict, ranks: List[int], group) -> None:
        """One task on the ranks of `group`: same sample and noise seed everywhere, every window call split by frames
        (parallel.FrameShard), one task at a time (collectives have to be issued in one order); the leader writes the results."""
        from .parallel import FrameShard
        s = self.sampler
        keep = s.result_writer
        s.frame_shard = FrameShard(group)
        if self.rank != ranks[0]:  # a follower: nobody writes or reads its images, so it neither decodes nor copies them to the host
            s.result_writer = None
            s.shard_follower = True
        try:
            s.execute_one_task(task)
        finally:
            s.frame_shard, s.result_writer, s.shard_follower = None, keep, False

    def inference(self):
        s = self.sampler
        import time
        pool = _make_writer_pool(s, self.writer_processes)
        # The sub-groups of every tail width the job will need are made HERE, by all ranks together (new_group is collective over the
        # default group): inside a round nothing then makes a rank wait for ranks outside its own sub-group, so the tail of a sub-group
        # starts as soon as ITS members have finished their main waves -- not when the slowest rank of the world has
        for width in sorted({self._split_round(ri)[2] for ri in range(len(s.all_tasks)) if self._split_round(ri)[1]}):
            self._subgroup(self._group_ranks(width, 0))
        self.timeline: List[Tuple[int, str, float]] = []  # (round, "main_end" | "tail_start" | "tail_end", time.time()) of this rank
        try:
            for ri in range(len(s.all_tasks)):
                mine = self.tasks_of(ri, self.rank)
                t0 = time.perf_counter()
                run_round_pipelined(s, mine, 0, self.prefetch_depth, self.writers, self.gpu_streams, pool, self.task_batch)
                dt = time.perf_counter() - t0
                self.timeline.append((ri, "main_end", time.time()))
                for task, ranks in self.tail_of(ri):  # the round's tail: tasks a group of ranks runs together, frame-sharded
                    if self.rank in ranks:            # (the group's own collectives are what its members meet at)
                        self.timeline.append((ri, "tail_start", time.time()))
                        self._run_sharded(task, ranks, self._subgroup(ranks))
                        self.timeline.append((ri, "tail_end", time.time()))
                self._plan_next_round(ri, dt, len(mine))  # before the exchange: it ships what the NEXT deal reads
                self.dist.barrier(self.group)
                self.exchange(ri)
        finally:
            if pool is not None:
                pool.shutdown()
        if s.result_writer is not None and self.rank == 0:
            if not check_sampling_results(s.spa_labels, s.tem_labels, s.output_dir):
                raise ValueError("Sampling failed.")
