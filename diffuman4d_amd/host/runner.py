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

from queue import Empty, Queue
from threading import Thread
from typing import Dict, List, Tuple

import torch

from .results import check_sampling_results
from .sampler import SlidingIterativeSampler


class SamplingRunner:
    def __init__(self, sampler: SlidingIterativeSampler):
        self.sampler = sampler

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
            s.execute_tasks()


class DistributedSamplingRunner:
    """One process per GPU.  Every rank builds the same sampler (same task lists); rank r executes
    ``sampler.partition(round, r, world)`` with its single pipeline, then the grid is re-partitioned."""

    def __init__(self, sampler: SlidingIterativeSampler, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.sampler = sampler
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    # cells a task reads from / writes to the grid: (spa_label, tem_label)
    def _task_cells(self, task: dict) -> List[Tuple[str, str]]:
        s = self.sampler
        if task["domain"] == "spatial":
            return [(c, task["domain_label"]) for c in s.spa_labels]
        cams = list(s.input_spa_labels) + [task["domain_label"]]  # the nearest input camera is one of these
        return [(c, t) for c in cams for t in s.tem_labels]

    def _owned_after(self, round_index: int, rank: int) -> List[Tuple[str, str]]:
        """Cells whose authoritative copy lives on `rank` after `round_index` (target cells only: input-camera
        rows are re-encoded by every task that conditions on them, so their grid value is never read back
        as data -- they only need to be non-None; owners = the lowest rank that touched them)."""
        s = self.sampler
        cells = []
        for t in s.partition(round_index, rank, self.world):
            if t["domain"] == "spatial":
                cells += [(c, t["domain_label"]) for c in s.spa_labels]
            else:
                cells += [(t["domain_label"], f) for f in s.tem_labels]
        return cells

    def exchange(self, round_index: int):
        """After round `round_index`: send each peer the cells its round+1 tasks read, receive ours."""
        s, dist = self.sampler, self.dist
        if round_index + 1 >= len(s.all_tasks) or self.world == 1:
            return
        owner: Dict[Tuple[str, str], int] = {}
        for r in range(self.world):
            for cell in self._owned_after(round_index, r):
                owner.setdefault(cell, r)
        need = {r: set() for r in range(self.world)}
        for r in range(self.world):
            for t in s.partition(round_index + 1, r, self.world):
                for cell in self._task_cells(t):
                    if cell in owner:
                        need[r].add(cell)
        # deterministic cell order on both sides of every pair
        send_cells = {q: sorted(c for c in need[q] if owner[c] == self.rank and q != self.rank) for q in range(self.world)}
        recv_cells = {q: sorted(c for c in need[self.rank] if owner[c] == q and q != self.rank) for q in range(self.world)}
        proto = next(l for d in s.latents.values() for l in d.values() if l is not None)
        ops_, recv_bufs = [], {}
        for q in range(self.world):
            if send_cells[q]:
                buf = torch.stack([s.latents[c][t] for c, t in send_cells[q]]).contiguous()
                idx = torch.tensor([s.timestep_indices[c][t] for c, t in send_cells[q]], dtype=torch.int64, device=buf.device)
                ops_ += [dist.P2POp(dist.isend, buf, q, self.group), dist.P2POp(dist.isend, idx, q, self.group)]
            if recv_cells[q]:
                buf = torch.empty((len(recv_cells[q]),) + tuple(proto.shape), dtype=proto.dtype, device=proto.device)
                idx = torch.empty(len(recv_cells[q]), dtype=torch.int64, device=proto.device)
                recv_bufs[q] = (buf, idx)
                ops_ += [dist.P2POp(dist.irecv, buf, q, self.group), dist.P2POp(dist.irecv, idx, q, self.group)]
        if ops_:
            for req in dist.batch_isend_irecv(ops_):
                req.wait()
        for q, (buf, idx) in recv_bufs.items():
            for k, (c, t) in enumerate(recv_cells[q]):
                s.latents[c][t] = buf[k]
                s.timestep_indices[c][t] = int(idx[k])

    def inference(self):
        s = self.sampler
        for ri in range(len(s.all_tasks)):
            for task in s.partition(ri, self.rank, self.world):
                s.execute_one_task(task, pipe_idx=0)
            self.dist.barrier(self.group)
            self.exchange(ri)
        if s.result_writer is not None and self.rank == 0:
            if not check_sampling_results(s.spa_labels, s.tem_labels, s.output_dir):
                raise ValueError("Sampling failed.")
