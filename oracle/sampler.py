"""Oracle sampler + runner: restatement of the reference's task loop.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``/root/reference/src/samplers/sliding_iterative_sampler.py``
  * ``:49-96``   labels, validation, the ``latents[spa][tem] = None`` / ``timestep_indices[spa][tem] = 0`` grid
  * ``:102-153`` ``load_sample``: input / target rows, ``dataset.get_item``, cond masks, grid gather under the lock,
                 ``latents = None`` iff the first target's index is 0 else ``torch.stack``
  * ``:155-190`` ``denoise``: the keyword set handed to ``pipeline.sliding_iterative_denoise``, write-back of
                 ``latent.cpu()`` / ``timestep_index.item()`` per cell, ``images.float().cpu()``
  * ``:192-212`` ``prepare_tasks`` / ``execute_one_task`` / ``execute_tasks``
and ``/root/reference/src/samplers/sampling_runner.py:18-62`` (per-round queues, one thread per pipeline).

Why it exists: the GPU box has no ``/root/reference``, so "the reference's own sampler drives the HIP pipeline" is tested
there with this restatement (tests/test_reference_protocol_gpu.py).  It is PINNED in the build container against the
reference's real classes imported through ``oracle/refshim.py``: tests/test_reference_protocol.py drives both with the
same recording pipeline and requires identical call traces (keyword names, tensor shapes / dtypes / devices, None-ness)
and identical final grids.
"""
from __future__ import annotations

from collections import defaultdict
from queue import Empty, Queue
from threading import Lock, Thread
from typing import Callable, List, Optional

import torch


class OracleSampler:
    def __init__(self, dataset, pipelines: list, output_dir: str = "./results/debug", window_size: int = 12,
                 sliding_stride: int = 1, sliding_shift: int = 0, bidirectional: bool = True, num_denoising_steps: int = 1,
                 alternation_rounds: int = 3, guidance_scale: float = 2.0, spa_label_range=(0, 48, 1),
                 tem_label_range=(0, 150, 1), spa_labels=None, tem_labels=None, input_spa_labels=(1, 13, 25, 37),
                 save: Optional[Callable] = None):
        self.dataset, self.pipelines, self.output_dir = dataset, pipelines, output_dir
        self.window_size, self.sliding_stride, self.sliding_shift = window_size, sliding_stride, sliding_shift
        self.bidirectional, self.num_denoising_steps = bidirectional, num_denoising_steps
        self.alternation_rounds, self.guidance_scale = alternation_rounds, guidance_scale
        self.save = save  # the reference hard-wires save_sampling_results (:204)
        if spa_labels is not None:  # :49-55
            self.spa_labels = [f"{int(i):02d}" for i in spa_labels]
        elif spa_label_range is not None:
            b, e, s = spa_label_range
            self.spa_labels = [f"{int(i):02d}" for i in range(b, e, s)]
        else:
            raise ValueError("spa_labels or spa_label_range must be provided")
        if tem_labels is not None:  # :57-63
            self.tem_labels = [f"{int(i):06d}" for i in tem_labels]
        elif tem_label_range is not None:
            b, e, s = tem_label_range
            self.tem_labels = [f"{int(i):06d}" for i in range(b, e, s)]
        else:
            raise ValueError("tem_labels or tem_label_range must be provided")
        self.input_spa_labels = [f"{int(i):02d}" for i in input_spa_labels]
        self.target_spa_labels = [c for c in self.spa_labels if c not in self.input_spa_labels]
        nt, nf = len(self.target_spa_labels), len(self.tem_labels)
        if window_size > nt:  # :71-88
            raise ValueError(f"window_size(={window_size}) must be <= len(target_spa_labels)(={nt})")
        if nt % sliding_stride != 0:
            raise ValueError(f"len(target_spa_labels)(={nt}) % sliding_stride(={sliding_stride}) must be 0")
        if nf % sliding_stride != 0:
            raise ValueError(f"len(tem_labels)(={nf}) % sliding_stride(={sliding_stride}) must be 0")
        if alternation_rounds > 1 and window_size > nf:
            raise ValueError(f"window_size(={window_size}) must be <= the number of tem_labels(={nf}) when alternation_rounds > 1")
        self.latents, self.timestep_indices = defaultdict(dict), defaultdict(dict)  # :91-96
        for c in self.spa_labels:
            for f in self.tem_labels:
                self.latents[c][f] = None
                self.timestep_indices[c][f] = 0
        self.lock = Lock()
        domains = (["spatial", "temporal"] * alternation_rounds)[:alternation_rounds]  # :192-199
        self.all_tasks = [[{"alt": i + 1, "domain": d, "domain_label": lab}
                           for lab in (self.tem_labels if d == "spatial" else self.target_spa_labels)]
                          for i, d in enumerate(domains)]

    def load_sample(self, alt: int, domain: str, domain_label: str) -> dict:  # :102-153
        if domain == "spatial":
            spa, tem = self.spa_labels, [domain_label]
            input_indices = torch.tensor([self.spa_labels.index(c) for c in self.input_spa_labels])
            target_indices = torch.tensor([self.spa_labels.index(c) for c in self.target_spa_labels])
        elif domain == "temporal":
            spa, tem = [domain_label], self.tem_labels
            half = len(self.tem_labels)
            input_indices, target_indices = torch.tensor(list(range(half))), torch.tensor(list(range(half, 2 * half)))
        sample = self.dataset.get_item(scene_label=self.dataset.scene_label, spa_labels=spa, tem_labels=tem,
                                       input_spa_labels=self.input_spa_labels)
        sample.update(alt=alt, domain=domain, domain_label=domain_label, input_indices=input_indices,
                      target_indices=target_indices)
        sample["cond_masks"][...] = 1.0
        sample["cond_masks"][input_indices, ...] = 0.0
        with self.lock:
            lats = [self.latents[c][f] for _, c, f in sample["labels"]]
            idx = [self.timestep_indices[c][f] for _, c, f in sample["labels"]]
        idx = torch.tensor(idx)
        sample["latents"] = None if idx[target_indices[0]] == 0 else torch.stack(lats, dim=0)
        sample["timestep_indices"] = idx
        return sample

    @torch.no_grad()
    def denoise(self, sample: dict, pipe_idx: int = 0) -> dict:  # :155-190
        pipeline = self.pipelines[pipe_idx]
        result = pipeline.sliding_iterative_denoise(
            pixel_values=sample["pixel_values"], plucker_embeds=sample["plucker_embeds"], skeletons=sample["skeletons"],
            cond_masks=sample["cond_masks"], latents=sample["latents"], domain=sample["domain"],
            timestep_indices=sample["timestep_indices"], window_size=self.window_size, sliding_stride=self.sliding_stride,
            sliding_shift=self.sliding_shift, bidirectional=self.bidirectional, num_denoising_steps=self.num_denoising_steps,
            alternation_rounds=self.alternation_rounds, guidance_scale=self.guidance_scale,
            tqdm=lambda it, **kw: it)  # the reference passes partial(tqdm, desc=...)
        with self.lock:
            for (_, c, f), latent, ti in zip(sample["labels"], result["latents"], result["timestep_indices"]):
                self.latents[c][f] = latent.cpu()
                self.timestep_indices[c][f] = ti.item()
        sample["images"] = result["images"].float().cpu()
        sample["timestep_indices"] = result["timestep_indices"].cpu()
        sample["fully_denoised"] = result["fully_denoised"].cpu()
        return sample

    def execute_one_task(self, task: dict, pipe_idx: int = 0):  # :201-204
        sample = self.denoise(self.load_sample(**task), pipe_idx=pipe_idx)
        if self.save is not None:
            self.save(sample, output_dir=self.output_dir)

    def execute_tasks(self):  # :206-212 (the completeness check belongs to the writer and is left to the caller)
        for tasks in self.all_tasks:
            for task in tasks:
                self.execute_one_task(task)


class OracleRunner:
    """sampling_runner.py:18-62: one queue per alternation round, one thread per pipeline draining it."""

    def __init__(self, sampler: OracleSampler):
        self.sampler = sampler

    def inference(self):
        s = self.sampler
        if len(s.pipelines) <= 1:  # :60-61
            s.execute_tasks()
            return
        for tasks in s.all_tasks:  # :18-43
            q: Queue = Queue()
            for t in tasks:
                q.put(t)
            errors: List[BaseException] = []

            def worker(pipe_idx):
                while True:
                    try:
                        task = q.get_nowait()
                    except Empty:
                        break
                    try:
                        s.execute_one_task(task, pipe_idx=pipe_idx)
                    except BaseException as e:  # noqa: BLE001  (a dead reference thread loses its error; tests want it)
                        errors.append(e)
                        break
            threads = [Thread(target=worker, args=(i,)) for i in range(len(s.pipelines))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
