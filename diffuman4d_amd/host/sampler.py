"""SlidingIterativeSampler: the (camera x frame) latent grid and its task list.

Mirror of ``/root/reference/src/samplers/sliding_iterative_sampler.py`` (ctor kwargs = the Hydra
sampler config, :16-35; validation + ``ValueError``s :71-88; grid :91-96; ``load_sample`` :102-153;
``denoise`` :155-190; ``prepare_tasks`` :192-199; ``execute_one_task`` / ``execute_tasks`` :201-212).

Differences that do not change results:
  * grid cells keep the tensor the pipeline returned (device resident) instead of ``latent.cpu()``;
  * ``result_writer`` is injectable (the reference hard-wires ``save_sampling_results``);
  * ``partition(rank, world)`` exposes the per-round task sharding used by the one-process-per-GPU
    runner (``DistributedSamplingRunner``); the reference shards the same task lists over threads.
"""
from __future__ import annotations

from collections import defaultdict
from functools import partial
from threading import Lock
from typing import Callable, Dict, List, Optional, Sequence

import torch

try:  # tqdm is cosmetic
    from tqdm import tqdm as _tqdm
except Exception:  # pragma: no cover
    def _tqdm(it, **kw):
        return it


class SlidingIterativeSampler:
    def __init__(
        self,
        dataset,
        pipelines: list,
        output_dir: str = "./results/debug",
        # denoising args
        window_size: int = 12,
        sliding_stride: int = 1,
        sliding_shift: int = 0,
        bidirectional: bool = True,
        num_denoising_steps: int = 1,
        alternation_rounds: int = 3,
        guidance_scale: float = 2.0,
        # sampling range args
        spa_label_range: Optional[Sequence[int]] = (0, 48, 1),
        tem_label_range: Optional[Sequence[int]] = (0, 150, 1),
        spa_labels: Optional[Sequence[int]] = None,
        tem_labels: Optional[Sequence[int]] = None,
        input_spa_labels: Sequence[int] = (1, 13, 25, 37),
        result_writer: Optional[Callable] = None,
        # MI355X pipeline extensions (off = the reference's behaviour; need a pipeline that accepts the kwargs)
        vae_cache: bool = False,
        decode_policy: str = "all",
    ):
        self.dataset = dataset
        self.pipelines = pipelines
        self.output_dir = output_dir
        self.window_size = window_size
        self.sliding_stride = sliding_stride
        self.sliding_shift = sliding_shift
        self.bidirectional = bidirectional
        self.num_denoising_steps = num_denoising_steps
        self.alternation_rounds = alternation_rounds
        self.guidance_scale = guidance_scale
        if result_writer is None:
            from .results import save_sampling_results as result_writer
        self.result_writer = result_writer
        if decode_policy not in ("all", "denoised"):
            raise ValueError("decode_policy must be 'all' or 'denoised'")
        self.vae_cache, self.decode_policy = bool(vae_cache), decode_policy
        if self.vae_cache:
            for p in pipelines:
                p.clear_vae_cache()

        if spa_labels is not None:
            self.spa_labels = [f"{int(i):02d}" for i in spa_labels]
        elif spa_label_range is not None:
            b, e, s = spa_label_range
            self.spa_labels = [f"{int(i):02d}" for i in range(b, e, s)]
        else:
            raise ValueError("spa_labels or spa_label_range must be provided")

        if tem_labels is not None:
            self.tem_labels = [f"{int(i):06d}" for i in tem_labels]
        elif tem_label_range is not None:
            b, e, s = tem_label_range
            self.tem_labels = [f"{int(i):06d}" for i in range(b, e, s)]
        else:
            raise ValueError("tem_labels or tem_label_range must be provided")

        self.input_spa_labels = [f"{int(i):02d}" for i in input_spa_labels]
        self.target_spa_labels = [label for label in self.spa_labels if label not in self.input_spa_labels]

        if self.window_size > len(self.target_spa_labels):
            raise ValueError(
                f"window_size(={self.window_size}) must be <= len(target_spa_labels)(={len(self.target_spa_labels)})"
            )
        if len(self.target_spa_labels) % self.sliding_stride != 0:
            raise ValueError(
                f"len(target_spa_labels)(={len(self.target_spa_labels)}) % sliding_stride(={self.sliding_stride}) must be 0"
            )
        if len(self.tem_labels) % self.sliding_stride != 0:
            raise ValueError(
                f"len(tem_labels)(={len(self.tem_labels)}) % sliding_stride(={self.sliding_stride}) must be 0"
            )
        if self.alternation_rounds > 1 and self.window_size > len(self.tem_labels):
            raise ValueError(
                f"window_size(={self.window_size}) must be <= the number of tem_labels(={len(self.tem_labels)}) "
                "when alternation_rounds > 1"
            )

        # spatio-temporal latent grid
        self.latents: Dict[str, Dict[str, Optional[torch.Tensor]]] = defaultdict(dict)
        self.timestep_indices: Dict[str, Dict[str, int]] = defaultdict(dict)
        for spa_label in self.spa_labels:
            for tem_label in self.tem_labels:
                self.latents[spa_label][tem_label] = None
                self.timestep_indices[spa_label][tem_label] = 0
        self.lock = Lock()
        self.prepare_tasks()

    # ------------------------------------------------------------------------------------------
    def prepare_tasks(self):
        domains = (["spatial", "temporal"] * self.alternation_rounds)[: self.alternation_rounds]
        self.all_tasks: List[List[dict]] = []
        for i, domain in enumerate(domains):
            domain_labels = self.tem_labels if domain == "spatial" else self.target_spa_labels
            self.all_tasks.append([{"alt": i + 1, "domain": domain, "domain_label": lb} for lb in domain_labels])

    def partition(self, round_index: int, rank: int, world: int) -> List[dict]:
        """Tasks of one alternation round owned by `rank`: round-robin, like threads draining one queue."""
        return self.all_tasks[round_index][rank::world]

    # ------------------------------------------------------------------------------------------
    def load_sample(self, alt: int, domain: str, domain_label: str) -> dict:
        def ref_indices(all_labels, ref_labels):
            return [all_labels.index(label) for label in ref_labels]

        if domain == "spatial":
            spa_labels = self.spa_labels
            tem_labels = [domain_label]
            input_indices = torch.tensor(ref_indices(self.spa_labels, self.input_spa_labels))
            target_indices = torch.tensor(ref_indices(self.spa_labels, self.target_spa_labels))
        elif domain == "temporal":
            spa_labels = [domain_label]
            tem_labels = self.tem_labels
            half = len(self.tem_labels)
            input_indices = torch.tensor(list(range(half)))  # first half: nearest input camera
            target_indices = torch.tensor(list(range(half, 2 * half)))  # second half: the target camera
        else:
            raise ValueError(f"unknown domain {domain!r}")

        sample = self.dataset.get_item(
            scene_label=self.dataset.scene_label,
            spa_labels=spa_labels,
            tem_labels=tem_labels,
            input_spa_labels=self.input_spa_labels,
        )
        sample["alt"] = alt
        sample["domain"] = domain
        sample["domain_label"] = domain_label
        sample["input_indices"] = input_indices
        sample["target_indices"] = target_indices

        cond_masks = sample["cond_masks"]
        cond_masks[...] = 1.0
        cond_masks[input_indices, ...] = 0.0
        sample["cond_masks"] = cond_masks

        with self.lock:
            latents, timestep_indices = [], []
            for _, spa_label, tem_label in sample["labels"]:
                latents.append(self.latents[spa_label][tem_label])
                timestep_indices.append(self.timestep_indices[spa_label][tem_label])
        timestep_indices = torch.tensor(timestep_indices)
        if timestep_indices[target_indices[0]] == 0:
            sample["latents"] = None
        else:
            dev = next(l.device for l in latents if l is not None)
            sample["latents"] = torch.stack([l.to(dev) for l in latents], dim=0)
        sample["timestep_indices"] = timestep_indices
        return sample

    @torch.no_grad()
    def denoise(self, sample: dict, pipe_idx: int = 0) -> dict:
        pipeline = self.pipelines[pipe_idx]
        task_label = f"alt{sample['alt']}_{'spa' if sample['domain'] == 'temporal' else 'tem'}{sample['domain_label']}"
        result = pipeline.sliding_iterative_denoise(
            pixel_values=sample["pixel_values"],
            plucker_embeds=sample["plucker_embeds"],
            skeletons=sample["skeletons"],
            cond_masks=sample["cond_masks"],
            latents=sample["latents"],
            domain=sample["domain"],
            timestep_indices=sample["timestep_indices"],
            window_size=self.window_size,
            sliding_stride=self.sliding_stride,
            sliding_shift=self.sliding_shift,
            bidirectional=self.bidirectional,
            num_denoising_steps=self.num_denoising_steps,
            alternation_rounds=self.alternation_rounds,
            guidance_scale=self.guidance_scale,
            tqdm=partial(_tqdm, desc=f"Denoising {task_label} on {pipeline.device}"),
            **self._pipeline_extensions(sample),
        )
        with self.lock:
            for label, latent, timestep_index in zip(sample["labels"], result["latents"], result["timestep_indices"]):
                _, spa_label, tem_label = label
                self.latents[spa_label][tem_label] = latent
                self.timestep_indices[spa_label][tem_label] = int(timestep_index)
        sample["images"] = result["images"].float().cpu()
        sample["timestep_indices"] = result["timestep_indices"].cpu()
        sample["fully_denoised"] = result["fully_denoised"].cpu()
        return sample

    def _pipeline_extensions(self, sample: dict) -> dict:
        """Keyword arguments beyond the reference protocol: VAE encoder-moment reuse keyed by (camera, frame) -- every
        image is otherwise re-encoded by each task that touches it, in every round (pipeline_diffuman4d.py:208-239) --
        and decoding only the rows that are saved (sampling_utils.py:103-104)."""
        kw = {}
        if self.vae_cache:
            kw["cache_keys"] = [(spa, tem) for _, spa, tem in sample["labels"]]
        if self.decode_policy != "all":
            kw["decode"] = self.decode_policy
        return kw

    def execute_one_task(self, task: dict, pipe_idx: int = 0) -> dict:
        sample = self.load_sample(**task)
        sample = self.denoise(sample, pipe_idx=pipe_idx)
        if self.result_writer is not None:
            self.result_writer(sample, output_dir=self.output_dir)
        return sample

    def execute_tasks(self):
        from .results import check_sampling_results
        for tasks in self.all_tasks:
            for task in tasks:
                self.execute_one_task(task)
        if self.result_writer is not None and not check_sampling_results(self.spa_labels, self.tem_labels, self.output_dir):
            raise ValueError("Sampling failed.")
