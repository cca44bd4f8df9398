"""SlidingIterativeSampler: the (camera x frame) latent grid and its task list.

Same contract as ``/root/reference/src/samplers/sliding_iterative_sampler.py``: the constructor keywords are the
Hydra sampler config (:16-35), bad window / stride / label arithmetic is a ``ValueError`` (:55,63,71-88), the grid
starts as ``latent = None, timestep_index = 0`` per cell (:91-96), ``load_sample`` (:102-153), ``denoise``
(:155-190), one task per frame in spatial rounds and per target camera in temporal rounds (:192-199).

What is organised differently here:
  * the sweep hyper-parameters live in one frozen ``SweepConfig`` that is splatted into the pipeline call;
  * grid cells keep the tensor the pipeline returned (device resident) instead of ``latent.cpu()``; gather / scatter of
    a task's cells are two small methods under one lock;
  * ``result_writer`` is injectable (the reference hard-wires ``save_sampling_results``);
  * ``partition(round, rank, world)`` exposes the per-round task sharding used by the one-process-per-GPU runner;
  * optional pipeline extensions (``vae_cache``, ``decode_policy``, ``prune_cond_rows``, ``plucker_on_device``,
    ``device_results``), off by default.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from functools import partial
from threading import Lock
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

try:  # tqdm is cosmetic
    from tqdm import tqdm as _tqdm
except Exception:  # pragma: no cover
    def _tqdm(it, **kw):
        return it

DOMAINS = ("spatial", "temporal")


@dataclass(frozen=True)
class SweepConfig:
    """Keyword arguments of ``pipeline.sliding_iterative_denoise`` that do not depend on the task."""
    window_size: int = 12
    sliding_stride: int = 1
    sliding_shift: int = 0
    bidirectional: bool = True
    num_denoising_steps: int = 1
    alternation_rounds: int = 3
    guidance_scale: float = 2.0


def _format_labels(explicit: Optional[Sequence[int]], label_range: Optional[Sequence[int]], width: int, name: str) -> List[str]:
    """Zero-padded string labels ("%02d" cameras, "%06d" frames) from an explicit list or a (begin, end, step) range."""
    if explicit is not None:
        values = [int(v) for v in explicit]
    elif label_range is not None:
        values = list(range(*[int(v) for v in label_range]))
    else:
        raise ValueError(f"{name}_labels or {name}_label_range must be provided")
    return [f"{v:0{width}d}" for v in values]


class SlidingIterativeSampler:
    def __init__(self, dataset, pipelines: list, output_dir: str = "./results/debug",
                 window_size: int = 12, sliding_stride: int = 1, sliding_shift: int = 0, bidirectional: bool = True,
                 num_denoising_steps: int = 1, alternation_rounds: int = 3, guidance_scale: float = 2.0,
                 spa_label_range: Optional[Sequence[int]] = (0, 48, 1), tem_label_range: Optional[Sequence[int]] = (0, 150, 1),
                 spa_labels: Optional[Sequence[int]] = None, tem_labels: Optional[Sequence[int]] = None,
                 input_spa_labels: Sequence[int] = (1, 13, 25, 37), result_writer: Optional[Callable] = None,
                 vae_cache: bool = False, decode_policy: str = "all", prune_cond_rows: bool = False,
                 plucker_on_device: bool = False, device_results: bool = False):
        self.dataset, self.pipelines, self.output_dir = dataset, pipelines, output_dir
        self.sweep = SweepConfig(window_size, sliding_stride, sliding_shift, bidirectional, num_denoising_steps,
                                 alternation_rounds, guidance_scale)
        # device_results: the arithmetic of the result writer (mosaic, |output - input|, down-scale, uint8 conversion) runs on the GPU
        # inside `denoise` and the task leaves it as a small uint8 package (results.pack_results_on_device); the writer then only
        # encodes files (imgwrite.write_package: in the runner's writer processes, or on the calling thread)
        # It applies only to the writer that understands packages: with a caller-supplied writer, or with pipelines that are not on a
        # HIP device, the sampler keeps the reference's contract (float CPU `images` handed to the writer).
        on_hip = bool(pipelines) and all(getattr(getattr(p, "device", None), "type", "cpu") == "cuda" for p in pipelines)
        self.device_results = bool(device_results) and result_writer is None and on_hip and torch.cuda.is_available()
        if result_writer is None:
            if self.device_results:
                from .results import write_packed_results as result_writer
            else:
                from .results import save_sampling_results as result_writer
        self.result_writer = result_writer
        if decode_policy not in ("all", "denoised"):
            raise ValueError("decode_policy must be 'all' or 'denoised'")
        self.vae_cache, self.decode_policy = bool(vae_cache), decode_policy
        self.plucker_on_device = bool(plucker_on_device)
        # set by DistributedSamplingRunner (modes "frame-shard" / "hybrid") around the tasks a GROUP of ranks runs together: the
        # parallel.FrameShard of that group, handed to the pipeline with a task-derived noise seed so that the group draws alike
        self.frame_shard = None
        self.noise_base_seed = 0
        if self.vae_cache:  # the cache is keyed by (camera, frame) of ONE scene
            for pipe in pipelines:
                pipe.clear_vae_cache()
        if prune_cond_rows:  # per-frame tail of the UNet only for rows whose noise prediction is consumed
            for pipe in pipelines:
                pipe.prune_cond_rows = True

        self.spa_labels = _format_labels(spa_labels, spa_label_range, 2, "spa")
        self.tem_labels = _format_labels(tem_labels, tem_label_range, 6, "tem")
        self.input_spa_labels = _format_labels(input_spa_labels, None, 2, "input_spa")
        inputs = set(self.input_spa_labels)
        self.target_spa_labels = [c for c in self.spa_labels if c not in inputs]
        self._validate()

        # the latent grid: every cell starts empty at timestep index 0
        self.latents: Dict[str, Dict[str, Optional[torch.Tensor]]] = {c: dict.fromkeys(self.tem_labels) for c in self.spa_labels}
        self.timestep_indices: Dict[str, Dict[str, int]] = {c: dict.fromkeys(self.tem_labels, 0) for c in self.spa_labels}
        self.lock = Lock()
        self.prepare_tasks()

    # the reference exposes the sweep arguments as attributes; keep them readable under the same names
    def __getattr__(self, name):
        sweep = self.__dict__.get("sweep")
        if sweep is not None and name in SweepConfig.__dataclass_fields__:
            return getattr(sweep, name)
        raise AttributeError(name)

    def _validate(self):
        """Same conditions and messages as sliding_iterative_sampler.py:71-88."""
        n_tgt, n_tem, sw = len(self.target_spa_labels), len(self.tem_labels), self.sweep
        if sw.window_size > n_tgt:
            raise ValueError(f"window_size(={sw.window_size}) must be <= len(target_spa_labels)(={n_tgt})")
        if n_tgt % sw.sliding_stride != 0:
            raise ValueError(f"len(target_spa_labels)(={n_tgt}) % sliding_stride(={sw.sliding_stride}) must be 0")
        if n_tem % sw.sliding_stride != 0:
            raise ValueError(f"len(tem_labels)(={n_tem}) % sliding_stride(={sw.sliding_stride}) must be 0")
        if sw.alternation_rounds > 1 and sw.window_size > n_tem:
            raise ValueError(f"window_size(={sw.window_size}) must be <= the number of tem_labels(={n_tem}) "
                             "when alternation_rounds > 1")

    # ------------------------------------------------------------------------------------------
    def prepare_tasks(self):
        """Round r is spatial for even r (one task per frame) and temporal for odd r (one task per target camera)."""
        self.all_tasks: List[List[dict]] = []
        for r in range(self.sweep.alternation_rounds):
            domain = DOMAINS[r % 2]
            units = self.tem_labels if domain == "spatial" else self.target_spa_labels
            self.all_tasks.append([dict(alt=r + 1, domain=domain, domain_label=u) for u in units])

    def partition(self, round_index: int, rank: int, world: int) -> List[dict]:
        """Tasks of one alternation round owned by `rank`: round-robin, like threads draining one queue."""
        return self.all_tasks[round_index][rank::world]

    # ------------------------------------------------------------------------------------------
    def _task_geometry(self, domain: str, domain_label: str) -> Tuple[List[str], List[str], List[int], List[int]]:
        """(cameras, frames, input rows, target rows) of the sample a task denoises."""
        if domain == "spatial":  # every camera of one frame; rows are cameras
            row_of = {c: i for i, c in enumerate(self.spa_labels)}
            return (self.spa_labels, [domain_label], [row_of[c] for c in self.input_spa_labels],
                    [row_of[c] for c in self.target_spa_labels])
        if domain == "temporal":  # all frames of the nearest input camera, then all frames of the target camera
            t = len(self.tem_labels)
            return [domain_label], self.tem_labels, list(range(t)), list(range(t, 2 * t))
        raise ValueError(f"unknown domain {domain!r}")

    def _gather_cells(self, labels) -> Tuple[List[Optional[torch.Tensor]], torch.Tensor]:
        with self.lock:
            cells = [(self.latents[c][f], self.timestep_indices[c][f]) for _, c, f in labels]
        return [lat for lat, _ in cells], torch.tensor([idx for _, idx in cells])

    def _scatter_cells(self, labels, latents, indices) -> None:
        with self.lock:
            for (_, c, f), lat, idx in zip(labels, latents, indices):
                self.latents[c][f] = lat
                self.timestep_indices[c][f] = int(idx)

    def load_sample(self, alt: int, domain: str, domain_label: str) -> dict:
        cams, frames, input_rows, target_rows = self._task_geometry(domain, domain_label)
        sample = self.dataset.get_item(scene_label=self.dataset.scene_label, spa_labels=cams, tem_labels=frames,
                                       input_spa_labels=self.input_spa_labels)
        sample.update(alt=alt, domain=domain, domain_label=domain_label, input_indices=torch.tensor(input_rows),
                      target_indices=torch.tensor(target_rows))
        mask = sample["cond_masks"]  # 0 = conditioning (input) row, 1 = row to denoise
        mask.fill_(1.0)
        mask[sample["input_indices"]] = 0.0

        cell_latents, cell_indices = self._gather_cells(sample["labels"])
        first_round = int(cell_indices[target_rows[0]]) == 0  # targets of a task always share one index
        if first_round:
            sample["latents"] = None  # the pipeline draws the initial noise
        else:
            dev = next(lat.device for lat in cell_latents if lat is not None)
            sample["latents"] = torch.stack([lat.to(dev) for lat in cell_latents])
            if dev.type == "cuda":
                # this may run on a loader thread (runner.run_round_pipelined): the gather above is queued on THIS thread's
                # current stream, the consumer is the denoise worker's own stream -> hand an event over with the sample
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                sample["_latents_ready"] = ev
        sample["timestep_indices"] = cell_indices
        return sample

    def _wait_for_cells(self, sample: dict, pipe) -> bool:
        on_gpu = torch.cuda.is_available() and getattr(pipe.device, "type", "cpu") == "cuda"
        ready = sample.pop("_latents_ready", None)
        if on_gpu and ready is not None:  # the grid cells were gathered on another stream (load_sample)
            cur = torch.cuda.current_stream(pipe.device)
            cur.wait_event(ready)
            sample["latents"].record_stream(cur)  # allocated in the loader stream's pool, consumed here
        return on_gpu

    def _task_tensors(self, sample: dict) -> dict:
        tensors = {k: sample[k] for k in ("pixel_values", "plucker_embeds", "skeletons", "cond_masks", "latents",
                                          "timestep_indices")}
        if self.plucker_on_device:
            tensors["plucker_embeds"] = None
        return tensors

    @staticmethod
    def _bar(sample: dict, pipe, more: int = 0):
        axis = "spa" if sample["domain"] == "temporal" else "tem"  # the label names the FIXED axis of the task
        also = f" (+{more})" if more else ""
        return partial(_tqdm, desc=f"Denoising alt{sample['alt']}_{axis}{sample['domain_label']}{also} on {pipe.device}")

    @torch.no_grad()
    def denoise(self, sample: dict, pipe_idx: int = 0) -> dict:
        pipe = self.pipelines[pipe_idx]
        on_gpu = self._wait_for_cells(sample, pipe)
        result = pipe.sliding_iterative_denoise(domain=sample["domain"], tqdm=self._bar(sample, pipe), **self._task_tensors(sample),
                                                **asdict(self.sweep), **self._pipeline_extensions(sample))
        return self._take_result(sample, result, pipe, on_gpu)

    def stackable(self, samples: List[dict], pipe_idx: int = 0) -> bool:
        """Tasks that can share their window calls (pipeline.sliding_iterative_denoise_stack): one domain, the same rows conditioned,
        the same timestep indices, the same tensor shapes -- what the tasks of one alternation round have -- and no frame sharding."""
        if len(samples) < 2 or self.frame_shard is not None or getattr(self, "shard_follower", False):
            return False
        if not hasattr(self.pipelines[pipe_idx], "sliding_iterative_denoise_stack"):
            return False
        a = samples[0]
        for b in samples[1:]:
            if b["domain"] != a["domain"] or b["pixel_values"].shape != a["pixel_values"].shape \
                    or (a["latents"] is None) != (b["latents"] is None) \
                    or not torch.equal(b["cond_masks"][:, 0, 0, 0] == 0.0, a["cond_masks"][:, 0, 0, 0] == 0.0) \
                    or not torch.equal(torch.as_tensor(b["timestep_indices"]), torch.as_tensor(a["timestep_indices"])):
                return False
        return True

    @torch.no_grad()
    def denoise_stack(self, samples: List[dict], pipe_idx: int = 0) -> List[dict]:
        """Extension (runner.task_batch): the samples' tasks through shared window calls; every sample ends up exactly as `denoise`
        leaves it (results are bitwise those of one `denoise` per sample in list order).  Samples that are not `stackable` run
        one by one."""
        if not self.stackable(samples, pipe_idx):
            return [self.denoise(s, pipe_idx=pipe_idx) for s in samples]
        pipe = self.pipelines[pipe_idx]
        on_gpu = [self._wait_for_cells(s, pipe) for s in samples][0]
        ext = [self._pipeline_extensions(s) for s in samples]
        decode = ext[0].pop("decode", "all")
        for e in ext[1:]:
            e.pop("decode", None)
        tasks = [dict(self._task_tensors(s), **e) for s, e in zip(samples, ext)]
        results = pipe.sliding_iterative_denoise_stack(tasks, domain=samples[0]["domain"], tqdm=self._bar(samples[0], pipe, len(samples) - 1),
                                                       decode=decode, **asdict(self.sweep))
        return [self._take_result(s, r, pipe, on_gpu) for s, r in zip(samples, results)]

    def _take_result(self, sample: dict, result: dict, pipe, on_gpu: bool) -> dict:
        sample["timestep_indices"] = result["timestep_indices"].cpu()
        sample["fully_denoised"] = result["fully_denoised"].cpu()
        if self.device_results and on_gpu and self.result_writer is not None:  # (device_results is off for caller-supplied writers)
            from .results import pack_results_on_device
            sample["_package"] = pack_results_on_device(sample, result["images"], output_dir=self.output_dir, device=pipe.device)
            sample["images"] = None  # the float images never leave the device (the package holds what gets written)
        elif getattr(self, "shard_follower", False):
            sample["images"] = None  # nothing was decoded on this rank and nothing will be written
        else:
            sample["images"] = result["images"].float().cpu()
        # hand the cells over only when they are complete: with several task streams per GPU (runner gpu_streams) another
        # task's thread, another stream or the round-boundary exchange may read the grid as soon as the cells are in it.
        # The copies above are not a reliable barrier (host-resident results do not synchronise), so drain explicitly.
        if on_gpu:
            torch.cuda.current_stream(pipe.device).synchronize()
        self._scatter_cells(sample["labels"], result["latents"], result["timestep_indices"])
        return sample

    def _pipeline_extensions(self, sample: dict) -> dict:
        """Keyword arguments beyond the reference protocol: VAE encoder-moment reuse keyed by (camera, frame) -- every
        image is otherwise re-encoded by each task that touches it, in every round (pipeline_diffuman4d.py:208-239) --
        and decoding only the rows that are saved (sampling_utils.py:103-104)."""
        kw = {}
        if self.plucker_on_device:  # cameras instead of full-resolution ray maps
            if sample.get("Ks") is None or sample.get("poses") is None:
                raise ValueError("plucker_on_device needs the dataset's 'Ks' and 'poses' (spatem_dataset.py:178-189)")
            kw["cameras"] = {"Ks": sample["Ks"], "poses": sample["poses"], "image_size": tuple(sample["pixel_values"].shape[-2:])}
        if self.vae_cache:
            kw["cache_keys"] = [(spa, tem) for _, spa, tem in sample["labels"]]
        if self.decode_policy != "all":
            kw["decode"] = self.decode_policy
        if getattr(self, "shard_follower", False):
            kw["decode"] = "none"  # a non-leading rank of a frame-shard group: nobody reads its images (the leader writes the results)
        if self.frame_shard is not None:
            kw["shard"] = self.frame_shard
            kw["noise_seed"] = self.task_noise_seed(sample["alt"], sample["domain"], sample["domain_label"])
        return kw

    def task_noise_seed(self, alt: int, domain: str, domain_label: str) -> int:
        """A seed every rank derives alike for one task (a pure function of the task's identity and `noise_base_seed`).  A hash of the
        tuple, not a sum of scaled fields: frame labels run to 999999 (`%06d`), so weighted sums made (round 1, frame 100003 + x) collide
        with (round 2, frame x) and spatial frame 50021 + c with temporal camera c -- tasks that would then draw identical noise."""
        import hashlib
        key = f"{int(self.noise_base_seed)}/{int(alt)}/{domain}/{domain_label}".encode()
        # 63 bits of the digest (torch.Generator.manual_seed takes 64-bit seeds): a 48 x 150 job has ~7 000 tasks, for which 31 bits
        # would leave a ~1 % chance of two tasks sharing a seed
        return int.from_bytes(hashlib.sha256(key).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF

    def execute_one_task(self, task: dict, pipe_idx: int = 0) -> dict:
        sample = self.denoise(self.load_sample(**task), pipe_idx=pipe_idx)
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
