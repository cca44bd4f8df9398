#!/usr/bin/env python
"""Golden vectors for a MULTI-ROUND job through the SAMPLER on the judged geometry: error growth across the
spatial -> temporal -> spatial alternation, measured instead of inferred from one round.

Reference control flow restated (read-only source: /root/reference):
  * src/samplers/sliding_iterative_sampler.py:49-212   grid of cells, task lists per alternation round, load_sample (latents = None in
                                                       round 1, torch.stack of the grid cells afterwards), denoise + write-back
  * src/samplers/sampling_runner.py:18-62              rounds in order, tasks of a round one after the other (one pipeline)
  * pipeline_diffuman4d.py:439-559                     the window sweep of every task
driven here by oracle/sampler.py (pinned to the reference's real classes: tests/test_reference_protocol.py) with the fp32 CPU oracle
pipeline (oracle/pipeline.py) at SD-2.1 width + SD VAE, 576 x 320 images = 72 x 40 latents.

Job: 8 cameras x 4 frames, input cameras [1, 5], window 4, stride 2, 3 alternation rounds (= 6 steps per latent):
  round 1: 4 spatial tasks  (8 views: 2 inputs + 6 targets; 3 window calls of F = 6)
  round 2: 6 temporal tasks (4 frames of the nearest input camera + 4 frames of the target camera; 2 window calls of F = 8)
  round 3: 4 spatial tasks  -- 36 UNet calls, 14 tasks, 224 VAE encodes.
The reference never seeds its draws; here call k of the job (tasks in the sampler's order) draws its posterior and
initial-latent noise from a CPU generator seeded NOISE_SEED + k (`SeededNoise`, used on both sides), rounded to bf16 so that every
precision sees the same numbers.

    python tests/golden/make_golden_multiround.py fp32     # ~25 min on 8 cores
    python tests/golden/make_golden_multiround.py bf16     # ~15 min: the oracle in bf16 = the reference's own arithmetic -> yardsticks

writes tests/golden/multiround_sd21_72x40.pt:
  latents [8, 4, 4, 72, 40] fp32   the final grid (camera, frame); timestep_indices [8, 4]
  images_u16                       decoded RGB of IMAGE_CELLS from the last round's tasks, 16-bit fixed point
  yard_latents / yard_images       rel-L2 of the bf16 oracle job against the fp32 one; checksums of weights and of the dataset's first task
The GPU test (tests/modelcheck.py::case_multiround_sd21) runs this repo's sampler + runner with the HIP pipeline on the same job.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parent))
OUT = Path(__file__).resolve().parent / "multiround_sd21_72x40.pt"

BF = torch.bfloat16
H, W = 576, 320
NOISE_SEED = 9000
KW = dict(spa_label_range=[0, 8, 1], tem_label_range=[0, 4, 1], input_spa_labels=[1, 5], window_size=4, sliding_stride=2,
          sliding_shift=0, bidirectional=False, num_denoising_steps=1, alternation_rounds=3, guidance_scale=2.0)
IMAGE_CELLS = [("00", "000000"), ("03", "000000"), ("06", "000003")]  # (camera, frame) cells whose decoded RGB is kept (targets)


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-12))


def dataset():
    from diffuman4d_amd.host.dataset import SyntheticSpaTemDataset
    return SyntheticSpaTemDataset(height=H, width=W, num_cameras=8)


class SeededNoise:
    """Pipeline wrapper: call k of the job gets its random draws from a CPU generator seeded NOISE_SEED + k (bf16-rounded).
    `oracle` selects the oracle pipeline's positional signature; otherwise the keywords go through as the sampler gave them."""

    def __init__(self, pipe, oracle: bool, keep_images_of=()):
        self.pipe, self.oracle, self.calls, self.device = pipe, oracle, 0, pipe.device
        self.keep_images_of, self.images = set(keep_images_of), {}  # call index -> the decoded images that call returned

    def __getattr__(self, name):  # everything else (prune_cond_rows, vae, ...) is the wrapped pipeline's
        return getattr(self.__dict__["pipe"], name)

    def _draws(self, n):
        g = torch.Generator().manual_seed(NOISE_SEED + self.calls)
        self.calls += 1
        return {k: torch.randn(n, 4, H // 8, W // 8, generator=g).to(BF) for k in ("pixel", "skeleton", "latents")}

    def sliding_iterative_denoise_stack(self, tasks, **kw):
        """The runner's task stacks (runner.task_batch; HIP pipeline only): every task of the stack gets the draws of ITS call number in
        the task-by-task job, so the stacked job is held to the same fixture."""
        assert not self.oracle
        first = self.calls
        outs = self.pipe.sliding_iterative_denoise_stack([dict(t, noise=self._draws(t["pixel_values"].shape[0])) for t in tasks], **kw)
        for k, out in enumerate(outs):
            if first + k in self.keep_images_of:
                self.images[first + k] = out["images"].float().cpu()
        return outs

    def sliding_iterative_denoise(self, **kw):
        noise = self._draws(kw["pixel_values"].shape[0])
        if not self.oracle:
            out = self.pipe.sliding_iterative_denoise(noise=noise, **kw)
            if self.calls - 1 in self.keep_images_of:
                self.images[self.calls - 1] = out["images"].float().cpu()
            return out
        kw.pop("tqdm", None)
        dt = self.pipe.dtype
        lat = kw.pop("latents")
        return self.pipe.sliding_iterative_denoise(kw.pop("pixel_values"), kw.pop("plucker_embeds"), kw.pop("skeletons"), kw.pop("cond_masks"),
                                                   None if lat is None else lat.to(dt), kw.pop("domain"), kw.pop("timestep_indices"),
                                                   {k: v.to(dt) for k, v in noise.items()}, **kw)


def run(dtype):
    from make_golden_demo3d import build_oracle
    from oracle.sampler import OracleRunner, OracleSampler
    op, usd, vsd = build_oracle(dtype)
    kept = {}

    def save(sample, output_dir):
        if sample["alt"] != KW["alternation_rounds"]:
            return
        for row, (_, c, f) in enumerate(sample["labels"]):
            if (c, f) in IMAGE_CELLS:
                kept[(c, f)] = sample["images"][row].clone()

    s = OracleSampler(dataset(), [SeededNoise(op, oracle=True)], "/tmp/unused", save=save, **KW)
    t0 = time.time()
    OracleRunner(s).inference()
    secs = time.time() - t0
    lat = torch.stack([torch.stack([s.latents[c][f].float() for f in s.tem_labels]) for c in s.spa_labels])
    idx = torch.tensor([[s.timestep_indices[c][f] for f in s.tem_labels] for c in s.spa_labels])
    images = torch.stack([kept[cell] for cell in IMAGE_CELLS]).float()
    first = dataset().get_item("synthetic", s.spa_labels, [s.tem_labels[0]], s.input_spa_labels)
    f = lambda t: float(t.float().abs().sum())  # noqa: E731
    chk = dict(pixel_values=f(first["pixel_values"]), plucker=f(first["plucker_embeds"]), skeletons=f(first["skeletons"]),
               unet_weights=float(sum(f(v) for v in usd.values())), vae_weights=float(sum(f(v) for v in vsd.values())))
    print(f"[{dtype}] job {secs:.0f}s, {s.pipelines[0].calls} tasks", flush=True)
    return lat, idx, images, chk, secs


def main():
    which = set(sys.argv[1:]) or {"fp32", "bf16"}
    blob = torch.load(OUT) if OUT.exists() else {}
    if "fp32" in which:
        lat, idx, images, chk, secs = run(torch.float32)
        blob.update(latents=lat, timestep_indices=idx, images_u16=(images * 65535.0).round().to(torch.int32).to(torch.uint16),
                    image_cells=IMAGE_CELLS, checksums=chk, kw=KW, noise_seed=NOISE_SEED, oracle_seconds_fp32=secs,
                    threads=torch.get_num_threads())
        torch.save(blob, OUT)
        print("fp32 pass stored", flush=True)
    if "bf16" in which:
        assert "latents" in blob, "run the fp32 pass first"
        lat, idx, images, chk, secs = run(BF)
        from make_golden_demo3d import same_checksums
        assert same_checksums(chk, blob["checksums"]) and torch.equal(idx, blob["timestep_indices"])
        ref_img = blob["images_u16"].to(torch.int32).float() / 65535.0
        tgt = blob["timestep_indices"] > 0
        blob.update(yard_latents=rel_l2(lat, blob["latents"]), yard_latents_targets=rel_l2(lat[tgt], blob["latents"][tgt]),
                    yard_images=rel_l2(images, ref_img), oracle_seconds_bf16=secs)
        torch.save(blob, OUT)
        print(f"bf16 pass stored: yardsticks latents {blob['yard_latents']:.3e} (targets {blob['yard_latents_targets']:.3e}) "
              f"images {blob['yard_images']:.3e}", flush=True)
    print("wrote", OUT, {k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in blob.items() if k != "checksums"})


if __name__ == "__main__":
    main()
