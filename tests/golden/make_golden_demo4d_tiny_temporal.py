#!/usr/bin/env python
"""Golden vectors for ONE FULL TEMPORAL TASK on the judged geometry: BASELINE.json configs[1] (`demo_4d_tiny`), middle round.

Reference configuration restated (read-only source: /root/reference):
  * configs/exp/demo_4d_tiny.yaml:6-9        48 cameras x 16 frames, input cameras [1, 13, 25, 37], sampler `sliding_fast`
  * configs/sampler/sliding_fast.yaml:6      sliding_stride 2 on top of sliding_default: window 12, shift 0, not bidirectional,
                                             1 denoising step per window, 3 alternation rounds, guidance 2.0
  * sliding_iterative_sampler.py:112-118     a temporal task = one target camera: N = 2 T = 32 rows, rows 0..15 the nearest input
                                             camera's frames (cond mask 0), rows 16..31 the target camera's frames
  * pipeline_diffuman4d.py:468-472,504-518   6 steps per alternation, 18 inference steps; 16 / 2 = 8 windows, each window = 12 target
                                             frames + the 12 input frames at the same times => 8 UNet calls of F = 24 (CFG batch 48)
  * sliding_iterative_sampler.py:192-199     the temporal round is round 2 of 3: the targets enter at timestep index 6 with the
                                             latents the first spatial round left in the grid, and leave at index 12

Geometry: SD-2.1 UNet (320, 640, 1280, 1280), SD VAE (128, 256, 512, 512), 576 x 320 images = 72 x 40 latents.

    python tests/golden/make_golden_demo4d_tiny_temporal.py fp32   # ~25 min on 8 cores: fp32 oracle latents + decoded RGB of 4 rows
    python tests/golden/make_golden_demo4d_tiny_temporal.py bf16   # ~35 min: the oracle in bf16 (the reference's own arithmetic) -> yardsticks

writes tests/golden/demo4dtiny_temporal_sd21_72x40.pt:
  latents            fp32 oracle result, all 32 rows [32, 4, 72, 40]
  images_u16         decoded RGB of IMAGE_ROWS (4 target frames), 16-bit fixed point of [0, 1]
  timestep_indices, fully_denoised      bit-exact bookkeeping
  yard_latents, yard_images             rel-L2 of the bf16 oracle against the fp32 oracle
  checksums of weights / inputs / noise (the GPU test rebuilds them from the seeds and verifies first)

Weights are NOT stored: both sides rebuild them with ``random_state_dict(shapes, seed, device="cpu")``.
The GPU test (tests/modelcheck.py::case_demo4dtiny_temporal) never runs the oracle at this size on the GPU box.
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
OUT = Path(__file__).resolve().parent / "demo4dtiny_temporal_sd21_72x40.pt"

BF = torch.bfloat16
LAT_H, LAT_W = 72, 40
T = 16                                   # configs/exp/demo_4d_tiny.yaml:8
N = 2 * T                                # sliding_iterative_sampler.py:112-118
IMAGE_ROWS = [16, 21, 26, 31]            # decoded rows kept in the fixture: four target frames
START_INDEX = 6                          # timestep index of the targets when round 2 starts (12 * 1 / 2 steps per round)
UNET_SEED, VAE_SEED, TASK_SEED, NOISE_SEED, LATENT_SEED = 0, 1, 2234, 5321, 6321
KW = dict(window_size=12, sliding_stride=2, sliding_shift=0, bidirectional=False, num_denoising_steps=1,
          alternation_rounds=3, guidance_scale=2.0)


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-12))


def task_inputs():
    """32 frames with the value ranges of spatem_dataset.py:191-228 (see make_golden_demo3d.task_inputs): rows 0..15 the input
    camera, rows 16..31 the target camera, cond mask 0 on the first half (sliding_iterative_sampler.py:112-118, 134-139)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(TASK_SEED)
    n, H, W = N, LAT_H * 8, LAT_W * 8
    base = F.interpolate(torch.randn(n, 3, H // 16, W // 16, generator=g), size=(H, W), mode="bilinear")
    pv = (0.6 * base + 0.15 * torch.randn(n, 3, H, W, generator=g)).clamp(-1, 1)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    inside = ((xs / 0.7) ** 2 + (ys / 0.9) ** 2) < 1.0
    pv = torch.where(inside, pv, torch.ones_like(pv))
    sk = -torch.ones(n, 3, H, W)
    for i in range(n):
        for _ in range(6):
            y0, x0 = int(torch.randint(40, H - 120, (1,), generator=g)), int(torch.randint(20, W - 60, (1,), generator=g))
            hh, ww = int(torch.randint(8, 100, (1,), generator=g)), int(torch.randint(4, 40, (1,), generator=g))
            sk[i, :, y0:y0 + hh, x0:x0 + ww] = (torch.rand(3, 1, 1, generator=g) * 2 - 1)
    # two cameras: the Pluecker map is constant over a camera's frames
    pl2 = F.interpolate(torch.randn(2, 6, H // 32, W // 32, generator=g) * 0.6, size=(H, W), mode="bilinear").clamp(-1, 1)
    pl = torch.cat([pl2[0:1].expand(T, -1, -1, -1), pl2[1:2].expand(T, -1, -1, -1)]).contiguous()
    cm = torch.ones(n, 1, H, W)
    cm[:T] = 0.0
    return pv, pl, sk, cm


def task_noise():
    g = torch.Generator().manual_seed(NOISE_SEED)
    return {k: torch.randn(N, 4, LAT_H, LAT_W, generator=g).to(BF) for k in ("pixel", "skeleton", "latents")}


def grid_latents():
    """What the sampler hands over in round 2 (sliding_iterative_sampler.py:142-151): the grid's latents of the 32 cells, in the model
    dtype.  Partially denoised latents have unit marginal variance; the input rows are overwritten by their encoded images inside the
    call (pipeline_diffuman4d.py:379)."""
    g = torch.Generator().manual_seed(LATENT_SEED)
    return torch.randn(N, 4, LAT_H, LAT_W, generator=g).to(BF)


def start_indices():
    idx = torch.zeros(N, dtype=torch.int64)
    idx[T:] = START_INDEX
    return idx


def checksums(pv, pl, sk, cm, noise, lat, usd, vsd):
    f = lambda t: float(t.float().abs().sum())  # noqa: E731
    return dict(pixel_values=f(pv), plucker=f(pl), skeletons=f(sk), cond_masks=f(cm), latents_in=f(lat),
                noise={k: f(v) for k, v in noise.items()},
                unet_weights=float(sum(f(v) for v in usd.values())), vae_weights=float(sum(f(v) for v in vsd.values())))


def run(dtype):
    import make_golden_demo3d as mk3
    op, usd, vsd = mk3.build_oracle(dtype)
    pv, pl, sk, cm = task_inputs()
    noise = task_noise()
    lat = grid_latents()
    nz = noise if dtype == BF else {k: v.float() for k, v in noise.items()}
    t0 = time.time()
    out = op.sliding_iterative_denoise(pv, pl, sk, cm, lat.to(dtype), "temporal", start_indices(), nz, decode=False, trace=None, **KW)
    t_den = time.time() - t0
    images = op.post_process(out["latents"][IMAGE_ROWS])
    print(f"[{dtype}] denoise {t_den:.0f}s, decode of {len(IMAGE_ROWS)} rows {time.time() - t0 - t_den:.0f}s", flush=True)
    return out, images, checksums(pv, pl, sk, cm, noise, lat, usd, vsd), t_den


def main():
    import make_golden_demo3d as mk3
    which = set(sys.argv[1:]) or {"fp32", "bf16"}
    blob = torch.load(OUT) if OUT.exists() else {}
    if "fp32" in which:
        out, images, chk, secs = run(torch.float32)
        blob.update(latents=out["latents"].float(), images_u16=(images.float() * 65535.0).round().to(torch.int32).to(torch.uint16),
                    image_rows=IMAGE_ROWS, timestep_indices=out["timestep_indices"], fully_denoised=out["fully_denoised"],
                    checksums=chk, kw=KW, oracle_seconds_fp32=secs, threads=torch.get_num_threads(), start_index=START_INDEX,
                    seeds=dict(unet=UNET_SEED, vae=VAE_SEED, task=TASK_SEED, noise=NOISE_SEED, latents=LATENT_SEED))
        torch.save(blob, OUT)
        print("fp32 pass stored", flush=True)
    if "bf16" in which:
        assert "latents" in blob, "run the fp32 pass first"
        out, images, chk, secs = run(BF)
        assert mk3.same_checksums(chk, blob["checksums"])
        assert torch.equal(out["timestep_indices"], blob["timestep_indices"])
        ref_img = blob["images_u16"].to(torch.int32).float() / 65535.0
        blob.update(yard_latents=rel_l2(out["latents"], blob["latents"]), yard_images=rel_l2(images, ref_img),
                    yard_latents_targets=rel_l2(out["latents"][T:], blob["latents"][T:]), oracle_seconds_bf16=secs)
        torch.save(blob, OUT)
        print(f"bf16 pass stored: yardsticks latents {blob['yard_latents']:.3e} images {blob['yard_images']:.3e}", flush=True)
    print("wrote", OUT, {k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in blob.items() if k != "checksums"})


if __name__ == "__main__":
    main()
