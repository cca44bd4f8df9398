#!/usr/bin/env python
"""End-to-end run of the CLI path on one MI355X with a synthetic checkpoint and the synthetic dataset:

    config compose -> load_pipelines (from_pretrained) -> SlidingIterativeSampler -> SamplingRunner.inference()
    (load_sample -> VAE encode -> window sweep -> VAE decode -> JPEG writer), i.e. everything `inference.py` does.

    python tools/e2e_demo.py --exp demo_3d                 # 48 cams x 1 frame, 1 task, 44 UNet calls
    python tools/e2e_demo.py --exp demo_4d_tiny --depth 1  # 48 x 16 grid, 16 + 44 + 16 tasks, 1056 UNet calls

Prints one JSON line: wall time, denoised latents/s END TO END (VAE, host data generation, H2D/D2H and image writes
included -- unlike bench.py, which times the resident window sweep only), and the time spent in the three task
stages.  --depth is the runner's prefetch depth (0 = the reference's serial load -> denoise -> save order).
"""
from __future__ import annotations

import argparse
import json
import shutil
import sys
import tempfile
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffuman4d_amd.host import config as cfglib  # noqa: E402
from diffuman4d_amd.host.runner import SamplingRunner  # noqa: E402
from diffuman4d_amd.host.weights import write_synthetic_checkpoint  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exp", default="demo_3d")
    ap.add_argument("--size", default="576x320", help="image HxW (latents are 1/8)")
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--writers", type=int, default=8)
    ap.add_argument("--gpu-streams", type=int, default=2, help="tasks of a round in flight on the GPU (runner.gpu_streams)")
    ap.add_argument("--fast-vae", action="store_true",
                    help="sampler.vae_cache=true sampler.decode_policy=denoised (encoder moments cached per grid cell, "
                         "decode only the rows that are saved)")
    ap.add_argument("--prune", action="store_true", help="sampler.prune_cond_rows=true (UNet tail only for consumed rows)")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args()
    H, W = (int(v) for v in a.size.lower().split("x"))
    work = Path(a.workdir or tempfile.mkdtemp(prefix="dm4d_e2e_"))
    t0 = time.perf_counter()
    ckpt = write_synthetic_checkpoint(work / "ckpt", device="cuda")
    t_ckpt = time.perf_counter() - t0
    cfg = cfglib.compose([f"exp={a.exp}", "model=diffuman4d_mi355x", "data=synthetic", f"model.model_dir={ckpt}",
                          "model.gpu_ids=[0]", f"data.height={H}", f"data.width={W}", f"result_dir={work / 'results'}"]
                         + (["sampler.vae_cache=true", "sampler.decode_policy=denoised"] if a.fast_vae else [])
                         + (["sampler.prune_cond_rows=true"] if a.prune else []) + a.overrides)
    t0 = time.perf_counter()
    dataset = cfglib.instantiate(cfg["data"])
    pipelines = cfglib.instantiate(cfg["model"])
    sampler = cfglib.instantiate(cfg["sampler"], dataset=dataset, pipelines=pipelines)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0

    acc, lock = {"load_sample": 0.0, "denoise": 0.0, "save": 0.0}, threading.Lock()

    def timed(name, fn):
        def w(*args, **kw):
            t = time.perf_counter()
            try:
                return fn(*args, **kw)
            finally:
                with lock:
                    acc[name] += time.perf_counter() - t
        return w

    sampler.load_sample = timed("load_sample", sampler.load_sample)
    sampler.denoise = timed("denoise", sampler.denoise)
    if sampler.result_writer is not None:
        sampler.result_writer = timed("save", sampler.result_writer)

    n_tasks = sum(len(t) for t in sampler.all_tasks)
    t0 = time.perf_counter()
    SamplingRunner(sampler, prefetch_depth=a.depth, writers=a.writers, gpu_streams=a.gpu_streams).inference()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    n_lat = len(sampler.target_spa_labels) * len(sampler.tem_labels)
    done = sum(sampler.timestep_indices[c][f] > 0 for c in sampler.target_spa_labels for f in sampler.tem_labels)
    n_img = len(list(Path(sampler.output_dir).rglob("*.jpg")))
    print(json.dumps({
        "exp": a.exp, "image_size": [H, W], "prefetch_depth": a.depth, "writers": a.writers, "gpu_streams": a.gpu_streams, "fast_vae": a.fast_vae, "prune_cond_rows": a.prune, "tasks": n_tasks,
        "target_latents": n_lat, "denoised": int(done), "images_written": n_img, "wall_s": round(wall, 3),
        "latents_per_s_end_to_end": round(n_lat / wall, 3),
        "stage_seconds": {k: round(v, 3) for k, v in acc.items()},
        "checkpoint_write_s": round(t_ckpt, 2), "pipeline_load_s": round(t_load, 2),
    }), flush=True)
    if a.workdir is None:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
