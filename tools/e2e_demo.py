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
from diffuman4d_amd.host.runner import DEFAULT_GPU_STREAMS, DEFAULT_TASK_BATCH, SamplingRunner  # noqa: E402
from diffuman4d_amd.host.weights import write_synthetic_checkpoint  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exp", default="demo_3d")
    ap.add_argument("--size", default="576x320", help="image HxW (latents are 1/8)")
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--writers", type=int, default=8)
    ap.add_argument("--gpu-streams", type=int, default=DEFAULT_GPU_STREAMS, help="stacks of tasks of a round in flight on the GPU (runner.gpu_streams)")
    ap.add_argument("--task-batch", type=int, default=DEFAULT_TASK_BATCH, help="tasks of a round per stack of shared window calls (runner.task_batch)")
    ap.add_argument("--fast-vae", action="store_true",
                    help="sampler.vae_cache=true sampler.decode_policy=denoised (encoder moments cached per grid cell, "
                         "decode only the rows that are saved)")
    ap.add_argument("--prune", action="store_true", help="sampler.prune_cond_rows=true (UNet tail only for consumed rows)")
    ap.add_argument("--device-results", action="store_true",
                    help="sampler.device_results=true: the result writer's arithmetic (mosaic, |out - in|, down-scale, uint8) on the GPU, "
                         "one uint8 package per task over PCIe")
    ap.add_argument("--writer-processes", type=int, default=0, help="runner.writer_processes: encode the packages in N processes")
    ap.add_argument("--host-threads", type=int, default=0, help="torch.set_num_threads for the host stages (0 = torch's default)")
    ap.add_argument("--timeline", default=None, help="write per-task stage intervals (load / denoise / save: start, end, thread) as JSON")
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
                         + (["sampler.prune_cond_rows=true"] if a.prune else [])
                         + (["sampler.device_results=true"] if a.device_results else []) + a.overrides)
    if a.host_threads > 0:
        torch.set_num_threads(a.host_threads)
    t0 = time.perf_counter()
    dataset = cfglib.instantiate(cfg["data"])
    pipelines = cfglib.instantiate(cfg["model"])
    sampler = cfglib.instantiate(cfg["sampler"], dataset=dataset, pipelines=pipelines)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0

    from collections import defaultdict
    acc, lock = defaultdict(float, {"load_sample": 0.0, "denoise": 0.0, "save": 0.0}), threading.Lock()
    events = []  # (stage, start, end, thread name): the timeline of the run

    def timed(name, fn):
        def w(*args, **kw):
            t = time.perf_counter()
            try:
                return fn(*args, **kw)
            finally:
                t1 = time.perf_counter()
                with lock:
                    acc[name] += t1 - t
                    events.append((name, t, t1, threading.current_thread().name))
        return w

    if a.timeline:  # finer host-side intervals inside `denoise` (no device synchronisation is added: these are the times the
        # worker THREAD spends in each part, including whatever blocks it)
        from diffuman4d_amd.host import results as _res
        for pipe in pipelines:
            pipe.prepare_all_latents = timed("d.prepare_all_latents", pipe.prepare_all_latents)
            pipe.denoise_latents = timed("d.denoise_latents", pipe.denoise_latents)
            pipe.vae.decode_to_images = timed("d.decode", pipe.vae.decode_to_images)
            pipe.vae.encode_scaled = timed("d.encode_scaled", pipe.vae.encode_scaled)
        # The window sweep and the decode are only ENQUEUED by the stages above (no host synchronisation on the hot path), so the
        # first host-side wait of a task -- the stream drain inside pack_results_on_device -- used to be booked as "d.pack"
        # (155 s of a 288 s run in round 3).  The timeline drains the stream explicitly first: "d.device_wait" is the time the
        # worker thread waits for the task's queued kernels, "d.pack" what the packaging itself costs.
        packed = timed("d.pack", _res.pack_results_on_device)

        def wait_then_pack(*args, **kw):
            timed("d.device_wait", lambda: torch.cuda.current_stream().synchronize())()
            return packed(*args, **kw)
        _res.pack_results_on_device = wait_then_pack
        sampler._scatter_cells = timed("d.scatter", sampler._scatter_cells)
    sampler.load_sample = timed("load_sample", sampler.load_sample)
    sampler.denoise = timed("denoise", sampler.denoise)
    sampler.denoise_stack = timed("denoise", sampler.denoise_stack)
    if sampler.result_writer is not None:
        sampler.result_writer = timed("save", sampler.result_writer)

    n_tasks = sum(len(t) for t in sampler.all_tasks)
    t0 = time.perf_counter()
    SamplingRunner(sampler, prefetch_depth=a.depth, writers=a.writers, gpu_streams=a.gpu_streams,
                   writer_processes=a.writer_processes, task_batch=a.task_batch).inference()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # GPU-stage occupancy: fraction of the wall time during which at least one / every denoise worker was inside `denoise`
    den = sorted((s0 - t0, e0 - t0) for n_, s0, e0, _ in events if n_ == "denoise")
    busy_any, cur_s, cur_e = 0.0, None, None
    for s0, e0 in den:
        if cur_e is None or s0 > cur_e:
            busy_any += (cur_e - cur_s) if cur_e is not None else 0.0
            cur_s, cur_e = s0, e0
        else:
            cur_e = max(cur_e, e0)
    busy_any += (cur_e - cur_s) if cur_e is not None else 0.0
    if a.timeline:
        Path(a.timeline).write_text(json.dumps([(n_, round(s0 - t0, 4), round(e0 - t0, 4), th) for n_, s0, e0, th in events]))
    n_lat = len(sampler.target_spa_labels) * len(sampler.tem_labels)
    done = sum(sampler.timestep_indices[c][f] > 0 for c in sampler.target_spa_labels for f in sampler.tem_labels)
    n_img = len(list(Path(sampler.output_dir).rglob("*.jpg")))
    print(json.dumps({
        "exp": a.exp, "image_size": [H, W], "prefetch_depth": a.depth, "writers": a.writers, "gpu_streams": a.gpu_streams, "task_batch": a.task_batch, "fast_vae": a.fast_vae, "prune_cond_rows": a.prune, "tasks": n_tasks,
        "target_latents": n_lat, "denoised": int(done), "images_written": n_img, "wall_s": round(wall, 3),
        "latents_per_s_end_to_end": round(n_lat / wall, 3),
        "stage_seconds": {k: round(v, 3) for k, v in acc.items()},
        "denoise_stage_busy_fraction": round(busy_any / wall, 4),
        "device_results": a.device_results, "writer_processes": a.writer_processes, "host_threads": torch.get_num_threads(),
        "checkpoint_write_s": round(t_ckpt, 2), "pipeline_load_s": round(t_load, 2),
    }), flush=True)
    if a.workdir is None:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
