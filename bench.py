#!/usr/bin/env python
"""bench.py -- sliding iterative denoiser throughput on MI355X.

Metric (BASELINE.json): denoised view-frame latents / second on the 44-target-camera x 150-frame grid
(`demo_4d`, `sliding_fast`: window 12, stride 2, 3 alternation rounds => 18 denoising steps per
latent, CFG 2.0), synthetic 72x40x4 latents, SD-2.1-geometry UNet with seeded random weights.

One "step" = one schedule unit of that run: 2 spatial window calls (F = 4 inputs + 12 targets = 16
frames, CFG batch 32) + 1 temporal window call (F = 12 + 12 = 24 frames, CFG batch 48) -- exactly
the 6600 : 3300 call mix of the full run.  The K timed steps are dealt to --task-streams (default 2)
independent tasks that are in flight at the same time, one HIP stream and worker thread each, as the runner does with the
tasks of a round (host/runner.py: gpu_streams).  Each call = model-input pack -> UNet ->
CFG + batched DDIM update on device-resident latents.  One unit advances 36 latent-steps = 2 fully
denoised latents; the full run is 3300 units.  VAE encode/decode is outside the timed region
 -- inputs are resident in HBM when timing starts.

Multi-GPU (one process per GPU, torchrun): tasks of a round are independent, so ranks run their
own units with no data-path collective (the only exchange of the real run, the grid transpose at
the 2 round boundaries, moves < 0.2 GB and is not part of a unit): weak scaling.

Prints ONE JSON line on rank 0: the contract fields (metric, value = whole-job denoised latents/s, ...), `roofline`
(attention kernel: HIP-event timing in a one-task-at-a-time pass of the same steps, PMC traffic from profiles/), `cpu_baseline` (N = 1 only:
the CPU oracle on one full spatial window), `secondary` (latent-steps/s, UNet calls/s, sustained TFLOP/s) and
`kernel_breakdown_one_step` (per kernel family, from one extra untimed instrumented step).
Options beyond the contract: --latent HxW, --mode frame-shard, --prune-cond-rows (opt-in extension).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

LAT_H, LAT_W = 72, 40
N_CAMS, N_FRAMES = 48, 150
INPUT_CAMS = [1, 13, 25, 37]
WINDOW, STRIDE, ROUNDS, GUIDANCE = 12, 2, 3, 2.0
STEPS_PER_LATENT = WINDOW // STRIDE * ROUNDS  # 18
LATENTS_PER_UNIT = 3 * WINDOW / STEPS_PER_LATENT  # 2.0
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["task", "frame-shard"], default="task",
                    help="multi-GPU decomposition: independent tasks per rank (default, weak scaling) or every window "
                         "frame-sharded over all ranks with RCCL K/V all-gathers (strong scaling, BASELINE config 4)")
    ap.add_argument("--latent", default="72x40",
                    help="latent grid HxW: 72x40 = BASELINE.json's synthetic grid (default, the judged line); 128x128 = the "
                         "1024^2 images the reference's demo configs run (SURVEY.md 8d asks for both)")
    ap.add_argument("--task-streams", type=int, default=2,
                    help="independent tasks in flight per GPU, each on its own HIP stream and worker thread (the runner's "
                         "gpu_streams): the K timed steps are dealt round-robin to the streams. 1 = one task at a time")
    ap.add_argument("--prune-cond-rows", action="store_true",
                    help="opt-in extension, NOT the judged configuration: skip the per-frame tail of the UNet (after the last "
                         "3-D attention) for conditioning frames, whose noise prediction the reference discards")
    ap.add_argument("--cpu-frames", type=int, default=16,
                    help="frames of the CPU-baseline UNet call (16 = a full spatial window); the default is the full window,"
                         " about 20 s with 32 threads")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="torch threads of the CPU baseline: on the 256-thread GPU hosts 32 threads are 3x faster than 64 "
                         "and 50x faster than 256 on this model (tools/dev/cpu_baseline_probe.py)")
    return ap.parse_args()


def build_tasks(pipe, dev, shard=None):
    """Device-resident synthetic task tensors + window plans for one spatial and one temporal task."""
    from diffuman4d_amd.host.schedule import plan_sweep
    g = torch.Generator(device=dev).manual_seed(1234)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=dev) * scale).to(torch.bfloat16)

    tasks = {}
    for domain, n, cond in (("spatial", N_CAMS, [i in INPUT_CAMS for i in range(N_CAMS)]),
                            ("temporal", 2 * N_FRAMES, [i < N_FRAMES for i in range(2 * N_FRAMES)])):
        plan = plan_sweep(cond, [0] * n, domain, WINDOW, STRIDE, 0, False, 1, ROUNDS)
        mask = torch.tensor([0.0 if c else 1.0 for c in cond], device=dev).to(torch.bfloat16)
        hw = LAT_H * LAT_W
        tasks[domain] = dict(
            pv=rnd(n, hw, 4, scale=0.18215 * 4), pl=rnd(n, hw, 6, scale=0.5).clamp(-1, 1),
            sk=rnd(n, hw, 4, scale=0.18215 * 4), lat=rnd(n, hw, 4),
            cm=mask[:, None, None].expand(n, hw, 1).contiguous(), plan=plan,
            tables=pipe.upload_plan(plan, GUIDANCE, shard), domain=domain)
    return tasks


def run_call(pipe, task, call_idx, shard=None):
    """One window call: pack -> UNet -> CFG + DDIM (Diffuman4DPipeline.window_call, the body of denoise_latents)."""
    tb = task["tables"]
    pipe.window_call(task["lat"], task["pv"], task["pl"], task["sk"], task["cm"], tb, call_idx % tb["calls"], LAT_H, LAT_W,
                     [task["domain"]] * 2, GUIDANCE, True, False, shard)


def run_unit(pipe, tasks, u, shard=None):
    run_call(pipe, tasks["spatial"], 2 * u, shard)
    run_call(pipe, tasks["spatial"], 2 * u + 1, shard)
    run_call(pipe, tasks["temporal"], u, shard)


def cpu_baseline(frames: int, threads: int):
    """Time the CPU oracle (oracle/: plain PyTorch restatement of the reference) on one spatial window UNet call."""
    from oracle.unet import UNetConfig, UNetMultiviewConditionModel
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    cfg = UNetConfig()
    with torch.no_grad():
        m = UNetMultiviewConditionModel(cfg).eval()
        for p in m.parameters():
            p.mul_(0.5)  # default torch init, damped: values do not matter for timing
        B = 2 * frames
        x = torch.randn(B, cfg.in_channels, LAT_H, LAT_W)
        t = torch.randint(0, 1000, (B,))
        t0 = time.time()
        m(x, t, domains=["spatial"] * 2, num_frames=frames)
        dt = time.time() - t0
    # a window of F frames carries the same 4:12 input:target ratio as the real spatial window
    targets = frames * (WINDOW / (WINDOW + len(INPUT_CAMS)))
    return {
        "value": round(targets / STEPS_PER_LATENT / dt, 5), "unit": "latents/s", "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"one spatial-window UNet forward of the CPU oracle (fp32, F={frames} of 16 frames, CFG batch {B}, "
                  f"{LAT_H}x{LAT_W} latents) = {dt:.1f} s; {targets:.1f} latent-steps / {STEPS_PER_LATENT} steps per latent"
                  + ("" if frames >= 16 else " (the 3-D attention share grows with F^2: the full F=16 window is slower per latent)"),
        "seconds": round(dt, 2),
    }


def main():
    global LAT_H, LAT_W
    args = parse()
    LAT_H, LAT_W = (int(v) for v in args.latent.lower().split("x"))
    if LAT_H % 8 or LAT_W % 8:
        raise SystemExit("--latent: both sides must be multiples of 8 (three UNet down-samplings)")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # DM4D_BENCH_SHARED_GPU=1 (testing only): all ranks share device 0 and rendezvous over gloo, so that the N > 1 code
    # paths can be exercised on a 1-GPU box; the numbers of such a run mean nothing
    shared = os.environ.get("DM4D_BENCH_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        # RCCL prints a version banner on stdout when the communicator comes up; this program's stdout carries exactly
        # one JSON line, so fd 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if shared:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from diffuman4d_amd.host import ops
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMScheduler
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    from diffuman4d_amd.host.weights import random_state_dict, unet_param_shapes

    cfg = UNetConfig()
    unet = UNetMultiviewConditionModel(cfg, random_state_dict(unet_param_shapes(cfg), 0, dev), dev)
    pipe = Diffuman4DPipeline(None, unet, DDIMScheduler(), dev)
    pipe.prune_cond_rows = bool(args.prune_cond_rows)
    shard = None
    if args.mode == "frame-shard" and world > 1:
        from diffuman4d_amd.host.parallel import FrameShard
        shard = FrameShard()
        if 16 % world or 24 % world:
            raise SystemExit("frame-shard mode needs a rank count dividing both window sizes (16 and 24): 1, 2, 4 or 8")
    # Tasks of a round are independent, so the runner keeps `gpu_streams` of them in flight per GPU (host/runner.py);
    # here: S task states, S worker threads, one HIP stream each, one set of weights.  Collectives of the frame-shard
    # mode must be issued in one order on every rank, so that mode runs one task at a time.
    S = 1 if shard is not None else max(1, min(args.task_streams, max(1, args.steps)))
    task_sets = [build_tasks(pipe, dev, shard) for _ in range(S)]
    tasks = task_sets[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [None]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_units(first, count):
        """`count` units starting at per-task unit index `first`, dealt round-robin to the S task streams."""
        def work(si):
            torch.cuda.set_device(dev)
            with torch.no_grad(), torch.cuda.stream(streams[si]):
                for j in range(si, count, S):
                    run_unit(pipe, task_sets[si], first + j // S, shard)
        if S == 1:
            with torch.no_grad():
                for j in range(count):
                    run_unit(pipe, tasks, first + j, shard)
            return
        import threading
        errs = []

        def guarded(si):
            try:
                work(si)
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=guarded, args=(si,)) for si in range(S)]
        [t.start() for t in th]
        [t.join() for t in th]
        if errs:
            raise errs[0]

    run_units(0, max(args.warmup, 0) * S if args.warmup > 0 else 0)
    barrier()
    t0 = time.perf_counter()
    run_units(args.warmup, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # roofline pass: the same K steps, one task at a time, an event pair around every attention launch on the launch
    # stream.  With several task streams the launches of different tasks overlap on the device, so per-launch intervals
    # taken inside the timed region above would measure the mix, not the kernel.
    with torch.no_grad():
        ops.KERNEL_TIMER = timer = []
        t1 = time.perf_counter()
        for u in range(args.steps):
            run_unit(pipe, tasks, args.warmup + args.steps + u, shard)
        torch.cuda.synchronize()
        dt_single = time.perf_counter() - t1
        ops.KERNEL_TIMER = None

    # per-family breakdown from one extra, UNTIMED unit with an event pair around every launch (the event records
    # themselves would cost about 1 % inside the timed region)
    breakdown = None
    if rank == 0 or shard is not None:  # a frame-sharded unit contains collectives: every rank has to run it
        with torch.no_grad():
            ops.PROFILE = prof = []
            run_unit(pipe, tasks, args.warmup + 2 * args.steps, shard)
            torch.cuda.synchronize()
            ops.PROFILE = None
    if rank == 0:
        fam = {}
        for name, work, unit, e0, e1 in prof:
            f = fam.setdefault(name, {"launches": 0, "ms": 0.0, "work": 0.0, "unit": unit})
            f["launches"] += 1
            f["ms"] += e0.elapsed_time(e1)
            f["work"] += work
        breakdown = {k: {"launches": v["launches"], "ms": round(v["ms"], 2),
                         ("tflops" if v["unit"] == "flop" else "gb_per_s"):
                             round(v["work"] / (v["ms"] * 1e-3) / (1e12 if v["unit"] == "flop" else 1e9), 1)}
                     for k, v in fam.items()}
    if world > 1:
        dist.barrier()
    finite = bool(torch.isfinite(tasks["spatial"]["lat"].float()).all() and torch.isfinite(tasks["temporal"]["lat"].float()).all())
    # HBM traffic of the attention kernel from PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    # this same command, tools/… -> profiles/r01_attn_traffic_pmc.json): bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB,
    # the factor 2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section).  Static file, not live.
    traffic = None
    tf = ROOT / "profiles" / "r01_attn_traffic_pmc.json"
    if tf.exists() and (LAT_H, LAT_W) == (72, 40):
        t = json.loads(tf.read_text())
        traffic = int((2 * t["FETCH_SIZE"]["avg_kb"] + t["WRITE_SIZE"]["avg_kb"]) * 1024)
    attn_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in timer)
    attn_fl = sum(f for _, f, _, _ in timer)
    achieved = attn_fl / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0

    if rank == 0:
        out = {
            "metric": "denoised view-frame latents/sec (44cam x 150fr grid)",
            "value": round((1 if shard is not None else world) * args.steps * LATENTS_PER_UNIT / dt, 4),
            "unit": "latents/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if shard is not None else "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": "demo_4d 44cam x 150fr, sliding_fast (window 12, stride 2, 3 rounds, 18 steps/latent), "
                            f"CFG 2.0, {LAT_H}x{LAT_W}x4 latents; step = 2 spatial (F=16) + 1 temporal (F=24) window calls "
                            "= 2 denoised latents; VAE excluded",
                "unet": "SD-2.1 geometry (320,640,1280,1280), 815.6M params, random init seed 0",
                "parallelism": (f"frame-shard x{world} (every window split over all ranks, RCCL K/V all-gather per 3-D "
                                f"attention layer)" if shard is not None else
                                f"task-parallel x{world} (independent tasks per round, no data-path collective)"),
                "task_streams": S,
                "finite_outputs": finite,
                "extensions": ["prune_cond_rows"] if args.prune_cond_rows else [],
            },
            "roofline": {
                "kernel": "attn_kernel (2-D + 3-D view/time attention, all 48 launches of a step)",
                "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_note": "avg HBM bytes per attn launch, PMC (2*FETCH_SIZE+WRITE_SIZE), profiles/r01_attn_traffic_pmc.json; "
                                "algorithmic Q+K+V+O bytes average 152e6 per launch",
                "avg_launch_ms": round(attn_ms / max(1, len(timer)), 4), "launches": len(timer),
                "measured_in": f"a second pass of the same {args.steps} steps with one task in flight "
                               f"({dt_single / args.steps * 1e3:.1f} ms per step), HIP events on the launch stream",
                "share_of_step_time": round(attn_ms * 1e-3 / dt_single, 4),
            },
        }
        # secondary figures of SURVEY.md 8d (whole job): latent-steps/s, UNet window calls/s, sustained UNet TFLOP/s
        ranks_units = (1 if shard is not None else world) * args.steps
        if (LAT_H, LAT_W) == (72, 40):
            unit_tflop = 2 * 20.13 + 33.06  # F=16 / F=24 UNet calls at 72x40
        elif (LAT_H, LAT_W) == (128, 128):
            unit_tflop = 2 * 261.9 + 485.8
        else:
            unit_tflop = None
        out["kernel_breakdown_one_step"] = breakdown
        out["secondary"] = {
            "latent_steps_per_s": round(ranks_units * 3 * WINDOW / dt, 2),
            "unet_calls_per_s": round(ranks_units * 3 / dt, 3),
            "unet_tflops_sustained": round(ranks_units * unit_tflop / dt, 1) if unit_tflop else None,
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only (the CPU sample would skew multi-rank timing)
            out["cpu_baseline"] = cpu_baseline(args.cpu_frames, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
