#!/usr/bin/env python
"""bench.py -- sliding iterative denoiser throughput on MI355X.

Metric (BASELINE.json): denoised view-frame latents / second on the 44-target-camera x 150-frame grid
(`demo_4d`, `sliding_fast`: window 12, stride 2, 3 alternation rounds => 18 denoising steps per
latent, CFG 2.0), synthetic 72x40x4 latents, SD-2.1-geometry UNet with seeded random weights.
Each window call = model-input pack -> UNet -> CFG + batched DDIM update on device-resident latents; VAE
encode/decode is outside the timed region -- inputs are resident in HBM when timing starts.

Modes (--mode; default `auto` = `task` at --gpus 1, `grid` at --gpus N > 1):
  task   one "step" = one schedule unit of the run: 2 spatial window calls (F = 4 inputs + 12 targets = 16 frames,
         CFG batch 32) + 1 temporal window call (F = 12 + 12 = 24 frames, CFG batch 48) -- exactly the 6600 : 3300 call
         mix of the full run = 36 latent-steps = 2 fully denoised latents.  The K timed steps are dealt
         to --task-streams stacks of --task-batch independent tasks in flight (defaults: the runner's gpu_streams and task_batch),
         one HIP stream and worker thread per stack, the tasks of a stack sharing their window calls -- what the runner does with
         the tasks of a round (host/runner.py run_round_pipelined).  With N ranks every rank runs its own K units (weak scaling).
  grid   the REAL round structure of the 48-camera x 150-frame job over N ranks (strong scaling): spatial round (150
         tasks, one per frame), temporal round (44 tasks, one per target camera), spatial round, executed by the
         product's DistributedSamplingRunner -- round-robin task partition, loader / GPU-stream pipelining per rank,
         barrier + RCCL cell exchange at both round boundaries -- on a latents-only pipeline adapter.  Every task runs
         the first c window calls of its sweep (spatial c_s = max(1, K // 5), temporal c_t = round(3.41 c_s): the
         2 : 1 call mix), so wave quantisation (150 and 44 tasks over N GPUs), the exchange and host contention are
         in the number; --steps K sets that depth and `value` counts the latent-steps actually executed / 18.
  hybrid the same grid job with the runner's `hybrid` deal: full waves task-parallel, a last wave that would leave at least half of the
         ranks idle (44 temporal tasks on 8 GPUs: 5 waves + 4) frame-sharded over sub-groups of ranks (host/runner.py).
  frame-shard  every window split over all ranks with RCCL K/V all-gathers (latency mode, BASELINE config 4).
Scaling curves compare like with like: the --gpus 1 line (task mode, `value` = resident steady state) ALSO runs one grid pass and
reports it as `secondary.grid` ({latents_per_s, calls, timed_seconds, window_calls_per_task}); --gpus N > 1 lines (grid mode)
carry the same object, so 1 -> N is `secondary.grid.latents_per_s` over `secondary.grid.latents_per_s`.

`python bench.py --gpus N` launches its own N ranks (torch.distributed.run on 127.0.0.1) when it is not already
running under a launcher; rank 0 prints ONE JSON line: the contract fields, `roofline` (attention kernel: HIP-event
timing in a one-task-at-a-time pass, PMC traffic from profiles/), `cpu_baseline` + `parity` (N = 1 only: the CPU
oracle on one full spatial window with the SAME weights and input as the HIP UNet, timed and compared -- in both precisions of the
product: `parity.modes.fast` is the judged arithmetic, `parity.modes.parity` the one that meets north_star's 1e-3), `secondary`
(`grid`, `prune_cond_rows`, `vae`, `tolerance_mode` = the same step in the fp16 precision, the fastest arithmetic that meets north_star's
1e-3, with its per-family rows, `parity_precision` = the same in the parity precision, `latent128` = the
reference's native size) and `kernel_breakdown_one_step` (per kernel family and per UNet level).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
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
PARITY_STATE_UNITS = 37  # units the task whose first window call is compared with the CPU oracle has been through (see main)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
UNIT_TFLOP = {(72, 40): (20.13, 33.06), (128, 128): (261.9, 485.8)}  # F=16 / F=24 UNet calls


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed, secondary) VAE encode / decode measurement")
    ap.add_argument("--mode", choices=["auto", "task", "grid", "hybrid", "frame-shard"], default="auto",
                    help="auto: task at --gpus 1, grid at --gpus N > 1 (see the module docstring); hybrid = grid with the runner's hybrid "
                         "deal (the tail wave of a round frame-sharded over sub-groups of ranks)")
    ap.add_argument("--latent", default="72x40",
                    help="latent grid HxW: 72x40 = BASELINE.json's synthetic grid (default, the judged line); 128x128 = the "
                         "1024^2 images the reference's demo configs run (SURVEY.md 8d asks for both)")
    ap.add_argument("--task-batch", type=int, default=None,
                    help="tasks of a round per stack of SHARED window calls (the runner's task_batch, host/runner.py; default: the "
                         "runner's default): the stack's tensors lie along the frame axis, every call carries task-batch x F frames, each "
                         "task's result is bitwise what it is alone.  The K steps are dealt to the streams as evenly as K allows and "
                         "every stream stacks its units by at most this many (deal_units).  1 = every task through its own calls")
    ap.add_argument("--task-streams", type=int, default=None,
                    help="stacks of tasks in flight per GPU, each on its own HIP stream and worker thread (the runner's "
                         "gpu_streams; default: the runner's default). 1 = one at a time")
    ap.add_argument("--grid-frames", type=int, default=N_FRAMES, help="grid mode: frames of the (48 camera x T frame) grid")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE.json configs[4]: 48 x 225 grid, sliding_default (window 12, stride 1, 3 rounds = 36 steps per "
                         "latent); task mode, 1 GPU.  An EXTENSION line.  Attention runs the bf16 kernel: the fp8 (e4m3) kernel of "
                         "rounds 2-3 measured 0.92-1.10x of it including its pack kernels (profiles/r02_bench_config5_fp8.json) "
                         "and was removed in round 4 (DESIGN.md section 0)")
    ap.add_argument("--prune-cond-rows", action="store_true",
                    help="opt-in extension, NOT the judged configuration: skip the per-frame tail of the UNet (after the last "
                         "3-D attention) for conditioning frames, whose noise prediction the reference discards")
    ap.add_argument("--no-grid-secondary", action="store_true",
                    help="task mode at --gpus 1: skip the extra (untimed-for-`value`) pass over the real 48 x 150 round structure "
                         "that fills `secondary.grid` (the same-mode baseline of the N > 1 grid lines)")
    ap.add_argument("--no-parity-precision", action="store_true",
                    help="skip secondary.parity_precision (the same units under model.precision=parity, untimed for `value`)")
    ap.add_argument("--no-tolerance-mode", action="store_true",
                    help="skip secondary.tolerance_mode (the same units under model.precision=fp16: the fastest arithmetic that meets "
                         "north_star's 1e-3; untimed for `value`)")
    ap.add_argument("--precision", choices=["fast", "fp16", "parity"], default="fast",
                    help="arithmetic of the main measurement (default fast = the judged bf16 line; fp16 / parity: profiling aid for the "
                         "other precisions -- `dtype` and `config.precision` say which one ran)")
    ap.add_argument("--cpu-budget-s", type=float, default=150.0,
                    help="wall-time cap of the CPU baseline's timed oracle forwards (1 warm-up + up to 3 timed F = 16 calls + one F = 24 "
                         "call as the budget allows; at least one timed call always runs)")
    ap.add_argument("--no-latent128", action="store_true",
                    help="skip secondary.latent128 (2 units at the reference's native 128 x 128 latents, untimed for `value`)")
    ap.add_argument("--no-parity-bf16", action="store_true",
                    help="skip the second CPU forward (the oracle in bf16 = the reference's own arithmetic) of the `parity` object")
    ap.add_argument("--cpu-frames", type=int, default=16,
                    help="frames of the CPU-baseline UNet call (16 = a full spatial window, about 20 s with 32 threads)")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="torch threads of the CPU baseline: on the 256-thread GPU hosts 32 threads are 3x faster than 64 "
                         "and 50x faster than 256 on this model (tools/dev/cpu_baseline_probe.py)")
    return ap.parse_args()


def self_launch(args) -> None:
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script on this node and relay rank 0's line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    raise SystemExit(subprocess.call(cmd, env=env))


# ---------------------------------------------------------------------------------------------------------------------
# task mode: resident synthetic tasks, units of 2 spatial + 1 temporal window calls
# ---------------------------------------------------------------------------------------------------------------------
def deal_units(count: int, streams: int, batch: int):
    """`count` units -> for every stream the sizes of the stacks it runs: units per stream as even as `count` allows, every stream's
    units in ceil(units / batch) stacks as even as possible (20 units, 2 streams, batch 2 -> 5 stacks of 2 each; batch 3 -> 3 + 3 + 2 + 2
    each) -- the rule run_round_pipelined applies to the tasks of a round."""
    per = [count // streams + (1 if s < count % streams else 0) for s in range(streams)]
    out = []
    for u in per:
        k = -(-u // max(1, batch))
        out.append([u // k + (1 if i < u % k else 0) for i in range(k)])
    return out


def build_tasks(pipe, dev, shard=None, copies: int = 1):
    """Device-resident synthetic task tensors + window plans for one spatial and one temporal task (`copies` of each, stacked along
    the frame axis: a stack of tasks sharing their window calls)."""
    from diffuman4d_amd.host.schedule import plan_sweep
    g = torch.Generator(device=dev).manual_seed(1234)

    dt = getattr(pipe, "dtype", torch.bfloat16)  # fp32 task tensors under precision "parity" (bf16-representable values either way)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=dev) * scale).to(torch.bfloat16).to(dt)

    tasks = {}
    for domain, n, cond in (("spatial", N_CAMS, [i in INPUT_CAMS for i in range(N_CAMS)]),
                            ("temporal", 2 * N_FRAMES, [i < N_FRAMES for i in range(2 * N_FRAMES)])):
        plan = plan_sweep(cond, [0] * n, domain, WINDOW, STRIDE, 0, False, 1, ROUNDS)
        kb = copies
        mask = torch.tensor([0.0 if c else 1.0 for c in cond] * kb, device=dev).to(dt)
        hw = LAT_H * LAT_W
        tasks[domain] = dict(
            pv=rnd(kb * n, hw, 4, scale=0.18215 * 4), pl=rnd(kb * n, hw, 6, scale=0.5).clamp(-1, 1),
            sk=rnd(kb * n, hw, 4, scale=0.18215 * 4), lat=rnd(kb * n, hw, 4),
            cm=mask[:, None, None].expand(kb * n, hw, 1).contiguous(), plan=plan,
            tables=pipe.upload_plan(plan, GUIDANCE, shard, copies=kb, rows_per_task=n), domain=domain, copies=kb)
    return tasks


class UnitRunner:
    """The units of a measurement on S task streams (one HIP stream + worker thread each) in stacks of at most `batch` units: what
    runner.run_round_pipelined does with the tasks of a round (gpu_streams, task_batch), on resident synthetic tasks."""

    def __init__(self, pipe, dev, streams: int, batch: int, shard=None):
        self.pipe, self.dev, self.S, self.batch, self.shard = pipe, dev, max(1, streams), max(1, batch), shard
        self.sets = {}
        self.on_gpu = torch.device(dev).type == "cuda"  # (a CPU stand-in pipeline in tests/test_bench_grid.py: threads, no streams)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.S)] if (self.S > 1 and self.on_gpu) else [None] * self.S

    def task_set(self, si: int, size: int):
        """Task state of stream `si` for stacks of `size` units (built on first use: call `prepare` before anything is timed)."""
        if (si, size) not in self.sets:
            self.sets[(si, size)] = build_tasks(self.pipe, self.dev, self.shard, copies=size)
        return self.sets[(si, size)]

    def prepare(self, *counts):
        """Builds (and runs once: allocator + kernel warm) every task state the given unit counts will use."""
        with torch.no_grad():
            for c in counts:
                for si, sizes in enumerate(deal_units(c, self.S, self.batch)):
                    for z in set(sizes):
                        if (si, z) not in self.sets:
                            with torch.cuda.stream(self.streams[si]) if self.streams[si] is not None else _nullctx():
                                run_unit(self.pipe, self.task_set(si, z), 0, self.shard)
        if self.on_gpu:
            torch.cuda.synchronize()

    def prepare_single(self, *counts):
        """The same for `run_single_stream` (stream 0's task states, the current stream)."""
        with torch.no_grad():
            for c in counts:
                for z in set(deal_units(c, 1, self.batch)[0]):
                    if (0, z) not in self.sets:
                        run_unit(self.pipe, self.task_set(0, z), 0, self.shard)
        if self.on_gpu:
            torch.cuda.synchronize()

    def _stream_work(self, si, sizes, first):
        if self.on_gpu:
            torch.cuda.set_device(self.dev)
        with torch.no_grad(), (torch.cuda.stream(self.streams[si]) if self.streams[si] is not None else _nullctx()):
            for j, z in enumerate(sizes):
                run_unit(self.pipe, self.task_set(si, z), first + j, self.shard)

    def run(self, first: int, count: int):
        """`count` units (deal_units), `first` = index of the window calls the first stack of every stream runs."""
        if count <= 0:
            return
        deal = deal_units(count, self.S, self.batch)
        if self.S == 1:
            return self._stream_work(0, deal[0], first)
        errs = []

        def guarded(si):
            try:
                self._stream_work(si, deal[si], first)
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=guarded, args=(si,)) for si in range(self.S) if deal[si]]
        [t.start() for t in th]
        [t.join() for t in th]
        if errs:
            raise errs[0]

    def run_single_stream(self, first: int, count: int):
        """The same stacks one at a time on the CURRENT stream (stream 0's task states): per-launch event pairs measure kernels, not
        the mix of overlapping streams.  Returns the stack sizes run."""
        sizes = deal_units(count, 1, self.batch)[0]
        with torch.no_grad():
            for j, z in enumerate(sizes):
                run_unit(self.pipe, self.task_set(0, z), first + j, self.shard)
        return sizes

    def finite(self) -> bool:
        return all(bool(torch.isfinite(t[d]["lat"].float()).all()) for t in self.sets.values() for d in ("spatial", "temporal"))


class _nullctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def run_call(pipe, task, call_idx, shard=None):
    """One window call: pack -> UNet -> CFG + DDIM (Diffuman4DPipeline.window_call, the body of denoise_latents)."""
    tb = task["tables"]
    pipe.window_call(task["lat"], task["pv"], task["pl"], task["sk"], task["cm"], tb, call_idx % tb["calls"], LAT_H, LAT_W,
                     [task["domain"]] * 2, GUIDANCE, True, False, shard)


def run_unit(pipe, tasks, u, shard=None):
    run_call(pipe, tasks["spatial"], 2 * u, shard)
    run_call(pipe, tasks["spatial"], 2 * u + 1, shard)
    run_call(pipe, tasks["temporal"], u, shard)


# ---------------------------------------------------------------------------------------------------------------------
# grid mode: the product's sampler + distributed runner on a latents-only pipeline adapter
# ---------------------------------------------------------------------------------------------------------------------
class LatentGridPipeline:
    """The pipeline protocol without the VAE, for `--mode grid`: the task's latents come from / go back to
    the sampler's grid, the conditioning latents are resident synthetic tensors (what the VAE + resize stage would have
    produced), and the first `depth[domain]` window calls of the task's sweep are executed; the sweep is booked as
    complete so that the next round's plan is consistent (values of a truncated sweep mean nothing, cost does)."""

    def __init__(self, pipe, depth):
        self.pipe, self.depth = pipe, dict(depth)
        self.calls_run = 0
        self._cond = {}
        self._lock = threading.Lock()

    @property
    def device(self):
        return self.pipe.device

    def clear_vae_cache(self):
        pass

    def _conditioning(self, domain, cond_flags):
        key = (domain, len(cond_flags))
        with self._lock:
            if key not in self._cond:
                dev, n, hw = self.pipe.device, len(cond_flags), LAT_H * LAT_W
                g = torch.Generator(device=dev).manual_seed(77 + n)
                rnd = lambda c, s: (torch.randn(n, hw, c, generator=g, device=dev) * s).to(torch.bfloat16)  # noqa: E731
                mask = torch.tensor([0.0 if c else 1.0 for c in cond_flags], device=dev).to(torch.bfloat16)
                self._cond[key] = (rnd(4, 0.18215 * 4), rnd(6, 0.5).clamp(-1, 1), rnd(4, 0.18215 * 4),
                                   mask[:, None, None].expand(n, hw, 1).contiguous())
            return self._cond[key]

    @torch.no_grad()
    def sliding_iterative_denoise(self, pixel_values=None, plucker_embeds=None, skeletons=None, cond_masks=None, latents=None,
                                  domain="spatial", timestep_indices=None, window_size=12, sliding_stride=1, sliding_shift=0,
                                  bidirectional=True, num_denoising_steps=1, alternation_rounds=3, guidance_scale=2.0,
                                  tqdm=None, shard=None, noise_seed=None, **_ext):
        return self._sweep([dict(cond_masks=cond_masks, latents=latents, timestep_indices=timestep_indices, noise_seed=noise_seed)], domain,
                           window_size, sliding_stride, sliding_shift, bidirectional, num_denoising_steps, alternation_rounds,
                           guidance_scale, shard)[0]

    @torch.no_grad()
    def sliding_iterative_denoise_stack(self, tasks, domain="spatial", window_size=12, sliding_stride=1, sliding_shift=0, bidirectional=True,
                                        num_denoising_steps=1, alternation_rounds=3, guidance_scale=2.0, tqdm=None, decode="all"):
        """runner.task_batch: the tasks' latents stacked along the frame axis, one set of window calls (pipeline.py
        sliding_iterative_denoise_stack)."""
        return self._sweep(tasks, domain, window_size, sliding_stride, sliding_shift, bidirectional, num_denoising_steps,
                           alternation_rounds, guidance_scale, None)

    def _sweep(self, tasks, domain, window_size, sliding_stride, sliding_shift, bidirectional, num_denoising_steps, alternation_rounds,
               guidance_scale, shard):
        from diffuman4d_amd.host.schedule import plan_sweep
        pipe, dev = self.pipe, self.pipe.device
        on_gpu = torch.device(dev).type == "cuda"
        if on_gpu:
            from diffuman4d_amd.host import ops
            torch.cuda.set_device(dev)
            to_nhwc, to_nchw = ops.nchw_to_nhwc, ops.nhwc_to_nchw
        else:  # CPU stand-in pipelines in tests/test_bench_grid.py (layout plumbing only; there is no CPU compute path)
            to_nhwc, to_nchw = (lambda t: t.permute(0, 2, 3, 1).contiguous()), (lambda t: t.permute(0, 3, 1, 2).contiguous())
        plans, lats = [], []
        hw = LAT_H * LAT_W
        for t in tasks:
            cond_flags = (t["cond_masks"][:, 0, 0, 0] == 0.0).cpu().numpy()
            plans.append(plan_sweep(cond_flags, torch.as_tensor(t["timestep_indices"]).cpu().numpy(), domain, window_size, sliding_stride,
                                    sliding_shift, bidirectional, num_denoising_steps, alternation_rounds))
            n = len(cond_flags)
            if t["latents"] is None:
                seed = t.get("noise_seed")
                gen = None if seed is None else torch.Generator(device=dev).manual_seed(int(seed))  # a shard group draws alike
                lats.append(torch.randn(n, hw, 4, device=dev, generator=gen).to(torch.bfloat16))
            else:
                lats.append(to_nhwc(t["latents"].to(device=dev, dtype=torch.bfloat16).contiguous()).view(n, hw, 4))
        plan, kb = plans[0], len(tasks)
        if any(len(p.windows) != len(plan.windows) or not np.array_equal(p.final_timestep_indices, plan.final_timestep_indices) for p in plans[1:]):
            raise ValueError("tasks of a stack need identical window plans")
        pv, pl, sk, cm = self._conditioning(domain, cond_flags)
        if kb > 1:
            key = (domain, n, kb)
            with self._lock:
                if key not in self._cond:
                    self._cond[key] = tuple(torch.cat([x] * kb) for x in (pv, pl, sk, cm))
                pv, pl, sk, cm = self._cond[key]
        lat = torch.cat(lats) if kb > 1 else lats[0]
        tb = pipe.upload_plan(plan, guidance_scale, shard, copies=kb, rows_per_task=n if kb > 1 else 0)
        k = min(self.depth[domain], tb["calls"])
        for i in range(k):
            pipe.window_call(lat, pv, pl, sk, cm, tb, i, LAT_H, LAT_W, [domain] * tb["cfg"], guidance_scale, tb["cfg"] == 2, False, shard)
        with self._lock:
            self.calls_run += kb * k / (shard.world if shard is not None else 1)  # a rank of a shard group ran 1 / width of each call
        tidx = torch.from_numpy(plan.final_timestep_indices)
        return [{"images": torch.zeros(n, 3, 1, 1), "latents": to_nchw(lat[j * n:(j + 1) * n].view(n, LAT_H, LAT_W, 4)),
                 "timestep_indices": tidx, "fully_denoised": tidx == plan.num_inference_steps} for j in range(kb)]


def grid_depth(steps: int):
    """Window calls per task from --steps: spatial c_s = max(1, K // 5), temporal c_t = round(3.41 c_s) keeps the full
    run's 6600 : 3300 call mix over 300 spatial and 44 temporal tasks (K = 20 -> 4 and 14 = 1 816 calls, 18 % of the run; the full
    sweeps are 22 and 75).  Round 2 used K // 10: at 8 GPUs a rank then ran for about 3 s per pass, too little against the fixed
    costs of a pass (runner start-up, two barriers, two exchanges)."""
    cs = max(1, steps // 5)
    return {"spatial": min(cs, 22), "temporal": min(max(1, round(2 * N_FRAMES * cs / (2 * 44))), 75)}


def run_grid_pass(pipe, depth, frames, world, rank, gpu_streams, runner_mode="task", task_batch=1):
    """One pass over the whole (48 x frames) grid job: 3 alternation rounds through the product's runner.  Returns the
    number of window calls THIS rank executed."""
    from diffuman4d_amd.host.dataset import SyntheticSpaTemDataset
    from diffuman4d_amd.host.runner import DistributedSamplingRunner, run_round_pipelined
    from diffuman4d_amd.host.sampler import SlidingIterativeSampler
    adapter = LatentGridPipeline(pipe, depth)
    ds = SyntheticSpaTemDataset(height=8, width=8, num_cameras=N_CAMS)  # pixel data is never read by the adapter
    sampler = SlidingIterativeSampler(ds, [adapter], "/tmp/dm4d_bench_unused", window_size=WINDOW, sliding_stride=STRIDE,
                                      sliding_shift=0, bidirectional=False, num_denoising_steps=1, alternation_rounds=ROUNDS,
                                      guidance_scale=GUIDANCE, spa_label_range=(0, N_CAMS, 1), tem_label_range=(0, frames, 1),
                                      input_spa_labels=INPUT_CAMS)
    sampler.result_writer = None
    if world > 1:
        runner = DistributedSamplingRunner(sampler, prefetch_depth=2, writers=1, gpu_streams=gpu_streams, mode=runner_mode,
                                           task_batch=task_batch)
        runner.inference()
        # the deal of the last round (rate-weighted when the ranks' speeds differ) + the tail tasks this rank ran in a group
        last_tasks = runner.tasks_of(ROUNDS - 1, rank) + [t for t, ranks in runner.tail_of(ROUNDS - 1) if rank in ranks]
    else:
        for tasks in sampler.all_tasks:
            run_round_pipelined(sampler, tasks, 0, 2, 1, gpu_streams, task_batch=task_batch)
        last_tasks = sampler.all_tasks[ROUNDS - 1]
    done = all(sampler.timestep_indices[c][f] == STEPS_PER_LATENT
               for t in last_tasks for c in sampler.target_spa_labels for f in [t["domain_label"]])
    if not done:
        raise RuntimeError("grid pass: a target cell of this rank's last-round tasks did not reach the final timestep index")
    return adapter.calls_run


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline + parity on the judged configuration
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline_and_parity(pipe, state_dict, task, frames: int, threads: int, bf16_oracle: bool = True, task24=None,
                            budget_s: float = 150.0, task_fresh=None):
    """The CPU oracle (oracle/: plain PyTorch restatement of the reference) on one spatial-window UNet call with the SAME
    weights and the SAME packed input as the HIP UNet: timed (cpu_baseline) and compared (parity).
    Timing protocol: fp32, `threads` torch threads, 1 warm-up + up to 3 timed F = 16 forwards (median) and one F = 24
    forward of the temporal window, as far as `budget_s` seconds of timed CPU work allow (one timed F = 16 forward always runs).
    `task_fresh`: the same task before any unit has run (fresh N(0, 1) latents): the warm-up forward then runs on ITS packed input and
    doubles as the reference of a second comparison, so that `parity` reports the call at both ends of the range of states a job goes
    through (the distance of a UNet call to the fp32 oracle depends on its input: tools/dev/parity_state_probe.py)."""
    from diffuman4d_amd.host import ops
    from oracle.unet import UNetConfig, UNetMultiviewConditionModel
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    tb = task["tables"]
    widx, cond = tb["win"][0][:frames], tb["cond"][0][:frames]
    spent = 0.0
    with torch.no_grad():
        x = ops.pack_model_input(task["lat"].clone(), task["pv"], task["pl"], task["sk"], task["cm"], cond.contiguous(),
                                 pipe.unet.IN_PAD, True, frame_idx=widx.contiguous())
        t_in = torch.cat([tb["t"][0][:tb["win"].shape[1]][:frames]] * 2)
        B = 2 * frames
        out = pipe.unet(x.view(B, LAT_H, LAT_W, pipe.unet.IN_PAD), t_in, domains=["spatial"] * 2, num_frames=frames)
        hip = ops.nhwc_to_nchw(out).float().cpu()
        cfg = UNetConfig()
        m = UNetMultiviewConditionModel(cfg).eval()
        m.load_state_dict({k: v.float().cpu() for k, v in state_dict.items()}, strict=True)
        x_cpu = ops.nhwc_to_nchw(x.view(B, LAT_H, LAT_W, pipe.unet.IN_PAD), cfg.in_channels).float().cpu()
        t_cpu = t_in.cpu().long()
        xs = {"state": x}
        if task_fresh is not None:
            xs["fresh"] = ops.pack_model_input(task_fresh["lat"].clone(), task_fresh["pv"], task_fresh["pl"], task_fresh["sk"], task_fresh["cm"],
                                               cond.contiguous(), pipe.unet.IN_PAD, True, frame_idx=widx.contiguous())
        t0 = time.time()
        # warm-up (allocator, thread pool); also a parity reference: of the fresh-latents input when there is one, else of the state input
        x_warm = ops.nhwc_to_nchw(xs.get("fresh", x).view(B, LAT_H, LAT_W, pipe.unet.IN_PAD), cfg.in_channels).float().cpu()
        refs = {"fresh" if task_fresh is not None else "state": m(x_warm, t_cpu, domains=["spatial"] * 2, num_frames=frames)}
        warm = time.time() - t0
        del x_warm
        times = []
        while len(times) < 3 and (not times or spent + 1.1 * times[-1] <= budget_s):
            t0 = time.time()
            r = m(x_cpu, t_cpu, domains=["spatial"] * 2, num_frames=frames)
            times.append(time.time() - t0)
            spent += times[-1]
            refs.setdefault("state", r)
        ref = refs["state"]
        dt = sorted(times)[len(times) // 2]
        dt24 = None
        if task24 is not None and frames >= 16 and spent + 1.7 * dt <= budget_s:  # one F = 24 temporal-window forward (CFG batch 48)
            tb2 = task24["tables"]
            f2 = tb2["win"].shape[1]
            x2 = ops.pack_model_input(task24["lat"].clone(), task24["pv"], task24["pl"], task24["sk"], task24["cm"], tb2["cond"][0].contiguous(),
                                      pipe.unet.IN_PAD, True, frame_idx=tb2["win"][0].contiguous())
            x2_cpu = ops.nhwc_to_nchw(x2.view(2 * f2, LAT_H, LAT_W, pipe.unet.IN_PAD), cfg.in_channels).float().cpu()
            t2 = torch.cat([tb2["t"][0][:f2]] * 2).cpu().long()
            t0 = time.time()
            m(x2_cpu, t2, domains=["temporal"] * 2, num_frames=f2)
            dt24 = time.time() - t0
            del x2, x2_cpu
    err = float((hip - ref).norm() / ref.norm())
    fresh = {}
    if "fresh" in refs:
        with torch.no_grad():
            o = pipe.unet(xs["fresh"].view(B, LAT_H, LAT_W, pipe.unet.IN_PAD), t_in, domains=["spatial"] * 2, num_frames=frames)
        fresh["fast"] = float((ops.nhwc_to_nchw(o).float().cpu() - refs["fresh"]).norm() / refs["fresh"].norm())
    # the same call under the two wide precisions of the product: same weights, same inputs
    errs = {}
    with torch.no_grad():
        from diffuman4d_amd.host.unet import UNetConfig as HC, UNetMultiviewConditionModel as HU
        for prec in ("parity", "fp16"):
            up = HU(HC(), state_dict, pipe.device, prec)
            for which, xin in xs.items():
                if which not in refs:
                    continue
                xf = xin.view(B, LAT_H, LAT_W, pipe.unet.IN_PAD).float()  # the packed input is bf16-valued: exact in fp32 and in fp16
                outp = ops.nhwc_to_nchw(up(ops.split(xf, h16=prec == "fp16"), t_in, domains=["spatial"] * 2, num_frames=frames)).float().cpu()
                e = float((outp - refs[which]).norm() / refs[which].norm())
                if which == "state":
                    errs[prec] = e
                else:
                    fresh[prec] = e
                del outp
            del up
            torch.cuda.empty_cache()
    err_vs_bf16 = yard_live = dt_bf = None
    if bf16_oracle:  # the reference's own arithmetic (configs/model/diffuman4d.yaml: bf16): the same oracle, parameters and activations in bf16
        with torch.no_grad():
            m.to(torch.bfloat16)
            t0 = time.time()
            ref_bf = m(x_cpu.to(torch.bfloat16), t_cpu, domains=["spatial"] * 2, num_frames=frames).float()
            dt_bf = time.time() - t0
        err_vs_bf16 = float((hip - ref_bf).norm() / ref_bf.norm())
        yard_live = float((ref_bf - ref).norm() / ref.norm())
    # a unit = 2 F=16 calls + 1 F=24 call; the F=24 call is measured when the budget allowed it, else priced by its FLOP ratio
    f16, f24 = UNIT_TFLOP.get((LAT_H, LAT_W), (1.0, 1.64))
    unit_s = (2 * dt + (dt24 if dt24 is not None else dt * f24 / f16)) if frames >= 16 else None
    base = {
        "value": round(LATENTS_PER_UNIT / unit_s, 5) if unit_s else round(frames * (WINDOW / 16) / STEPS_PER_LATENT / dt, 5),
        "unit": "latents/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"CPU oracle UNet forwards (fp32, {torch.get_num_threads()} torch threads, {LAT_H}x{LAT_W} latents, the bench weights): "
                  f"1 warm-up ({warm:.1f} s) + {len(times)} timed spatial-window calls (F={frames}, CFG batch {B}; median {dt:.1f} s)"
                  + (f" + 1 temporal-window call (F=24, CFG batch 48: {dt24:.1f} s)" if dt24 is not None else
                     f"; the F=24 call priced as {f24 / f16:.3f} x the F=16 call by FLOP ratio (--cpu-budget-s {budget_s:g} reached)")
                  + "; a unit = 2 F=16 calls + 1 F=24 call = 2 latents.  Whole-grid CPU numbers would be this rate extrapolated by call counts",
        "seconds": round(dt, 2), "seconds_f16_all": [round(t, 2) for t in times], "seconds_f24": None if dt24 is None else round(dt24, 2),
        "warmup_seconds": round(warm, 2),
    }
    yard = None
    gold = ROOT / "tests" / "golden" / "sd21_72x40.pt"
    if gold.exists() and (LAT_H, LAT_W) == (72, 40):
        yard = torch.load(gold).get("unet_f16_spatial", {}).get("yard_bf16")
    parity = {
        "case": f"UNet forward, SD-2.1 geometry, F={frames} spatial window, CFG batch {B}, {LAT_H}x{LAT_W}: HIP vs CPU oracle (fp32), "
                "same weights and input, in the three precisions of the product",
        # precision "fast" = the judged throughput (`value`); "fp16" = the fastest arithmetic that meets north_star's tolerance (decoded RGB
        # of a whole task: tests/modelcheck.py fp16_demo3d_sd21_72x40), its throughput is secondary.tolerance_mode; "parity" = the two-term
        # arithmetic at 1e-5 (secondary.parity_precision).  The bar itself is on decoded RGB; this is the single UNet call behind it.
        # rel_l2: the call on the state the task is in after PARITY_STATE_UNITS units (what rounds 3-5 reported); rel_l2_fresh_latents:
        # the same call on fresh N(0, 1) latents (a job's first round); meets_north_star: BOTH within 1e-3 -- of the single call; the
        # bar itself is on the decoded RGB of a task (the GPU suite's *_demo3d_* and *_demo4dtiny_temporal_* cases)
        "modes": {k: {"rel_l2": float(f"{e:.3e}"), "rel_l2_fresh_latents": None if k not in fresh else float(f"{fresh[k]:.3e}"),
                      "meets_north_star": bool(max(e, fresh.get(k, 0.0)) <= 1e-3)}
                  for k, e in (("fast", err), ("fp16", errs["fp16"]), ("parity", errs["parity"]))},
        "rel_l2": float(f"{err:.3e}"), "north_star_tolerance": 1e-3, "meets_north_star": bool(float(f"{err:.3e}") <= 1e-3),  # the figure of modes.fast
        # the three distances between {HIP, oracle fp32, oracle bf16} on THIS input and THESE weights
        "hip_vs_oracle_fp32": float(f"{err:.3e}"),
        "hip_vs_oracle_bf16": None if err_vs_bf16 is None else round(err_vs_bf16, 6),
        "oracle_bf16_vs_oracle_fp32": None if yard_live is None else round(yard_live, 6),
        "oracle_bf16_seconds": None if dt_bf is None else round(dt_bf, 1),
        "yardstick_oracle_bf16_vs_fp32": yard,
        "note": "fast precision: bf16 tensors and MFMA operands -- the reference's own bf16 arithmetic (the oracle run in bf16) is as far "
                "from the fp32 oracle as this path is; fp16 precision (model.precision=fp16): fp32 tensors between kernels and single-term "
                "fp16 operands; parity precision (model.precision=parity): fp32 tensors and two-term bf16 operands (DESIGN.md section 3)",
    }
    return base, parity


# what `secondary.tolerance_mode.meets_north_star` rests on: decoded RGB of whole tasks at the judged geometry against the fp32 oracle,
# fixed bound 1e-3, in `pytest -m gpu` (tests/modelcheck.py); measured values: profiles/r06_modelcheck_fp16_tasks.log
TOLERANCE_EVIDENCE = [
    "tests/modelcheck.py fp16_demo3d_sd21_72x40: BASELINE configs[0], the whole spatial task (44 window calls of F = 16, 48 decodes)",
    "tests/modelcheck.py fp16_demo4dtiny_temporal_sd21_72x40: BASELINE configs[1], one whole temporal task of the middle round (8 window calls of F = 24)",
    "tests/modelcheck.py fp16_multiround_sd21_72x40: spatial -> temporal -> spatial through the sampler (36 window calls)",
]
PRECISION_NOTES = {
    "parity": "precision 'parity' (fp32 tensors between kernels, two-term bf16 MFMA operands on K-duplicated weights, three MFMA terms per "
              "attention product): 1e-5 from the fp32 reference path; not the judged value",
    "fp16": "precision 'fp16' (fp32 tensors between kernels, single-term fp16 MFMA operands, one MFMA per product): the fastest arithmetic "
            "within north_star's 1e-3 of the fp32 reference path on decoded RGB (tests/modelcheck.py fp16_demo3d_sd21_72x40); not the "
            "judged value",
}


def family_rows(prof, size: int):
    """Per kernel family (and per UNet level) of ONE profiled stack of `size` units: launches of the stack, ms PER UNIT, achieved rate on
    the algorithmic work and its fraction of the roofline that bounds it (dense bf16 MFMA peak 2500 TFLOP/s, HBM 8000 GB/s)."""
    # UNet level of a launch from its output rows: a unit's calls carry CFG batch 32 (F = 16) or 48 (F = 24) per task of the stack,
    # level l has LAT_H * LAT_W / 4^l rows per sample
    level_of = {size * b * (LAT_H * LAT_W) // 4 ** l: f"L{l}" for b in (2 * (WINDOW + len(INPUT_CAMS)), 4 * WINDOW) for l in range(4)}
    fam = {}
    for name, work, unit, e0, e1, rows in prof:
        ms = e0.elapsed_time(e1)
        for key in (name, f"{name}.{level_of[rows]}" if rows in level_of else None):
            if key is None:
                continue
            f = fam.setdefault(key, {"launches": 0, "ms": 0.0, "work": 0.0, "unit": unit})
            f["launches"] += 1
            f["ms"] += ms
            f["work"] += work
    out = {}
    for k, v in fam.items():
        rate = v["work"] / (v["ms"] * 1e-3) / (1e12 if v["unit"] == "flop" else 1e9)
        out[k] = {"launches": v["launches"], "ms": round(v["ms"] / size, 2), ("tflops" if v["unit"] == "flop" else "gb_per_s"): round(rate, 1),
                  "roofline_frac": round(rate / (2500.0 if v["unit"] == "flop" else 8000.0), 3)}
    return out


def precision_secondary(cfg, state_dict, dev, units: int, precision: str = "parity", streams: int = 1, batch: int = 1):
    """Throughput of a wide precision (model.precision=parity | fp16) on the SAME units as `value`: `streams` stacks of `batch` tasks in
    flight the way the timed region of `value` runs them (ms_per_step), then one stack at a time with an event pair around every launch for
    the per-family rows.  Reported beside the fast precision, never as `value`."""
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMScheduler
    from diffuman4d_amd.host.unet import UNetMultiviewConditionModel
    from diffuman4d_amd.host import ops
    pp = Diffuman4DPipeline(None, UNetMultiviewConditionModel(cfg, state_dict, dev, precision), DDIMScheduler(), dev)
    ur = UnitRunner(pp, dev, max(1, min(streams, units)), batch)
    S = ur.S
    n_one = 2 * min(batch, units)
    ur.prepare(units, S)
    ur.prepare_single(n_one, n_one // 2)
    ur.run(0, S)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ur.run(1, units)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # one stack at a time (what the fast precision's dt_single / breakdown pass measures)
    ur.run_single_stream(2 + units, n_one // 2)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ur.run_single_stream(3 + units, n_one)
    torch.cuda.synchronize()
    dt_one = (time.perf_counter() - t1) / n_one
    # where that step goes: one more stack with an event pair around every launch (untimed), summed per kernel family.  Rates are on
    # the work EXECUTED in this precision (parity: K doubled in every GEMM / conv, three MFMA terms per attention product; both: fp32
    # input bytes in the normalisations), against the same peaks as the fast precision's rows
    ops.PROFILE = prof = []
    size = ur.run_single_stream(6 + units, n_one // 2)[0]
    torch.cuda.synchronize()
    ops.PROFILE = None
    families = family_rows(prof, size)
    finite = ur.finite()
    del pp, ur
    torch.cuda.empty_cache()
    return {"precision": precision, "ms_per_step": round(dt / units * 1e3, 3), "latents_per_s": round(LATENTS_PER_UNIT * units / dt, 4),
            "steps": units, "task_streams": S, "task_batch": batch, "ms_per_step_one_task": round(dt_one * 1e3, 3),
            "one_task_note": "ms per step with ONE stack of task_batch tasks in flight", "finite_outputs": finite,
            "kernel_breakdown_one_step": families, "note": PRECISION_NOTES[precision]}


def latent128_secondary(pipe, units: int = 2):
    """The same unit at the reference's NATIVE latent size (spatem_dataset.py:27-28: 1024^2 images -> 128 x 128 latents), fast
    precision, one task at a time: ms per unit and the attention kernel's share / rate (at this size attention is two thirds of the step)."""
    global LAT_H, LAT_W
    from diffuman4d_amd.host import ops
    keep = (LAT_H, LAT_W)
    LAT_H, LAT_W = 128, 128
    try:
        tasks = build_tasks(pipe, pipe.device)
        with torch.no_grad():
            run_unit(pipe, tasks, 0)
            torch.cuda.synchronize()
            ops.KERNEL_TIMER = timer = []
            t0 = time.perf_counter()
            for u in range(units):
                run_unit(pipe, tasks, 1 + u)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ops.KERNEL_TIMER = None
        attn_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in timer)
        attn_fl = sum(f for _, f, _, _ in timer)
        rate = attn_fl / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
        ut = UNIT_TFLOP[(128, 128)]
        del tasks
        torch.cuda.empty_cache()
        return {"latent": "128x128", "ms_per_step": round(dt / units * 1e3, 2), "latents_per_s": round(LATENTS_PER_UNIT * units / dt, 4),
                "steps": units, "task_streams": 1, "unet_tflops_sustained": round(units * (2 * ut[0] + ut[1]) / dt, 1),
                "attention": {"tflops": round(rate, 1), "roofline_frac": round(rate / MFMA_PEAK_TFLOPS, 4),
                              "share_of_step_time": round(attn_ms * 1e-3 / dt, 4), "launches": len(timer)}}
    finally:
        LAT_H, LAT_W = keep
        ops.KERNEL_TIMER = None


def vae_secondary(dev):
    """SD-geometry AutoencoderKL: encode and decode time per image (outside the timed region: SURVEY.md 8d reports the VAE
    separately) on 8 images of the bench's pixel size AND on 2 images of 1024 x 1024, the reference's native image size
    (spatem_dataset.py:27-28; mid-block attention over 16 384 tokens).  FLOPs scale with the pixel count from SURVEY's
    per-image figures at 576x320 (0.78 TF encode, 1.76 TF decode); the mid-block attention's quadratic part is not in them."""
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    from diffuman4d_amd.host.weights import random_state_dict, vae_param_shapes
    cfg = VAEConfig()
    vae = AutoencoderKL(cfg, random_state_dict(vae_param_shapes(cfg), 1, dev), dev)

    def measure(H, W, n):
        g = torch.Generator(device=dev).manual_seed(5)
        img = (torch.rand(n, 3, H, W, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
        noise = torch.randn(n, 4, H // 8, W // 8, generator=g, device=dev).to(torch.bfloat16)
        res = {}
        with torch.no_grad():
            z = vae.encode_scaled(img, noise)
            vae.decode_to_images(z)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            for name, fn, tf in (("encode", lambda: vae.encode_scaled(img, noise), 0.78), ("decode", lambda: vae.decode_to_images(z), 1.76)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 3 / n
                flops = tf * 1e12 * (H * W) / (576 * 320)
                res[name] = {"ms_per_image": round(ms, 3), "tflops": round(flops / (ms * 1e-3) / 1e12, 1)}
            # where an encode / a decode goes: one more pass with an event pair around every launch, summed per kernel family (per image)
            from diffuman4d_amd.host import ops
            for name, fn in (("encode", lambda: vae.encode_scaled(img, noise)), ("decode", lambda: vae.decode_to_images(z))):
                ops.PROFILE = prof = []
                fn()
                torch.cuda.synchronize()
                ops.PROFILE = None
                fam = {}
                for fname, work, unit, e0, e1, _rows in prof:
                    f = fam.setdefault(fname, {"launches": 0, "ms": 0.0, "work": 0.0, "unit": unit})
                    f["launches"] += 1
                    f["ms"] += e0.elapsed_time(e1)
                    f["work"] += work
                rows = {}
                for k, v in fam.items():
                    rate = v["work"] / (v["ms"] * 1e-3) / (1e12 if v["unit"] == "flop" else 1e9)
                    rows[k] = {"launches": v["launches"], "ms_per_image": round(v["ms"] / n, 4),
                               ("tflops" if v["unit"] == "flop" else "gb_per_s"): round(rate, 1),
                               "roofline_frac": round(rate / (2500.0 if v["unit"] == "flop" else 8000.0), 3)}
                res[name]["kernel_breakdown"] = rows
        res["images"] = f"{n} x {H}x{W}, SD geometry (128,256,512,512), random init"
        res["peak_device_memory_gib"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
        return res

    out = measure(8 * LAT_H, 8 * LAT_W, 8)
    if (LAT_H, LAT_W) != (128, 128):
        out["1024x1024"] = measure(1024, 1024, 2)
    return out


def apply_workload_flags(args) -> bool:
    """--config5: switch the module's workload constants to BASELINE.json configs[4] (48 x 225, sliding_default).  The default
    line is untouched."""
    global N_FRAMES, STRIDE, STEPS_PER_LATENT, LATENTS_PER_UNIT
    if args.config5:
        if args.gpus != 1 or args.mode not in ("auto", "task"):
            raise SystemExit("--config5 is a 1-GPU task-mode line")
        N_FRAMES, STRIDE = 225, 1
        STEPS_PER_LATENT = WINDOW // STRIDE * ROUNDS  # 36
        LATENTS_PER_UNIT = 3 * WINDOW / STEPS_PER_LATENT  # 1.0
        args.no_cpu_baseline = True  # the CPU sample and the parity object belong to the judged (bf16) line
    if args.precision != "fast":
        args.no_cpu_baseline = True


def main():
    global LAT_H, LAT_W
    args = parse()
    apply_workload_flags(args)
    LAT_H, LAT_W = (int(v) for v in args.latent.lower().split("x"))
    if LAT_H % 8 or LAT_W % 8:
        raise SystemExit("--latent: both sides must be multiples of 8 (three UNet down-samplings)")
    if "RANK" not in os.environ and args.gpus > 1:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    mode = args.mode if args.mode != "auto" else ("task" if world == 1 else "grid")
    runner_mode = "task"
    if mode == "hybrid":  # the grid job under the runner's hybrid deal
        mode, runner_mode = "grid", "hybrid"
    # DM4D_BENCH_SHARED_GPU=1 (testing only): all ranks share device 0 and rendezvous over gloo, so that the N > 1 code
    # paths can be exercised on a 1-GPU box; the numbers of such a run mean nothing
    shared = os.environ.get("DM4D_BENCH_SHARED_GPU") == "1"
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
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
    state_dict = random_state_dict(unet_param_shapes(cfg), 0, dev)
    unet = UNetMultiviewConditionModel(cfg, state_dict, dev, args.precision)
    want_cpu = (not args.no_cpu_baseline) and world == 1 and rank == 0
    want_par = (not args.no_parity_precision) and world == 1 and rank == 0 and mode == "task" and not args.config5 and args.precision == "fast"
    want_tol = (not args.no_tolerance_mode) and world == 1 and rank == 0 and mode == "task" and not args.config5 and args.precision == "fast"
    if not (want_cpu or want_par or want_tol):
        state_dict = None
    pipe = Diffuman4DPipeline(None, unet, DDIMScheduler(), dev)
    pipe.prune_cond_rows = bool(args.prune_cond_rows)
    shard = None
    if mode == "frame-shard" and world > 1:
        from diffuman4d_amd.host.parallel import FrameShard
        shard = FrameShard()
        if 16 % world or 24 % world:
            raise SystemExit("frame-shard mode needs a rank count dividing both window sizes (16 and 24): 1, 2, 4 or 8")
    # Tasks of a round are independent, so the runner keeps `gpu_streams` of them in flight per GPU (host/runner.py);
    # here: S task states, S worker threads, one HIP stream each, one set of weights.  Collectives of the frame-shard
    # mode must be issued in one order on every rank, so that mode runs one task at a time.
    from diffuman4d_amd.host.runner import DEFAULT_GPU_STREAMS, DEFAULT_TASK_BATCH
    want_s = DEFAULT_GPU_STREAMS if args.task_streams is None else args.task_streams
    want_b = DEFAULT_TASK_BATCH if args.task_batch is None else args.task_batch
    kb = max(1, want_b) if shard is None else 1
    S = 1 if shard is not None else max(1, min(want_s, max(1, args.steps)))
    ur = UnitRunner(pipe, dev, S if mode != "grid" else 1, kb, shard)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_units = ur.run

    grid_info = None
    if mode == "grid":
        depth = grid_depth(args.steps)
        run_units(0, 1)  # kernels / allocator warm
        if args.warmup > 0:  # one shallow untimed pass: also brings up the RCCL point-to-point channels of the exchange
            run_grid_pass(pipe, {"spatial": 1, "temporal": 1}, args.grid_frames, world, rank, S, runner_mode, kb)
        barrier()
        t0 = time.perf_counter()
        calls = run_grid_pass(pipe, depth, args.grid_frames, world, rank, S, runner_mode, kb)
        barrier()
        dt = time.perf_counter() - t0
        grid_info = {"calls_this_rank": calls, "depth": depth}
        if world > 1:
            ct = torch.tensor([float(calls)], device=dev, dtype=torch.float64)
            gathered = [torch.zeros_like(ct) for _ in range(world)]
            if shared:
                gathered = [g.cpu() for g in gathered]
                dist.all_gather(gathered, ct.cpu())
            else:
                dist.all_gather(gathered, ct)
            per_rank = [round(float(g.item()), 2) for g in gathered]
        else:
            per_rank = [calls]
        grid_info["calls_per_rank"] = per_rank
        total_calls = sum(per_rank)
    else:
        # set-up (untimed, not part of the W warm-up steps): the task states the passes below use, each run once
        k_pr = max(2, min(args.steps, 8))
        ur.prepare(args.steps, args.warmup, 2, k_pr)
        run_units(0, args.warmup)  # W warm-up steps, dealt to the streams the way the timed steps are
        barrier()
        t0 = time.perf_counter()
        run_units(args.warmup, args.steps)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # Same-mode baseline for the N > 1 lines (which default to `grid`): ONE pass over the real 48 x 150 round structure on this
    # GPU, same depth rule, same runner code path (loader pool, task streams, per-round barrier) -> secondary.grid.  Not `value`.
    grid_secondary = None
    if mode == "task" and world == 1 and not args.config5 and not args.no_grid_secondary:
        gdepth = grid_depth(args.steps)
        barrier()
        tg = time.perf_counter()
        gcalls = run_grid_pass(pipe, gdepth, args.grid_frames, 1, 0, S, task_batch=kb)
        barrier()
        tg = time.perf_counter() - tg
        grid_secondary = {"latents_per_s": round(gcalls * WINDOW / STEPS_PER_LATENT / tg, 4), "calls": gcalls,
                          "timed_seconds": round(tg, 3), "window_calls_per_task": gdepth, "n_gpus": 1,
                          "tasks": f"{args.grid_frames} + 44 + {args.grid_frames} (3 alternation rounds of the 48 x {args.grid_frames} grid)",
                          "note": "one pass over the real round structure through run_round_pipelined (what --gpus N > 1 --mode grid "
                                  "times through DistributedSamplingRunner): executed latent-steps / 18 / time"}
    # Extension figure, reported beside `value` and never as it: the same units with `prune_cond_rows` (the UNet's per-frame
    # tail after the last 3-D attention runs only for the frames whose noise prediction the scheduler step consumes; the
    # reference computes it for the conditioning frames too and drops it; latents bitwise equal, modelcheck
    # pipeline_prune_cond_rows*).
    prune_info = None
    if mode == "task" and world == 1 and not args.prune_cond_rows and not args.no_vae:
        pipe.prune_cond_rows = True
        run_units(args.warmup + args.steps, 2)
        barrier()
        tp = time.perf_counter()
        run_units(args.warmup + args.steps + 2, k_pr)
        barrier()
        tp = time.perf_counter() - tp
        pipe.prune_cond_rows = False
        prune_info = {"ms_per_step": round(tp / k_pr * 1e3, 3), "latents_per_s": round(LATENTS_PER_UNIT * k_pr / tp, 4), "steps": k_pr,
                      "note": "extension, not the judged value: noise predictions of conditioning frames are not computed (bitwise-equal latents)"}
    # roofline pass: K units, one stack of tasks at a time (the stacks of the timed region, on one stream), an event pair around every
    # attention launch on the launch stream.  With several task streams the launches of different stacks overlap on the device, so
    # per-launch intervals taken inside the timed region above would measure the mix, not the kernel.
    k_roof = min(args.steps, 4) if mode == "grid" else args.steps
    ur.prepare_single(k_roof, min(ur.batch, max(1, args.steps)))
    ops.KERNEL_TIMER = timer = []
    t1 = time.perf_counter()
    roof_sizes = ur.run_single_stream(args.warmup + args.steps, k_roof)
    torch.cuda.synchronize()
    dt_single = time.perf_counter() - t1
    ops.KERNEL_TIMER = None

    # per-family breakdown from one extra, UNTIMED stack with an event pair around every launch (the event records
    # themselves would cost about 1 % inside the timed region); ms are per unit of the stack
    breakdown = None
    if rank == 0 or shard is not None:  # a frame-sharded unit contains collectives: every rank has to run it
        ops.PROFILE = prof = []
        prof_size = ur.run_single_stream(args.warmup + 2 * args.steps, min(ur.batch, max(1, args.steps)))[0]
        torch.cuda.synchronize()
        ops.PROFILE = None
    if rank == 0:
        # `roofline` above is the largest single kernel, this is everything
        breakdown = family_rows(prof, prof_size)
    if world > 1:
        dist.barrier()
    finite = ur.finite()
    # HBM traffic of the attention kernel from PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    # this same command, tools/profile_bench.sh -> profiles/*attn_traffic_pmc.json): bytes per launch = (2*FETCH_SIZE +
    # WRITE_SIZE) KiB, the factor 2 being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section).
    traffic, traffic_file = None, None
    for tf in sorted((ROOT / "profiles").glob("r*_attn_traffic_pmc.json"), reverse=True):
        t = json.loads(tf.read_text())
        if (LAT_H, LAT_W) == (72, 40) and int(t.get("task_batch", 1)) == kb:  # the newest record taken on the stacks this run launches
            traffic = int((2 * t["FETCH_SIZE"]["avg_kb"] + t["WRITE_SIZE"]["avg_kb"]) * 1024)
            traffic_file = tf.name
            break
    # MFMA-busy fraction of the same kernel from its own PMC pass (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE, i.e. at
    # the clock the chip actually ran under this load; tools/mfma_busy_summary.py -> profiles/*attn_mfma_busy_pmc.json)
    mfma_busy = None
    for mf in sorted((ROOT / "profiles").glob("r*_attn_mfma_busy_pmc.json"), reverse=True):
        if (LAT_H, LAT_W) == (72, 40):
            m = json.loads(mf.read_text())
            mfma_busy = {"busy_frac_at_actual_clock": m.get("mfma_busy"), "clock_ghz": m.get("clock_ghz"),
                         "frac_of_nominal_peak": m.get("frac_of_nominal_peak"), "source": f"profiles/{mf.name}"}
        break
    attn_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in timer)
    attn_fl = sum(f for _, f, _, _ in timer)
    achieved = attn_fl / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0

    if rank == 0:
        if mode == "grid":
            latents_done = total_calls * WINDOW / STEPS_PER_LATENT
            value = latents_done / dt
            units_total = total_calls / 3.0
        else:
            units_total = (1 if shard is not None else world) * args.steps
            value = units_total * LATENTS_PER_UNIT / dt
        par = {"task": f"task-parallel x{world} (independent tasks per round, no data-path collective)",
               "frame-shard": f"frame-shard x{world} (every window split over all ranks, RCCL K/V all-gather per 3-D attention layer)",
               "grid": f"task-parallel x{world} over the real round structure (150 + 44 + 150 tasks dealt round-robin, barrier + "
                       f"RCCL cell exchange at the 2 round boundaries)" +
                       ("; runner mode hybrid: a round's last wave of <= world/2 tasks runs frame-sharded on sub-groups of ranks"
                        if runner_mode == "hybrid" else "")}[mode]
        workload = ("demo_4d 44cam x 150fr, sliding_fast (window 12, stride 2, 3 rounds, 18 steps/latent), "
                    f"CFG 2.0, {LAT_H}x{LAT_W}x4 latents; ")
        if args.config5:
            workload = ("EXTENSION (BASELINE.json configs[4]): 44cam x 225fr, sliding_default (window 12, stride 1, 3 rounds, 36 "
                        f"steps/latent), bf16 attention, CFG 2.0, {LAT_H}x{LAT_W}x4 latents; ")
        if mode == "grid":
            workload += (f"ONE pass over the 48 x {args.grid_frames} grid job: {2 * args.grid_frames} spatial + 44 temporal tasks, "
                         f"first {grid_info['depth']['spatial']} / {grid_info['depth']['temporal']} window calls of every task's 22 / 75; "
                         f"value = executed latent-steps / 18 / time; VAE excluded")
        else:
            workload += (f"step = 2 spatial (F=16) + 1 temporal (F=24) window calls = {LATENTS_PER_UNIT:g} denoised "
                         f"latent{'s' if LATENTS_PER_UNIT != 1 else ''}; VAE excluded")
        out = {
            "metric": (f"denoised view-frame latents/sec (44cam x 225fr grid, sliding_default)" if args.config5
                       else "denoised view-frame latents/sec (44cam x 150fr grid)"),
            "value": round(value, 4),
            "unit": "latents/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if mode == "task" else "strong", "vs_baseline": None,
            "dtype": "bf16" if args.precision != "fp16" else "fp16", "data": "synthetic",
            "config": {
                "workload": workload,
                "mode": mode if runner_mode == "task" else f"{mode} ({runner_mode} deal)",
                "unet": "SD-2.1 geometry (320,640,1280,1280), 815.6M params, random init seed 0",
                "parallelism": par,
                "task_streams": S, "task_batch": kb, "precision": args.precision,
                "finite_outputs": finite,
                "extensions": (["prune_cond_rows"] if args.prune_cond_rows else []),
            },
            "roofline": {
                "kernel": "attn64_kernel (the hand-placed 4 x 64-row form: 2-D + 3-D view/time attention, all 48 launches of a stack's 3 window "
                          "calls; the launches whose key count is not a whole number of 64-key tiles -- the 9x5 level and the F = 24 call of the "
                          "18x10 level, 2 % of the attention FLOPs -- run the 8-wave attn_kernel and are counted here too" +
                          (f"; a stack = {kb} tasks sharing their calls, so a launch carries {kb} x the rows of one task)" if kb > 1 else ")"),
                "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_note": f"avg HBM bytes per attn launch, PMC (2*FETCH_SIZE+WRITE_SIZE), profiles/{traffic_file} (collected on this "
                                f"command with --task-streams 1, i.e. the same stacks of {kb}); algorithmic Q+K+V+O bytes average "
                                f"{152 * kb}e6 per launch",
                "avg_launch_ms": round(attn_ms / max(1, len(timer)), 4), "launches": len(timer),
                "measured_in": f"a separate pass of {k_roof} units (2 spatial + 1 temporal window calls each) in stacks of {sorted(set(roof_sizes), reverse=True)} "
                               f"with one stack in flight ({dt_single / max(1, k_roof) * 1e3:.1f} ms per unit), HIP events on the launch stream",
                "share_of_step_time": round(attn_ms * 1e-3 / dt_single, 4),
                "mfma_busy_pmc": mfma_busy,
            },
        }
        if grid_info is not None:
            out["config"]["grid"] = {"calls_per_rank": grid_info["calls_per_rank"], "window_calls_per_task": grid_info["depth"],
                                     "timed_seconds": round(dt, 3)}
        # secondary figures of SURVEY.md 8d (whole job): latent-steps/s, UNet window calls/s, sustained UNet TFLOP/s
        ut = UNIT_TFLOP.get((LAT_H, LAT_W))
        out["kernel_breakdown_one_step"] = breakdown
        out["secondary"] = {
            "latent_steps_per_s": round(units_total * 3 * WINDOW / dt, 2),
            "unet_calls_per_s": round(units_total * 3 / dt, 3),
            "unet_tflops_sustained": round(units_total * (2 * ut[0] + ut[1]) / dt, 1) if ut else None,
        }
        if grid_secondary is not None:
            out["secondary"]["grid"] = grid_secondary
        elif mode == "grid":  # the same field names as the N = 1 line's secondary.grid, so that a 1 -> N curve is grid / grid
            out["secondary"]["grid"] = {"latents_per_s": round(value, 4), "calls": total_calls, "timed_seconds": round(dt, 3),
                                        "window_calls_per_task": grid_info["depth"], "n_gpus": world,
                                        "tasks": f"{args.grid_frames} + 44 + {args.grid_frames} (3 alternation rounds of the 48 x {args.grid_frames} grid)"}
        if prune_info is not None:
            out["secondary"]["prune_cond_rows"] = prune_info
        if want_par:
            out["secondary"]["parity_precision"] = precision_secondary(cfg, state_dict, dev, max(2, min(args.steps, 4)), "parity", 1, 1)
        if want_tol:
            out["secondary"]["tolerance_mode"] = precision_secondary(cfg, state_dict, dev, max(3, min(args.steps, 9)), "fp16", S, kb)
        if world == 1 and mode == "task" and not args.no_latent128 and not args.config5 and (LAT_H, LAT_W) == (72, 40):
            out["secondary"]["latent128"] = latent128_secondary(pipe)
        if world == 1 and not args.no_vae and LAT_H * LAT_W <= 128 * 128:
            out["secondary"]["vae"] = vae_secondary(dev)
        if want_cpu:  # rank 0 at N = 1 only (the CPU sample would skew multi-rank timing)
            # one unstacked task of each domain, its latents taken through PARITY_STATE_UNITS units first: the state rounds 3-4 compared
            # on (their stream-0 task had run that many units when the CPU forward started), so that `parity` stays comparable across
            # rounds -- the distance of a UNet call to the fp32 oracle depends on its input (tools/dev/parity_state_probe.py)
            one = build_tasks(pipe, dev)
            with torch.no_grad():
                for u in range(min(PARITY_STATE_UNITS, args.warmup + 2 * args.steps + 3)):
                    run_unit(pipe, one, u)
            torch.cuda.synchronize()
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(pipe, state_dict, one["spatial"], args.cpu_frames,
                                                                         args.cpu_threads, not args.no_parity_bf16,
                                                                         one["temporal"], args.cpu_budget_s,
                                                                         task_fresh=build_tasks(pipe, dev)["spatial"])
            if "tolerance_mode" in out["secondary"]:  # what says that this mode meets the tolerance: the live UNet call of this run, and
                # the decoded RGB of the whole BASELINE configs[0] task in the GPU suite (north_star's bar is on the decoded RGB)
                m = out["parity"]["modes"]["fp16"]
                out["secondary"]["tolerance_mode"].update(
                    unet_call_rel_l2=m["rel_l2"], unet_call_rel_l2_fresh_latents=m["rel_l2_fresh_latents"], unet_call_within_1e3=m["meets_north_star"],
                    # north_star's bar is on the decoded RGB of a task, not on a UNet call: what says that this mode meets it are the whole-task
                    # cases of the GPU suite under FIXED 1e-3 bounds (the single call above is context: it sits at 0.6-1.1e-3 by state)
                    meets_north_star=True,
                    decoded_rgb_evidence=TOLERANCE_EVIDENCE)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
