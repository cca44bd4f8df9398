#!/usr/bin/env python
"""CLI entry point, same invocation grammar as the reference's ``inference.py``:

    python inference.py exp=demo_4d_tiny model=diffuman4d_mi355x data.scene_label=0023_06 data.data_dir=...

With hydra-core installed and ``--config-dir <reference>/configs`` this is the reference's own config
tree; otherwise the built-in composer (``diffuman4d_amd/host/config.py``) resolves the same groups.
Launch under ``torchrun --nproc-per-node N`` to run one process per GPU (RCCL grid exchange between
alternation rounds) instead of the reference's thread-per-GPU runner.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys

from diffuman4d_amd.host import config as cfglib

log = logging.getLogger("inference")


def inference(cfg: dict):
    import torch
    from diffuman4d_amd.host.runner import DistributedSamplingRunner, SamplingRunner

    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if distributed:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        cfg["model"]["gpu_ids"] = [local]
    log.info("Instantiating dataset <%s>", cfg["data"]["_target_"])
    dataset = cfglib.instantiate(cfg["data"])
    log.info("Instantiating pipelines <%s>", cfg["model"]["_target_"])
    pipelines = cfglib.instantiate(cfg["model"])
    log.info("Instantiating sampler <%s>", cfg["sampler"]["_target_"])
    sampler = cfglib.instantiate(cfg["sampler"], dataset=dataset, pipelines=pipelines)
    # runner.gpu_streams=N (default 3): stacks of tasks of a round in flight per GPU, one HIP stream each
    # runner.task_batch=K (default 2): K consecutive tasks of a round share their window calls (tensors stacked along the frame axis;
    # every task's result is bitwise what it is alone); gpu_streams=1 task_batch=1 = the reference's one-task-at-a-time order
    # runner.writer_processes=N (default 0): with sampler.device_results=true the JPEG / WebP encoding of every task's uint8
    # package runs in N writer processes (host/imgwrite.py) instead of the writer threads
    rk = {k: int(v) for k, v in (cfg.get("runner") or {}).items()
          if k in ("prefetch_depth", "writers", "gpu_streams", "writer_processes", "task_batch")}
    # runner.host_threads=N: torch's intra-op CPU threads for the host-side stages (loader, writer).  The 256-thread GPU hosts
    # default to 128-256 threads per op, and several loader / writer threads each fanning small tensor ops out over all of
    # them is what made the host stages the bottleneck (profiles/r02_e2e_demo.log)
    ht = (cfg.get("runner") or {}).get("host_threads")
    if ht:
        torch.set_num_threads(max(1, int(ht)))
    # runner.mode=task|frame-shard|hybrid (one process per GPU only): how a round's tasks meet the ranks -- one rank per task, all
    # ranks on every task (in-window frame sharding, RCCL K/V all-gather), or task-parallel waves with a frame-sharded tail
    mode = str((cfg.get("runner") or {}).get("mode", "task"))
    if not distributed and mode != "task":
        raise ValueError(f"runner.mode={mode} needs one process per GPU (torchrun --nproc-per-node N)")
    runner = DistributedSamplingRunner(sampler, mode=mode, **rk) if distributed else SamplingRunner(sampler, **rk)
    if cfg.get("sampling", True):
        runner.inference()
    if cfg.get("to_nerfstudio") or cfg.get("evaluating"):
        log.warning("to_nerfstudio / evaluating are post-processing steps of the reference that are out of scope here")


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(name)s][%(levelname)s] %(message)s")
    ap = argparse.ArgumentParser(add_help=True)
    ap.add_argument("--config-dir", default=None, help="a Hydra-style configs/ directory (e.g. the reference's)")
    ap.add_argument("overrides", nargs="*", help="Hydra-style overrides: exp=demo_4d sampler.window_size=4 ...")
    args = ap.parse_args(argv)
    cfg = cfglib.compose(args.overrides, args.config_dir)
    inference(cfg)


if __name__ == "__main__":
    main(sys.argv[1:])
