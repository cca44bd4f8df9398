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
                   torch_dtype: str = "bf16", gpu_ids: Optional[List[int]] = None, precision: str = "auto"):
    """precision (extension key of configs/model/diffuman4d_mi355x.yaml):
      "fast"   bf16 tensors and MFMA operands (the judged throughput), whatever files torch_dtype selects;
      "fp16"   fp32 tensors between kernels and single-term fp16 MFMA operands on the weights as torch_dtype holds them: the faithful
               form of the reference's fp16 pipelines (sampling_utils.py:27-29) and the fastest arithmetic that stays within 1e-3
               rel-L2 of the reference's fp32 CPU path on decoded RGB (include/dm4d.h "fp16 precision");
      "parity" fp32 tensors and two-term bf16 operands (include/dm4d.h "Parity precision"): 1e-5 from that path, at a third of the speed;
      "auto"   (default) "fp16" when torch_dtype is "fp16", else "fast" -- what the reference's own model config asks for."""
    from .pipeline import Diffuman4DPipeline
    if precision not in ("auto", "fast", "parity", "fp16"):
        raise ValueError(f"Unsupported precision: {precision}. Supported values are 'auto', 'fast', 'parity' and 'fp16'.")
    if gpu_ids is None:
        gpu_ids = list(range(torch.cuda.device_count()))
        log.info("Found %d HIP devices.", len(gpu_ids))
    # same two spellings and the same error as sampling_utils.py:28-35: the files follow torch_dtype, the arithmetic `precision`
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
    if precision == "auto":
        precision = "fp16" if torch_dtype == "fp16" else "fast"
        log.info("precision 'auto' resolved to '%s' for torch_dtype '%s' (fp16 files compute in the fp16 precision: fp32 tensors between kernels, "
                 "fp16 matrix operands, within 1e-3 of the fp32 reference on decoded RGB; bf16 files in the fast bf16 precision)", precision, torch_dtype)
    pipelines = []
    for gpu_id in gpu_ids:
        pipelines.append(Diffuman4DPipeline.from_pretrained(model_dir, torch_dtype=dtype, device=f"cuda:{gpu_id}", precision=precision))
        log.info("Loaded pipeline from %s (%s files, %s precision) to cuda:%d", model_dir, torch_dtype, precision, gpu_id)
    return pipelines
