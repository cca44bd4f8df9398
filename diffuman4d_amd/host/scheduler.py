"""DDIM bookkeeping on the host (integer timesteps + fp32 alpha tables).

The reference takes whatever scheduler the checkpoint's ``scheduler/scheduler_config.json`` names
(pipeline_diffuman4d.py:28,134,265-271) and deep-copies it per latent because some Karras-family
schedulers are stateful.  DDIM (eta = 0) is stateless, so one table serves every latent and the
per-latent ``.step()`` Python loop (:413-422) becomes one batched device kernel fed with the
coefficient rows computed here.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, fields
from pathlib import Path
from typing import Dict

import numpy as np


@dataclass
class DDIMConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_schedule: str = "scaled_linear"
    steps_offset: int = 1
    set_alpha_to_one: bool = False
    prediction_type: str = "epsilon"
    timestep_spacing: str = "leading"
    clip_sample: bool = False

    @classmethod
    def from_dict(cls, d: Dict) -> "DDIMConfig":
        names = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})


class DDIMScheduler:
    init_noise_sigma = 1.0

    def __init__(self, config: DDIMConfig = DDIMConfig()):
        self.config = c = config
        if c.clip_sample:
            raise NotImplementedError("clip_sample=True is not supported")
        n = c.num_train_timesteps
        # built with the same torch fp32 ops as diffusers' DDIMScheduler so the table is bit-identical
        import torch
        if c.beta_schedule == "scaled_linear":
            betas = torch.linspace(c.beta_start**0.5, c.beta_end**0.5, n, dtype=torch.float32) ** 2
        elif c.beta_schedule == "linear":
            betas = torch.linspace(c.beta_start, c.beta_end, n, dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {c.beta_schedule}")
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).numpy()
        self.final_alpha_cumprod = np.float32(1.0) if c.set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None

    @classmethod
    def from_pretrained(cls, path) -> "DDIMScheduler":
        cfg = json.loads((Path(path) / "scheduler_config.json").read_text())
        name = cfg.get("_class_name", "DDIMScheduler")
        if name != "DDIMScheduler":
            raise NotImplementedError(f"scheduler {name}: only DDIMScheduler is implemented (SURVEY.md D7)")
        return cls(DDIMConfig.from_dict(cfg))

    def set_timesteps(self, num_inference_steps: int) -> np.ndarray:
        c = self.config
        if num_inference_steps > c.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        if c.timestep_spacing == "leading":
            ratio = c.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            ratio = c.num_train_timesteps / num_inference_steps
            ts = np.round(np.arange(c.num_train_timesteps, 0, -ratio)).astype(np.int64) - 1
        else:
            raise NotImplementedError(f"timestep_spacing {c.timestep_spacing}")
        self.timesteps = ts
        return ts

    def step_coefficients(self, t: np.ndarray) -> np.ndarray:
        """[..., 4] fp32 rows {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)} for integer timesteps t."""
        t = np.asarray(t, dtype=np.int64)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = np.where(prev >= 0, self.alphas_cumprod[np.clip(prev, 0, None)], self.final_alpha_cumprod).astype(np.float32)
        out = np.stack([np.sqrt(a_t), np.sqrt(1 - a_t), np.sqrt(a_p), np.sqrt(1 - a_p)], axis=-1)
        return out.astype(np.float32)
