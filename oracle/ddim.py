"""Oracle DDIM scheduler (diffusers==0.33.1 DDIMScheduler, eta=0).  TEST INFRASTRUCTURE ONLY.

The reference does not pin the scheduler class; DDIM with the SD-2.1 scheduler config
(scaled_linear betas 0.00085..0.012, 1000 train steps, "leading" spacing, steps_offset=1,
clip_sample=False, set_alpha_to_one=False) is the working assumption.  Call sites:
``pipeline_diffuman4d.py:190`` (init_noise_sigma), ``:268-270`` (set_timesteps/timesteps),
``:376`` (scale_model_input = identity), ``:420`` (step).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class DDIMConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_schedule: str = "scaled_linear"
    steps_offset: int = 1
    set_alpha_to_one: bool = False
    prediction_type: str = "epsilon"  # or "v_prediction"
    timestep_spacing: str = "leading"
    rescale_betas_zero_snr: bool = False


class DDIMScheduler:
    init_noise_sigma = 1.0

    def __init__(self, cfg: DDIMConfig = DDIMConfig()):
        self.cfg = cfg
        n = cfg.num_train_timesteps
        if cfg.beta_schedule == "scaled_linear":
            betas = torch.linspace(cfg.beta_start**0.5, cfg.beta_end**0.5, n, dtype=torch.float32) ** 2
        elif cfg.beta_schedule == "linear":
            betas = torch.linspace(cfg.beta_start, cfg.beta_end, n, dtype=torch.float32)
        else:
            raise NotImplementedError(cfg.beta_schedule)
        if cfg.rescale_betas_zero_snr:  # diffusers rescale_zero_terminal_snr
            abs_ = torch.cumprod(1.0 - betas, dim=0).sqrt()
            a0, aT = abs_[0].clone(), abs_[-1].clone()
            abs_ = (abs_ - aT) * (a0 / (a0 - aT))
            ab = abs_**2
            betas = 1 - torch.cat([ab[0:1], ab[1:] / ab[:-1]])
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg.set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int):
        cfg = self.cfg
        if num_inference_steps > cfg.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        if cfg.timestep_spacing == "leading":
            ratio = cfg.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += cfg.steps_offset
        elif cfg.timestep_spacing == "trailing":
            ratio = cfg.num_train_timesteps / num_inference_steps
            ts = np.round(np.arange(cfg.num_train_timesteps, 0, -ratio)).astype(np.int64) - 1
        elif cfg.timestep_spacing == "linspace":
            ts = np.linspace(0, cfg.num_train_timesteps - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise NotImplementedError(cfg.timestep_spacing)
        self.timesteps = torch.from_numpy(ts)
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample  # identity => aliases (pipeline_diffuman4d.py:375-379)

    def coefficients(self, t: int):
        """(alpha_prod_t, alpha_prod_t_prev) as python floats of the fp32 table entries."""
        prev_t = t - self.cfg.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step(self, model_output: torch.Tensor, t: int, sample: torch.Tensor) -> torch.Tensor:
        a_t, a_prev = self.coefficients(int(t))
        b_t = 1 - a_t
        # 0-dim fp32 "scalars" do not promote a bf16 tensor: arithmetic stays in the tensor dtype
        if self.cfg.prediction_type == "epsilon":
            x0 = (sample - b_t**0.5 * model_output) / a_t**0.5
            eps = model_output
        elif self.cfg.prediction_type == "v_prediction":
            x0 = (a_t**0.5) * sample - (b_t**0.5) * model_output
            eps = (a_t**0.5) * model_output + (b_t**0.5) * sample
        else:
            raise NotImplementedError(self.cfg.prediction_type)
        direction = (1 - a_prev) ** 0.5 * eps
        return a_prev**0.5 * x0 + direction
