"""Oracle DPM-Solver++ multistep scheduler (diffusers==0.33.1 DPMSolverMultistepScheduler, algorithm_type "dpmsolver++",
orders 1 and 2, no Karras sigmas, no thresholding, no SDE variant).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: diffusers is an un-vendored dependency of the reference (requirements.txt:5) and is not installed
here; this file restates the published algorithm (Lu et al., "DPM-Solver++", and the 0.33.1 scheduler's step order:
convert_model_output in the sample dtype, history shift, first / second order update with the sample upcast to fp32, result
cast back to the model dtype).  The reference takes whatever scheduler the checkpoint names and deep-copies it per latent
because such schedulers carry state (pipeline_diffuman4d.py:265-271, 420, 500-501, 535); its loop calls
``scale_model_input`` with a VECTOR of per-frame timesteps (:376), which only identity implementations survive -- DDIM,
DDPM, PNDM, DPM-Solver, UniPC -- so the Euler / Heun / LMS family cannot be what the checkpoint ships.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class DPMSolverConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    beta_schedule: str = "scaled_linear"
    solver_order: int = 2
    prediction_type: str = "epsilon"
    algorithm_type: str = "dpmsolver++"
    solver_type: str = "midpoint"
    lower_order_final: bool = True
    euler_at_final: bool = False
    final_sigmas_type: str = "zero"
    timestep_spacing: str = "linspace"
    steps_offset: int = 0


class DPMSolverMultistepScheduler:
    init_noise_sigma = 1.0

    def __init__(self, cfg: DPMSolverConfig = DPMSolverConfig()):
        self.cfg = cfg
        n = cfg.num_train_timesteps
        if cfg.beta_schedule == "scaled_linear":
            betas = torch.linspace(cfg.beta_start**0.5, cfg.beta_end**0.5, n, dtype=torch.float32) ** 2
        elif cfg.beta_schedule == "linear":
            betas = torch.linspace(cfg.beta_start, cfg.beta_end, n, dtype=torch.float32)
        else:
            raise NotImplementedError(cfg.beta_schedule)
        if cfg.algorithm_type != "dpmsolver++" or cfg.solver_order not in (1, 2) or cfg.solver_type not in ("midpoint", "heun"):
            raise NotImplementedError((cfg.algorithm_type, cfg.solver_order, cfg.solver_type))
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.timesteps = None
        self.sigmas = None
        self.model_outputs = [None] * cfg.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    def set_timesteps(self, num_inference_steps: int):
        cfg, n = self.cfg, num_inference_steps
        last = cfg.num_train_timesteps  # lambda_min_clipped = -inf: nothing is clipped
        if cfg.timestep_spacing == "linspace":
            ts = np.linspace(0, last - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif cfg.timestep_spacing == "leading":
            ratio = last // (n + 1)
            ts = (np.arange(0, n + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + cfg.steps_offset
        elif cfg.timestep_spacing == "trailing":
            ratio = cfg.num_train_timesteps / n
            ts = np.arange(last, 0, -ratio).round().copy().astype(np.int64) - 1
        else:
            raise NotImplementedError(cfg.timestep_spacing)
        ac = self.alphas_cumprod.numpy()
        sig = np.array(((1 - ac) / ac) ** 0.5)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        if cfg.final_sigmas_type == "sigma_min":
            last_sigma = ((1 - ac[0]) / ac[0]) ** 0.5
        elif cfg.final_sigmas_type == "zero":
            last_sigma = 0.0
        else:
            raise NotImplementedError(cfg.final_sigmas_type)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [last_sigma]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * cfg.solver_order
        self.lower_order_nums = 0
        self._step_index = None
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma**2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def _init_step_index(self, timestep):
        cand = (self.timesteps == int(timestep)).nonzero()
        if len(cand) == 0:
            self._step_index = len(self.timesteps) - 1
        elif len(cand) > 1:
            self._step_index = int(cand[1])
        else:
            self._step_index = int(cand[0])

    def convert_model_output(self, model_output, sample):
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[self._step_index])
        if self.cfg.prediction_type == "epsilon":
            return (sample - sigma_t * model_output) / alpha_t
        if self.cfg.prediction_type == "v_prediction":
            return alpha_t * sample - sigma_t * model_output
        raise NotImplementedError(self.cfg.prediction_type)

    def _first_order(self, x0, sample):
        i = self._step_index
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i + 1])
        alpha_s, sigma_s = self._alpha_sigma(self.sigmas[i])
        h = (torch.log(alpha_t) - torch.log(sigma_t)) - (torch.log(alpha_s) - torch.log(sigma_s))
        return (sigma_t / sigma_s) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * x0

    def _second_order(self, sample):
        i = self._step_index
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i + 1])
        alpha_s0, sigma_s0 = self._alpha_sigma(self.sigmas[i])
        alpha_s1, sigma_s1 = self._alpha_sigma(self.sigmas[i - 1])
        lam_t = torch.log(alpha_t) - torch.log(sigma_t)
        lam_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        lam_s1 = torch.log(alpha_s1) - torch.log(sigma_s1)
        m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
        h, h_0 = lam_t - lam_s0, lam_s0 - lam_s1
        r0 = h_0 / h
        D0, D1 = m0, (1.0 / r0) * (m0 - m1)
        if self.cfg.solver_type == "midpoint":
            return (sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * D0 - 0.5 * (alpha_t * (torch.exp(-h) - 1.0)) * D1
        return (sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * D0 + (alpha_t * ((torch.exp(-h) - 1.0) / h + 1.0)) * D1

    def step(self, model_output: torch.Tensor, t: int, sample: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        if self._step_index is None:
            self._init_step_index(t)
        n = len(self.timesteps)
        lower_order_final = (self._step_index == n - 1) and (
            cfg.euler_at_final or (cfg.lower_order_final and n < 15) or cfg.final_sigmas_type == "zero")
        x0 = self.convert_model_output(model_output, sample)
        for k in range(cfg.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0
        sample32 = sample.to(torch.float32)  # "upcast to avoid precision issues when computing prev_sample"
        if cfg.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final:
            prev = self._first_order(x0, sample32)
        else:
            prev = self._second_order(sample32)
        if self.lower_order_nums < cfg.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return prev.to(model_output.dtype)
