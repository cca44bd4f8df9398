"""Scheduler bookkeeping on the host (integer timesteps + fp32 coefficient rows).

The reference takes whatever scheduler the checkpoint's ``scheduler/scheduler_config.json`` names
(pipeline_diffuman4d.py:28,134,265-271) and deep-copies it per latent because some Karras-family
schedulers are stateful.  Its loop calls ``scale_model_input`` with a VECTOR of per-frame timesteps (:376), which only
identity implementations survive, so the class has to be one of DDIM / DDPM / PNDM / DPM-Solver / UniPC / DEIS.  Built here:

* ``DDIMScheduler`` (eta = 0): stateless; one table serves every latent and the per-latent ``.step()`` Python loop
  (:413-422) becomes one batched device kernel fed with the coefficient rows computed here.
* ``DPMSolverMultistepScheduler`` (DPM-Solver++, orders 1 and 2): every update of it is LINEAR in (sample, model output,
  previous x0 prediction), and which update a latent gets -- first order on its first step of a call (the reference makes
  fresh scheduler copies per ``sliding_iterative_denoise`` call, :500-501) and on the final step, second order otherwise --
  depends only on the planned timestep indices.  So the whole stateful object collapses to rows
  ``x' = a x + b m + c p,  p' = d x + e m`` (p = the latent's stored x0 prediction) consumed by
  ``dm4d_cfg_linear_step_bf16``; the only state is one bf16 tensor the shape of the task's latents.

Anything else raises in ``load_scheduler`` instead of guessing (stochastic samplers need the reference's RNG stream, PNDM
evaluates the model twice on its first step).
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
    rescale_betas_zero_snr: bool = False
    thresholding: bool = False

    @classmethod
    def from_dict(cls, d: Dict) -> "DDIMConfig":
        names = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})


def rescale_zero_terminal_snr(betas):
    """Betas with zero terminal SNR (Lin et al. 2023, Algorithm 1; diffusers' `rescale_betas_zero_snr`): checkpoints fine-tuned
    that way (v-prediction, trailing spacing) reach alpha_cumprod = 0 at the last training step."""
    import torch
    alphas_bar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    a0, aT = alphas_bar_sqrt[0].clone(), alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt = (alphas_bar_sqrt - aT) * (a0 / (a0 - aT))
    alphas_bar = alphas_bar_sqrt**2
    alphas = torch.cat([alphas_bar[0:1], alphas_bar[1:] / alphas_bar[:-1]])
    return 1 - alphas


class DDIMScheduler:
    init_noise_sigma = 1.0

    def __init__(self, config: DDIMConfig = DDIMConfig()):
        self.config = c = config
        if c.clip_sample or c.thresholding:
            raise NotImplementedError("DDIMScheduler: clip_sample / thresholding are not supported (the step kernel does not clamp x0)")
        n = c.num_train_timesteps
        # built with the same torch fp32 ops as diffusers' DDIMScheduler so the table is bit-identical
        import torch
        if c.beta_schedule == "scaled_linear":
            betas = torch.linspace(c.beta_start**0.5, c.beta_end**0.5, n, dtype=torch.float32) ** 2
        elif c.beta_schedule == "linear":
            betas = torch.linspace(c.beta_start, c.beta_end, n, dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {c.beta_schedule}")
        if c.rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).numpy()
        self.final_alpha_cumprod = np.float32(1.0) if c.set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None

    is_multistep = False

    @classmethod
    def from_pretrained(cls, path):
        """The scheduler the checkpoint names (kept under this name for callers that predate `load_scheduler`)."""
        return load_scheduler(path)

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
        elif c.timestep_spacing == "linspace":
            ts = np.linspace(0, c.num_train_timesteps - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
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
    thresholding: bool = False
    use_karras_sigmas: bool = False
    use_exponential_sigmas: bool = False
    use_beta_sigmas: bool = False
    use_lu_lambdas: bool = False
    use_flow_sigmas: bool = False
    variance_type: object = None
    # keys that change the sigma table: carried so that a checkpoint which sets them fails loudly instead of getting wrong coefficients
    rescale_betas_zero_snr: bool = False
    trained_betas: object = None
    lambda_min_clipped: float = -float("inf")

    @classmethod
    def from_dict(cls, d: Dict) -> "DPMSolverConfig":
        names = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})


class DPMSolverMultistepScheduler:
    """DPM-Solver++ (multistep, orders 1 and 2) as coefficient rows; see the module docstring.  Restated from the published
    algorithm and diffusers 0.33.1's step order (oracle/dpmsolver.py is the stateful form the tests compare against)."""

    init_noise_sigma = 1.0
    is_multistep = True
    ROW = 8  # floats per coefficient row: a, b, c, d, e, 3 x padding

    def __init__(self, config: DPMSolverConfig = DPMSolverConfig()):
        self.config = c = config
        unsupported = [k for k in ("thresholding", "use_karras_sigmas", "use_exponential_sigmas", "use_beta_sigmas", "use_lu_lambdas",
                                   "use_flow_sigmas", "rescale_betas_zero_snr") if getattr(c, k)]
        if c.trained_betas is not None:
            unsupported.append("trained_betas")
        if c.lambda_min_clipped is not None and c.lambda_min_clipped > -1e30:
            unsupported.append("lambda_min_clipped")
        if unsupported or c.algorithm_type != "dpmsolver++" or c.solver_order not in (1, 2) or \
                c.solver_type not in ("midpoint", "heun") or c.variance_type not in (None, "fixed_small", "fixed_large") or \
                c.prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError(
                f"DPMSolverMultistepScheduler: only deterministic dpmsolver++ of order 1 / 2 with plain sigmas is implemented "
                f"(got algorithm_type={c.algorithm_type!r}, solver_order={c.solver_order}, solver_type={c.solver_type!r}, "
                f"prediction_type={c.prediction_type!r}, switched on: {unsupported})")
        import torch
        n = c.num_train_timesteps
        if c.beta_schedule == "scaled_linear":
            betas = torch.linspace(c.beta_start**0.5, c.beta_end**0.5, n, dtype=torch.float32) ** 2
        elif c.beta_schedule == "linear":
            betas = torch.linspace(c.beta_start, c.beta_end, n, dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {c.beta_schedule}")
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).numpy()
        self.num_inference_steps = None
        self.timesteps = None
        self.sigmas = None

    def set_timesteps(self, num_inference_steps: int) -> np.ndarray:
        c, n = self.config, num_inference_steps
        if n > c.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        last = c.num_train_timesteps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, last - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ts = (np.arange(0, n + 1) * (last // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            ts = np.arange(last, 0, -c.num_train_timesteps / n).round().copy().astype(np.int64) - 1
        else:
            raise NotImplementedError(f"timestep_spacing {c.timestep_spacing}")
        if len(np.unique(ts)) != len(ts):
            # the reference finds a latent's step by searching its timestep VALUE (index_for_timestep) and takes the second of
            # two equal entries; a plan of indices cannot reproduce that quirk
            raise NotImplementedError("duplicate timesteps in the schedule (num_inference_steps too close to num_train_timesteps)")
        ac = self.alphas_cumprod
        sig = np.interp(ts, np.arange(0, len(ac)), np.array(((1 - ac) / ac) ** 0.5))
        if c.final_sigmas_type == "sigma_min":
            last_sigma = ((1 - ac[0]) / ac[0]) ** 0.5
        elif c.final_sigmas_type == "zero":
            last_sigma = 0.0
        else:
            raise NotImplementedError(f"final_sigmas_type {c.final_sigmas_type}")
        self.sigmas = np.concatenate([sig, [last_sigma]]).astype(np.float32)  # fp32 table, as the reference keeps it
        self.timesteps = ts
        self.num_inference_steps = n
        return ts

    def step_rows(self, step_index: np.ndarray, has_prev: np.ndarray) -> np.ndarray:
        """[..., 8] fp32 rows (a, b, c, d, e, 0, 0, 0) for latents at `step_index` (index into timesteps) whose previous x0
        prediction is (has_prev) or is not available in this call:  x' = a x + b m + c p;  p' = d x + e m."""
        c, n = self.config, self.num_inference_steps
        idx = np.asarray(step_index, dtype=np.int64)
        hp = np.asarray(has_prev, dtype=bool)
        s = self.sigmas.astype(np.float64)

        def alpha_sigma(sig):
            al = 1.0 / np.sqrt(sig * sig + 1.0)
            return al, sig * al

        with np.errstate(divide="ignore", invalid="ignore"):
            al_t, sg_t = alpha_sigma(s[idx + 1])
            al_s, sg_s = alpha_sigma(s[idx])
            lam_t, lam_s = np.log(al_t) - np.log(sg_t), np.log(al_s) - np.log(sg_s)  # lam_t = +inf at a final sigma of zero
            h = lam_t - lam_s
            K = al_t * (np.exp(-h) - 1.0)
            al_p, sg_p = alpha_sigma(s[np.maximum(idx - 1, 0)])
            r0 = (lam_s - (np.log(al_p) - np.log(sg_p))) / h
            if c.solver_type == "midpoint":
                q = 0.5 / r0                    # weight of (x0 - p) / r0 in D: x' = ratio x - K (x0 + q (x0 - p))
                w_x0, w_p = -K * (1.0 + q), K * q
            else:                               # heun: x' = ratio x - K x0 + al_t ((exp(-h) - 1) / h + 1) (x0 - p) / r0
                g = al_t * ((np.exp(-h) - 1.0) / h + 1.0) / r0
                w_x0, w_p = -K + g, -g
        final = idx == n - 1
        lower_final = final & (c.euler_at_final or (c.lower_order_final and n < 15) or c.final_sigmas_type == "zero")
        first = (c.solver_order == 1) | (~hp) | lower_final
        w_x0 = np.where(first, -K, w_x0)
        w_p = np.where(first, 0.0, w_p)
        if c.prediction_type == "epsilon":      # x0 = (x - sigma m) / alpha
            d, e = 1.0 / al_s, -sg_s / al_s
        else:                                   # v_prediction: x0 = alpha x - sigma m
            d, e = al_s, -sg_s
        ratio = sg_t / sg_s
        rows = np.zeros(idx.shape + (self.ROW,), dtype=np.float64)
        rows[..., 0] = ratio + w_x0 * d
        rows[..., 1] = w_x0 * e
        rows[..., 2] = w_p
        rows[..., 3] = d
        rows[..., 4] = e
        return rows.astype(np.float32)


def load_scheduler(path):
    """`scheduler/scheduler_config.json` of a diffusers checkpoint -> the scheduler object of this package."""
    cfg = json.loads((Path(path) / "scheduler_config.json").read_text())
    name = cfg.get("_class_name", "DDIMScheduler")
    if name == "DDIMScheduler":
        return DDIMScheduler(DDIMConfig.from_dict(cfg))
    if name == "DPMSolverMultistepScheduler":
        return DPMSolverMultistepScheduler(DPMSolverConfig.from_dict(cfg))
    raise NotImplementedError(
        f"scheduler {name}: DDIMScheduler and DPMSolverMultistepScheduler (dpmsolver++) are implemented.  The reference's loop "
        f"passes a vector of per-frame timesteps to scale_model_input (pipeline_diffuman4d.py:376), so Euler / Heun / LMS cannot "
        f"be what a working checkpoint names; DDPM and the ancestral / SDE samplers draw noise from the reference's RNG stream; "
        f"PNDM evaluates the model twice on its first step (SURVEY.md D7)")
