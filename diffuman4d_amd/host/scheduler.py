"""Scheduler bookkeeping on the host (integer timesteps + fp32 coefficient rows).

The reference takes whatever scheduler the checkpoint's ``scheduler/scheduler_config.json`` names
(pipeline_diffuman4d.py:28,134,265-271) and deep-copies it per latent because some Karras-family
schedulers are stateful.  Its loop calls ``scale_model_input`` with a VECTOR of per-frame timesteps (:376), which only
identity implementations survive, so the class has to be one of DDIM / DDPM / PNDM / DPM-Solver / UniPC / DEIS.  Built here (all but DDPM):

* ``DDIMScheduler`` (eta = 0): stateless; one table serves every latent and the per-latent ``.step()`` Python loop
  (:413-422) becomes one batched device kernel fed with the coefficient rows computed here.
* ``UniPCMultistepScheduler`` (orders 1 and 2, bh1 / bh2, with its corrector) and ``DEISMultistepScheduler`` (orders 1-3): the same
  idea with up to three stored tensors per latent and 16-float rows (see the block comment above ``UniPCConfig``).
* ``DPMSolverMultistepScheduler`` (DPM-Solver++, orders 1 and 2): every update of it is LINEAR in (sample, model output,
  previous x0 prediction), and which update a latent gets -- first order on its first step of a call (the reference makes
  fresh scheduler copies per ``sliding_iterative_denoise`` call, :500-501) and on the final step, second order otherwise --
  depends only on the planned timestep indices.  So the whole stateful object collapses to rows
  ``x' = a x + b m + c p,  p' = d x + e m`` (p = the latent's stored x0 prediction) consumed by
  ``dm4d_cfg_linear_step_bf16``; the only state is one bf16 tensor the shape of the task's latents.

* ``PNDMScheduler`` with ``skip_prk_steps`` (PLMS, the Stable Diffusion family's stock scheduler): Adams-Bashforth combinations of up to
  four model outputs and the repeated second step, as rows of the same kernel (see the class).

Anything else raises in ``load_scheduler`` instead of guessing (stochastic samplers need the reference's RNG stream).
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


# ---- UniPC and DEIS: linear multistep solvers with up to three stored tensors per latent ---------------------------------------------
# One update of either is linear in (sample x, model output m, stored tensors s1, s2, s3), so the stateful object the reference keeps
# per latent (pipeline_diffuman4d.py:265-271, 420, 500-501) collapses to one 16-float row per (call, frame), consumed by
# dm4d_cfg_multistep_step_bf16:
#     conv = k0 x + k1 m                                   the converted model output (x0 prediction for UniPC, noise form for DEIS)
#     xc   = k2 x + k3 s3 + k4 s1 + k5 s2 + k6 conv        UniPC's corrector (the sample of the PREVIOUS predictor step, refined with
#                                                          this step's prediction); xc = x (k2 = 1) when there is nothing to correct
#     x'   = k7 xc + k8 conv + k9 s1 + k10 s2              the predictor
#     s3' = xc,  s2' = s1,  s1' = conv                     (s1 / s2 = the last two converted outputs, s3 = UniPC's last_sample)
# Which formula a latent gets depends on its step index and on how many steps it has taken IN THIS CALL (the reference makes fresh
# scheduler copies per sliding_iterative_denoise call): both known from the plan (schedule.history_counts).
ROW16 = 16


def _sigma_tables(c, n):
    """(timesteps, alphas_cumprod-derived sigma at each timestep) shared by the DPM-Solver family of diffusers 0.33.1."""
    last = c.num_train_timesteps
    if n > last:
        raise ValueError("num_inference_steps > num_train_timesteps")
    if c.timestep_spacing == "linspace":
        ts = np.linspace(0, last - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
    elif c.timestep_spacing == "leading":
        ts = (np.arange(0, n + 1) * (last // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + c.steps_offset
    elif c.timestep_spacing == "trailing":
        ts = np.arange(last, 0, -c.num_train_timesteps / n).round().copy().astype(np.int64) - 1
    else:
        raise NotImplementedError(f"timestep_spacing {c.timestep_spacing}")
    if len(np.unique(ts)) != len(ts):
        raise NotImplementedError("duplicate timesteps in the schedule (num_inference_steps too close to num_train_timesteps)")
    return ts


def _alphas_cumprod(c):
    import torch
    n = c.num_train_timesteps
    if c.beta_schedule == "scaled_linear":
        betas = torch.linspace(c.beta_start**0.5, c.beta_end**0.5, n, dtype=torch.float32) ** 2
    elif c.beta_schedule == "linear":
        betas = torch.linspace(c.beta_start, c.beta_end, n, dtype=torch.float32)
    else:
        raise NotImplementedError(f"beta_schedule {c.beta_schedule}")
    return torch.cumprod(1.0 - betas, dim=0).numpy()


def _alpha_sigma64(sig):
    al = 1.0 / np.sqrt(sig * sig + 1.0)
    return al, sig * al


_SIGMA_SWITCHES = ("thresholding", "use_karras_sigmas", "use_exponential_sigmas", "use_beta_sigmas", "use_flow_sigmas", "rescale_betas_zero_snr")


@dataclass
class UniPCConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.0001
    beta_end: float = 0.02
    beta_schedule: str = "linear"
    solver_order: int = 2
    prediction_type: str = "epsilon"
    predict_x0: bool = True
    solver_type: str = "bh2"
    lower_order_final: bool = True
    disable_corrector: tuple = ()
    timestep_spacing: str = "linspace"
    steps_offset: int = 0
    final_sigmas_type: str = "zero"
    thresholding: bool = False
    use_karras_sigmas: bool = False
    use_exponential_sigmas: bool = False
    use_beta_sigmas: bool = False
    use_flow_sigmas: bool = False
    rescale_betas_zero_snr: bool = False
    trained_betas: object = None
    solver_p: object = None

    @classmethod
    def from_dict(cls, d: Dict) -> "UniPCConfig":
        names = {f.name for f in fields(cls)}
        kw = {k: v for k, v in d.items() if k in names}
        if "disable_corrector" in kw:
            kw["disable_corrector"] = tuple(kw["disable_corrector"] or ())
        return cls(**kw)


class _MultistepRows:
    """Common part of the row-planned multistep schedulers."""
    init_noise_sigma = 1.0
    is_multistep = True
    general_rows = True  # rows of ROW16 floats for dm4d_cfg_multistep_step_*, planned from (step index, steps taken in this call)
    ROW = ROW16

    def _check_plain_sigmas(self, c, what):
        on = [k for k in _SIGMA_SWITCHES if getattr(c, k, False)]
        if getattr(c, "trained_betas", None) is not None:
            on.append("trained_betas")
        if on or c.prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError(f"{what}: plain sigmas and epsilon / v_prediction only (got prediction_type={c.prediction_type!r}, switched on: {on})")

    def _set(self, n, last_sigma):
        ts = _sigma_tables(self.config, n)
        ac = self.alphas_cumprod
        sig = np.interp(ts, np.arange(0, len(ac)), np.array(((1 - ac) / ac) ** 0.5))
        self.sigmas = np.concatenate([sig, [last_sigma]]).astype(np.float32)  # fp32 table, as the reference keeps it
        self.timesteps, self.num_inference_steps = ts, n
        return ts

    def _x0_coef(self, al_s, sg_s):
        """x0 = d x + e m at the current sigma."""
        if self.config.prediction_type == "epsilon":
            return 1.0 / al_s, -sg_s / al_s
        return al_s, -sg_s


class UniPCMultistepScheduler(_MultistepRows):
    """UniPC (data prediction, bh1 / bh2, orders 1 and 2) as coefficient rows; oracle/multistep.py is the stateful form the tests
    compare against (restated from the published algorithm and diffusers 0.33.1's step order: unpinned like every diffusers internal)."""
    state_slots = 3

    def __init__(self, config: UniPCConfig = UniPCConfig()):
        self.config = c = config
        self._check_plain_sigmas(c, "UniPCMultistepScheduler")
        if not c.predict_x0 or c.solver_type not in ("bh1", "bh2") or c.solver_order not in (1, 2) or c.solver_p is not None:
            raise NotImplementedError(f"UniPCMultistepScheduler: data-prediction bh1 / bh2 of order 1 or 2 is implemented (got predict_x0="
                                      f"{c.predict_x0}, solver_type={c.solver_type!r}, solver_order={c.solver_order}, solver_p={c.solver_p})")
        if c.final_sigmas_type == "zero" and not c.lower_order_final and c.solver_order > 1:
            raise NotImplementedError("UniPCMultistepScheduler: a zero final sigma needs lower_order_final (the second-order update divides by 0 there)")
        self.alphas_cumprod = _alphas_cumprod(c)
        self.num_inference_steps = self.timesteps = self.sigmas = None

    def set_timesteps(self, num_inference_steps: int) -> np.ndarray:
        ac0 = float(self.alphas_cumprod[0])
        if self.config.final_sigmas_type not in ("zero", "sigma_min"):
            raise NotImplementedError(f"final_sigmas_type {self.config.final_sigmas_type}")
        return self._set(num_inference_steps, 0.0 if self.config.final_sigmas_type == "zero" else ((1 - ac0) / ac0) ** 0.5)

    def _bh(self, hh, order, rk):
        """(h phi_1, B(h), rhos) of an update of `order` (1 or 2) with ratio rk (order 2), as diffusers' R / b system gives them."""
        h_phi_1 = np.expm1(hh)
        B_h = hh if self.config.solver_type == "bh1" else np.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1.0
        b1 = h_phi_k / B_h                   # i = 1: factorial 1
        h_phi_k2 = h_phi_k / hh - 0.5        # i = 2: factorial 2
        b2 = h_phi_k2 * 2.0 / B_h
        return h_phi_1, B_h, b1, b2

    def step_rows(self, step_index: np.ndarray, n_prev: np.ndarray) -> np.ndarray:
        """[..., 16] fp32 rows for latents at `step_index` that have taken `n_prev` steps in this call."""
        c, n = self.config, self.num_inference_steps
        idx = np.asarray(step_index, dtype=np.int64)
        cnt = np.minimum(np.asarray(n_prev, dtype=np.int64), c.solver_order)
        s = self.sigmas.astype(np.float64)
        rows = np.zeros(idx.shape + (self.ROW,), dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            lam = lambda i: (lambda a, g: np.log(a) - np.log(g))(*_alpha_sigma64(s[np.clip(i, 0, n)]))  # noqa: E731
            al_s0, sg_s0 = _alpha_sigma64(s[idx])
            d, e = self._x0_coef(al_s0, sg_s0)
            rows[..., 0], rows[..., 1] = d, e
            # ---- corrector: refine the sample of step index - 1 with this step's prediction (order = that step's predictor order)
            cap_prev = np.minimum(c.solver_order, n - (idx - 1)) if c.lower_order_final else np.full(idx.shape, c.solver_order)
            corr_order = np.minimum(cap_prev, cnt)
            use_corr = (cnt > 0) & (idx > 0) & ~np.isin(idx - 1, np.asarray(c.disable_corrector, dtype=np.int64))
            al_t, sg_t = al_s0, sg_s0                     # corrector's "t" is the current step, its "s0" the previous one
            al_p, sg_p = _alpha_sigma64(s[np.maximum(idx - 1, 0)])
            h = lam(idx) - lam(idx - 1)
            rk = (lam(idx - 2) - lam(idx - 1)) / h
            h_phi_1, B_h, b1, b2 = self._bh(-h, 2, rk)
            # order 1: rhos_c = [1/2];  order 2: solve [[1, 1], [rk, 1]] rho = [b1, b2]
            rho_last = np.where(corr_order >= 2, (b2 - rk * b1) / (1.0 - rk), 0.5)
            rho_0 = np.where(corr_order >= 2, (b1 - b2) / (1.0 - rk), 0.0)
            k3 = sg_t / sg_p                                                                  # last_sample
            k4 = -al_t * h_phi_1 - al_t * B_h * (np.where(corr_order >= 2, -rho_0 / rk, 0.0) - rho_last)   # m0 (= s1)
            k5 = np.where(corr_order >= 2, -al_t * B_h * rho_0 / rk, 0.0)                       # m1 (= s2)
            k6 = -al_t * B_h * rho_last                                                       # this step's prediction
            rows[..., 2] = np.where(use_corr, 0.0, 1.0)
            for j, k in ((3, k3), (4, k4), (5, k5), (6, k6)):
                rows[..., j] = np.where(use_corr, k, 0.0)
            # ---- predictor of order min(solver_order, steps left, steps taken + 1)
            cap = np.minimum(c.solver_order, n - idx) if c.lower_order_final else np.full(idx.shape, c.solver_order)
            order = np.minimum(cap, cnt + 1)
            al_n, sg_n = _alpha_sigma64(s[idx + 1])
            hp = lam(idx + 1) - lam(idx)             # +inf at a final sigma of zero
            rkp = (lam(idx - 1) - lam(idx)) / hp
            hh = -hp
            hphi1 = np.expm1(hh)
            Bp = hh if c.solver_type == "bh1" else np.expm1(hh)
            w = np.where(order >= 2, al_n * Bp * 0.5 / rkp, 0.0)   # rhos_p = [1/2]: pred_res = (s1 - conv) / (2 rk)
            rows[..., 7] = sg_n / sg_s0
            rows[..., 8] = -al_n * hphi1 + w
            rows[..., 9] = -w
        return np.nan_to_num(rows, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)


@dataclass
class DEISConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.0001
    beta_end: float = 0.02
    beta_schedule: str = "linear"
    solver_order: int = 2
    prediction_type: str = "epsilon"
    algorithm_type: str = "deis"
    solver_type: str = "logrho"
    lower_order_final: bool = True
    timestep_spacing: str = "linspace"
    steps_offset: int = 0
    thresholding: bool = False
    use_karras_sigmas: bool = False
    use_exponential_sigmas: bool = False
    use_beta_sigmas: bool = False
    use_flow_sigmas: bool = False
    rescale_betas_zero_snr: bool = False
    trained_betas: object = None

    @classmethod
    def from_dict(cls, d: Dict) -> "DEISConfig":
        names = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})


class DEISMultistepScheduler(_MultistepRows):
    """DEIS (tAB-DEIS in log-rho space, orders 1-3) as coefficient rows; the stored tensors are the last two noise-form model outputs."""
    state_slots = 2

    def __init__(self, config: DEISConfig = DEISConfig()):
        self.config = c = config
        self._check_plain_sigmas(c, "DEISMultistepScheduler")
        if c.algorithm_type != "deis" or c.solver_type != "logrho" or c.solver_order not in (1, 2, 3):
            raise NotImplementedError(f"DEISMultistepScheduler: algorithm_type 'deis', solver_type 'logrho', orders 1-3 (got {c.algorithm_type!r}, "
                                      f"{c.solver_type!r}, {c.solver_order})")
        self.alphas_cumprod = _alphas_cumprod(c)
        self.num_inference_steps = self.timesteps = self.sigmas = None

    def set_timesteps(self, num_inference_steps: int) -> np.ndarray:
        ac0 = float(self.alphas_cumprod[0])
        return self._set(num_inference_steps, ((1 - ac0) / ac0) ** 0.5)

    def step_rows(self, step_index: np.ndarray, n_prev: np.ndarray) -> np.ndarray:
        c, n = self.config, self.num_inference_steps
        idx = np.asarray(step_index, dtype=np.int64)
        cnt = np.minimum(np.asarray(n_prev, dtype=np.int64), c.solver_order)
        s = self.sigmas.astype(np.float64)
        small = c.lower_order_final and n < 15
        order = np.where((c.solver_order == 1) | (cnt < 1) | ((idx == n - 1) & small), 1,
                         np.where((c.solver_order == 2) | (cnt < 2) | ((idx == n - 2) & small), 2, 3))
        rows = np.zeros(idx.shape + (self.ROW,), dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            (al_t, sg_t), (al_0, sg_0) = _alpha_sigma64(s[idx + 1]), _alpha_sigma64(s[idx])
            al_1, sg_1 = _alpha_sigma64(s[np.maximum(idx - 1, 0)])
            al_2, sg_2 = _alpha_sigma64(s[np.maximum(idx - 2, 0)])
            d, e = self._x0_coef(al_0, sg_0)
            rows[..., 0] = (1.0 - al_0 * d) / sg_0      # conv = (x - alpha x0) / sigma: the noise form DEIS integrates
            rows[..., 1] = -al_0 * e / sg_0
            rows[..., 2] = 1.0                          # no corrector: xc = x
            rt, r0, r1, r2 = sg_t / al_t, sg_0 / al_0, sg_1 / al_1, sg_2 / al_2
            ln = np.log
            h = (ln(al_t) - ln(sg_t)) - (ln(al_0) - ln(sg_0))
            first = (-sg_t * (np.exp(h) - 1.0), 0.0, 0.0)

            def ind2(t, b, cc):
                return t * (-ln(cc) + ln(t) - 1.0) / (ln(b) - ln(cc))

            def ind3(t, b, cc, dd):
                num = t * (ln(cc) * (ln(dd) - ln(t) + 1.0) - ln(dd) * ln(t) + ln(dd) + ln(t) ** 2 - 2.0 * ln(t) + 2.0)
                return num / ((ln(b) - ln(cc)) * (ln(b) - ln(dd)))
            second = (al_t * (ind2(rt, r0, r1) - ind2(r0, r0, r1)), al_t * (ind2(rt, r1, r0) - ind2(r0, r1, r0)), 0.0)
            third = (al_t * (ind3(rt, r0, r1, r2) - ind3(r0, r0, r1, r2)), al_t * (ind3(rt, r1, r2, r0) - ind3(r0, r1, r2, r0)),
                     al_t * (ind3(rt, r2, r0, r1) - ind3(r0, r2, r0, r1)))
            rows[..., 7] = al_t / al_0
            for j in range(3):
                rows[..., 8 + j] = np.where(order == 1, first[j], np.where(order == 2, second[j], third[j]))
        return np.nan_to_num(rows, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)


@dataclass
class PNDMConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.0001
    beta_end: float = 0.02
    beta_schedule: str = "linear"
    trained_betas: object = None
    skip_prk_steps: bool = False
    set_alpha_to_one: bool = False
    prediction_type: str = "epsilon"
    timestep_spacing: str = "leading"
    steps_offset: int = 0

    @classmethod
    def from_dict(cls, d: Dict) -> "PNDMConfig":
        names = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in names})


class PNDMScheduler(_MultistepRows):
    """PNDM with ``skip_prk_steps`` (= PLMS, the stock scheduler of the Stable Diffusion family) as coefficient rows.

    The linear multistep part of Liu et al., "Pseudo Numerical Methods for Diffusion Models on Manifolds": the noise estimate that
    enters the transfer formula is an Adams-Bashforth combination of up to four model outputs,
        1: e0;   2: (3 e0 - e1) / 2;   3: (23 e0 - 16 e1 + 5 e2) / 12;   4+: (55 e0 - 59 e1 + 37 e2 - 9 e3) / 24,
    with one warm-up twist in diffusers 0.33.1's ``step_plms``: the scheduler object's SECOND call does not advance -- it re-does the
    first step from the stored sample with the average of the two outputs (improved Euler), and its output does not enter the history.
    ``set_timesteps`` repeats the second timestep for that purpose, so the table has n + 1 entries.  The reference makes fresh scheduler
    copies per ``sliding_iterative_denoise`` call and indexes the table with its own per-latent indices (pipeline_diffuman4d.py:265-271,
    413-422, 500-501): every latent therefore spends the second step of EVERY call on the repetition, wherever its index stands, and the
    object's counter -- not the timestep value -- decides.  Reproduced as planned rows of the general multistep kernel
    (dm4d_cfg_multistep_step_*: k11 = weight of the third stored output, k12 = keep / shift the history); stored tensors:
    s1, s2, s3 = the last three model outputs, s3 doubling as the stored sample between a call's first and second step.
    oracle/multistep.py::PNDMScheduler is the stateful form the tests compare against (unpinned like every diffusers internal)."""
    state_slots = 3

    def __init__(self, config: PNDMConfig = PNDMConfig()):
        self.config = c = config
        if not c.skip_prk_steps:
            raise NotImplementedError("PNDMScheduler: skip_prk_steps=false runs four Runge-Kutta model evaluations per step for the first "
                                      "three steps, which the reference's one-evaluation-per-step loop (pipeline_diffuman4d.py:398-422) cannot "
                                      "drive either; the Stable Diffusion family ships skip_prk_steps=true")
        if c.trained_betas is not None or c.prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError(f"PNDMScheduler: trained_betas / prediction_type {c.prediction_type!r} are not implemented")
        self.alphas_cumprod = _alphas_cumprod(c)
        self.final_alpha_cumprod = np.float32(1.0) if c.set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = self.timesteps = None

    def set_timesteps(self, num_inference_steps: int) -> np.ndarray:
        c, n = self.config, num_inference_steps
        last = c.num_train_timesteps
        if n > last:
            raise ValueError("num_inference_steps > num_train_timesteps")
        if c.timestep_spacing == "linspace":
            base = np.linspace(0, last - 1, n).round().astype(np.int64)
        elif c.timestep_spacing == "leading":
            base = (np.arange(0, n) * (last // n)).round().astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            base = np.round(np.arange(last, 0, -last / n))[::-1].astype(np.int64) - 1
        else:
            raise NotImplementedError(f"timestep_spacing {c.timestep_spacing}")
        # the second step of the schedule is repeated: n + 1 entries (the reference's indices stop at n - 1: its last latents end one
        # transfer short of the clean image, as they do under the reference)
        self.timesteps = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy().astype(np.int64)
        self.num_inference_steps = n
        return self.timesteps

    def step_rows(self, step_index: np.ndarray, n_prev: np.ndarray) -> np.ndarray:
        """[..., 16] fp32 rows for latents whose table index is `step_index` and that have taken `n_prev` steps in this call."""
        c, n = self.config, self.num_inference_steps
        idx = np.asarray(step_index, dtype=np.int64)
        cnt = np.asarray(n_prev, dtype=np.int64)
        ratio = c.num_train_timesteps // n
        t_given = self.timesteps[np.clip(idx, 0, len(self.timesteps) - 1)]
        redo = cnt == 1                                    # the object's second call: from t + ratio back to t, on the stored sample
        t = np.where(redo, t_given + ratio, t_given)
        prev = np.where(redo, t_given, t_given - ratio)
        ac = self.alphas_cumprod.astype(np.float64)
        a_t = ac[np.clip(t, 0, len(ac) - 1)]
        a_p = np.where(prev >= 0, ac[np.clip(prev, 0, len(ac) - 1)], np.float64(self.final_alpha_cumprod))
        b_t, b_p = 1.0 - a_t, 1.0 - a_p
        sample_coeff = np.sqrt(a_p / a_t)
        g = (a_p - a_t) / (a_t * np.sqrt(b_p) + np.sqrt(a_t * b_t * a_p))     # x' = sample_coeff xs - g eps
        if c.prediction_type == "v_prediction":                              # eps = sqrt(a_t) M + sqrt(b_t) xs
            k_x, k_m = sample_coeff - g * np.sqrt(b_t), -g * np.sqrt(a_t)
        else:
            k_x, k_m = sample_coeff, -g
        # Adams-Bashforth weights on (this output, s1, s2, s3) by the number of outputs in the history AFTER this step's append:
        # cnt 0 -> 1 (plain);  cnt 1 -> the repetition: (m + s1) / 2, nothing appended;  cnt 2 -> 2;  cnt 3 -> 3;  cnt >= 4 -> 4
        w = np.zeros(idx.shape + (4,), dtype=np.float64)
        w[cnt == 0] = (1.0, 0.0, 0.0, 0.0)
        w[cnt == 1] = (0.5, 0.5, 0.0, 0.0)
        w[cnt == 2] = (1.5, -0.5, 0.0, 0.0)
        w[cnt == 3] = (23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0, 0.0)
        w[cnt >= 4] = (55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0)
        rows = np.zeros(idx.shape + (self.ROW,), dtype=np.float64)
        rows[..., 1] = 1.0                                  # conv = the raw model output (what the history stores)
        rows[..., 2] = np.where(redo, 0.0, 1.0)             # xc = x ...
        rows[..., 3] = np.where(redo, 1.0, 0.0)             # ... or the stored sample (s3 after a call's first step)
        rows[..., 7] = k_x
        rows[..., 8] = k_m * w[..., 0]
        rows[..., 9] = k_m * w[..., 1]
        rows[..., 10] = k_m * w[..., 2]
        rows[..., 11] = k_m * w[..., 3]
        # stored tensors: first step (s1, s2, s3) <- (m, -, x);  repetition: kept;  afterwards a history shift.  After the repetition
        # s1 still holds the first output, so the second-order weights above read the right tensor
        rows[..., 12] = np.where(cnt == 0, 0.0, np.where(redo, 1.0, 2.0))
        return rows.astype(np.float32)


def load_scheduler(path):
    """`scheduler/scheduler_config.json` of a diffusers checkpoint -> the scheduler object of this package."""
    cfg = json.loads((Path(path) / "scheduler_config.json").read_text())
    name = cfg.get("_class_name", "DDIMScheduler")
    if name == "DDIMScheduler":
        return DDIMScheduler(DDIMConfig.from_dict(cfg))
    if name == "DPMSolverMultistepScheduler":
        return DPMSolverMultistepScheduler(DPMSolverConfig.from_dict(cfg))
    if name == "UniPCMultistepScheduler":
        return UniPCMultistepScheduler(UniPCConfig.from_dict(cfg))
    if name == "DEISMultistepScheduler":
        return DEISMultistepScheduler(DEISConfig.from_dict(cfg))
    if name == "PNDMScheduler":
        return PNDMScheduler(PNDMConfig.from_dict(cfg))
    raise NotImplementedError(
        f"scheduler {name}: DDIMScheduler, PNDMScheduler (skip_prk_steps), DPMSolverMultistepScheduler (dpmsolver++), UniPCMultistepScheduler "
        f"and DEISMultistepScheduler are implemented.  The reference's loop "
        f"passes a vector of per-frame timesteps to scale_model_input (pipeline_diffuman4d.py:376), so Euler / Heun / LMS cannot "
        f"be what a working checkpoint names; DDPM and the ancestral / SDE samplers draw noise from the reference's RNG stream")
