"""Oracle UniPC and DEIS multistep schedulers (diffusers==0.33.1 ``UniPCMultistepScheduler`` / ``DEISMultistepScheduler``), stateful,
one object per latent as the reference keeps them.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

PARITY UNPINNED: diffusers is an un-vendored dependency of the reference (requirements.txt:5) and is not installed here; this file
restates the published algorithms -- Zhao et al., "UniPC" (B(h) = bh1 / bh2, data-prediction form, the order-1 / order-2 simplifications
rho = 1/2 of the 0.33.1 scheduler, corrector applied to the sample of the previous predictor step, order warm-up and lower order at the end)
and Zhang & Chen, "DEIS" (tAB-DEIS in log-rho space: the coefficients are integrals of the Lagrange basis through the last 1-3 points)
-- with the 0.33.1 schedulers' step order (convert_model_output, history shift, update, ``lower_order_nums``).  Why they matter here:
the reference takes whatever ``scheduler/scheduler_config.json`` names, deep-copies it per latent because these objects carry state
(pipeline_diffuman4d.py:265-271, 420, 500-501, 535) and hands ``scale_model_input`` a vector of timesteps (:376), which only
identity implementations survive -- DDIM, DPM-Solver, UniPC and DEIS among them.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch


def _betas(cfg):
    n = cfg.num_train_timesteps
    if cfg.beta_schedule == "scaled_linear":
        return torch.linspace(cfg.beta_start**0.5, cfg.beta_end**0.5, n, dtype=torch.float32) ** 2
    if cfg.beta_schedule == "linear":
        return torch.linspace(cfg.beta_start, cfg.beta_end, n, dtype=torch.float32)
    raise NotImplementedError(cfg.beta_schedule)


def _timesteps(cfg, n):
    last = cfg.num_train_timesteps
    if cfg.timestep_spacing == "linspace":
        return np.linspace(0, last - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
    if cfg.timestep_spacing == "leading":
        return (np.arange(0, n + 1) * (last // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + cfg.steps_offset
    if cfg.timestep_spacing == "trailing":
        return np.arange(last, 0, -last / n).round().copy().astype(np.int64) - 1
    raise NotImplementedError(cfg.timestep_spacing)


def _alpha_sigma(sigma):
    alpha_t = 1 / ((sigma**2 + 1) ** 0.5)
    return alpha_t, sigma * alpha_t


class _Base:
    init_noise_sigma = 1.0

    def _setup(self, cfg):
        self.cfg = cfg
        self.alphas_cumprod = torch.cumprod(1.0 - _betas(cfg), dim=0)
        self.timesteps = self.sigmas = None
        self.model_outputs = [None] * cfg.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    def _set(self, n, last_sigma):
        ts = _timesteps(self.cfg, n)
        ac = self.alphas_cumprod.numpy()
        sig = np.interp(ts, np.arange(0, len(ac)), np.array(((1 - ac) / ac) ** 0.5))
        self.sigmas = torch.from_numpy(np.concatenate([sig, [last_sigma]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * self.cfg.solver_order
        self.lower_order_nums = 0
        self._step_index = None
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _init_step_index(self, timestep):
        cand = (self.timesteps == int(timestep)).nonzero()
        self._step_index = len(self.timesteps) - 1 if len(cand) == 0 else int(cand[1] if len(cand) > 1 else cand[0])

    def _x0(self, model_output, sample):
        alpha_t, sigma_t = _alpha_sigma(self.sigmas[self._step_index])
        if self.cfg.prediction_type == "epsilon":
            return (sample - sigma_t * model_output) / alpha_t
        if self.cfg.prediction_type == "v_prediction":
            return alpha_t * sample - sigma_t * model_output
        raise NotImplementedError(self.cfg.prediction_type)


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
    disable_corrector: List[int] = field(default_factory=list)
    timestep_spacing: str = "linspace"
    steps_offset: int = 0
    final_sigmas_type: str = "zero"


class UniPCMultistepScheduler(_Base):
    def __init__(self, cfg: UniPCConfig = UniPCConfig()):
        if not cfg.predict_x0 or cfg.solver_type not in ("bh1", "bh2") or cfg.solver_order not in (1, 2, 3):
            raise NotImplementedError((cfg.predict_x0, cfg.solver_type, cfg.solver_order))
        self._setup(cfg)
        self.last_sample, self.this_order = None, None

    def set_timesteps(self, n):
        ac0 = float(self.alphas_cumprod[0])
        last = {"zero": 0.0, "sigma_min": ((1 - ac0) / ac0) ** 0.5}[self.cfg.final_sigmas_type]
        self.last_sample, self.this_order = None, None
        return self._set(n, last)

    def _rb(self, rks, hh, order):
        """R (Vandermonde rows of rks), b (the h phi_k factorial / B(h) column), h phi_1 and B(h) of an update of `order`."""
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.cfg.solver_type == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return torch.stack(R), torch.stack(b), h_phi_1, B_h

    def _lam(self, i):
        alpha, sigma = _alpha_sigma(self.sigmas[i])
        return torch.log(alpha) - torch.log(sigma), alpha, sigma

    def _predict(self, sample, order):
        i = self._step_index
        m0 = self.model_outputs[-1]
        lam_t, alpha_t, sigma_t = self._lam(i + 1)
        lam_s0, _, sigma_s0 = self._lam(i)
        h = lam_t - lam_s0
        rks, D1s = [], []
        for k in range(1, order):
            rk = (self._lam(i - k)[0] - lam_s0) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(k + 1)] - m0) / rk)
        rks = torch.stack(rks + [torch.tensor(1.0)])
        R, b, h_phi_1, B_h = self._rb(rks, -h, order)
        x = sigma_t / sigma_s0 * sample - alpha_t * h_phi_1 * m0
        if D1s:
            rhos = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
            x = x - alpha_t * B_h * sum(r_ * d for r_, d in zip(rhos, D1s))
        return x

    def _correct(self, model_t, last_sample, order):
        i = self._step_index
        m0 = self.model_outputs[-1]
        lam_t, alpha_t, sigma_t = self._lam(i)
        lam_s0, _, sigma_s0 = self._lam(i - 1)
        h = lam_t - lam_s0
        rks, D1s = [], []
        for k in range(1, order):
            rk = (self._lam(i - (k + 1))[0] - lam_s0) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(k + 1)] - m0) / rk)
        rks = torch.stack(rks + [torch.tensor(1.0)])
        R, b, h_phi_1, B_h = self._rb(rks, -h, order)
        rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
        corr = sum(r_ * d for r_, d in zip(rhos[:-1], D1s)) if D1s else 0.0
        return sigma_t / sigma_s0 * last_sample - alpha_t * h_phi_1 * m0 - alpha_t * B_h * (corr + rhos[-1] * (model_t - m0))

    def step(self, model_output: torch.Tensor, t: int, sample: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        if self._step_index is None:
            self._init_step_index(t)
        i = self._step_index
        use_corrector = i > 0 and (i - 1) not in cfg.disable_corrector and self.last_sample is not None
        x0 = self._x0(model_output, sample)
        if use_corrector:
            sample = self._correct(x0, self.last_sample, self.this_order)
        for k in range(cfg.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0
        this_order = min(cfg.solver_order, len(self.timesteps) - i) if cfg.lower_order_final else cfg.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < cfg.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return prev.to(model_output.dtype)


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


class DEISMultistepScheduler(_Base):
    def __init__(self, cfg: DEISConfig = DEISConfig()):
        if cfg.algorithm_type != "deis" or cfg.solver_type != "logrho" or cfg.solver_order not in (1, 2, 3):
            raise NotImplementedError((cfg.algorithm_type, cfg.solver_type, cfg.solver_order))
        self._setup(cfg)

    def set_timesteps(self, n):
        ac0 = float(self.alphas_cumprod[0])
        return self._set(n, ((1 - ac0) / ac0) ** 0.5)  # the sigma of training step 0: DEIS has no zero final sigma

    def _convert(self, model_output, sample):
        alpha_t, sigma_t = _alpha_sigma(self.sigmas[self._step_index])
        return (sample - alpha_t * self._x0(model_output, sample)) / sigma_t  # the noise-prediction form DEIS integrates

    def _update(self, sample, order):
        i = self._step_index
        a, s = zip(*[_alpha_sigma(self.sigmas[i + 1 - k]) for k in range(order + 1)])  # t, s0, s1, s2
        m = [self.model_outputs[-(k + 1)] for k in range(order)]
        if order == 1:
            h = (torch.log(a[0]) - torch.log(s[0])) - (torch.log(a[1]) - torch.log(s[1]))
            return (a[0] / a[1]) * sample - (s[0] * (torch.exp(h) - 1.0)) * m[0]
        rho = [float(s_ / a_) for a_, s_ in zip(a, s)]
        ln = np.log
        if order == 2:
            def ind(t, b, c):  # Integrate[(log t - log c) / (log b - log c), t]
                return t * (-ln(c) + ln(t) - 1) / (ln(b) - ln(c))
            c1 = ind(rho[0], rho[1], rho[2]) - ind(rho[1], rho[1], rho[2])
            c2 = ind(rho[0], rho[2], rho[1]) - ind(rho[1], rho[2], rho[1])
            return a[0] * (sample / a[1] + c1 * m[0] + c2 * m[1])

        def ind(t, b, c, d):  # Integrate[(log t - log c)(log t - log d) / ((log b - log c)(log b - log d)), t]
            num = t * (ln(c) * (ln(d) - ln(t) + 1) - ln(d) * ln(t) + ln(d) + ln(t) ** 2 - 2 * ln(t) + 2)
            return num / ((ln(b) - ln(c)) * (ln(b) - ln(d)))
        c1 = ind(rho[0], rho[1], rho[2], rho[3]) - ind(rho[1], rho[1], rho[2], rho[3])
        c2 = ind(rho[0], rho[2], rho[3], rho[1]) - ind(rho[1], rho[2], rho[3], rho[1])
        c3 = ind(rho[0], rho[3], rho[1], rho[2]) - ind(rho[1], rho[3], rho[1], rho[2])
        return a[0] * (sample / a[1] + c1 * m[0] + c2 * m[1] + c3 * m[2])

    def step(self, model_output: torch.Tensor, t: int, sample: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        if self._step_index is None:
            self._init_step_index(t)
        i, n = self._step_index, len(self.timesteps)
        lower_final = i == n - 1 and cfg.lower_order_final and n < 15
        lower_second = i == n - 2 and cfg.lower_order_final and n < 15
        conv = self._convert(model_output, sample)
        for k in range(cfg.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = conv
        if cfg.solver_order == 1 or self.lower_order_nums < 1 or lower_final:
            prev = self._update(sample, 1)
        elif cfg.solver_order == 2 or self.lower_order_nums < 2 or lower_second:
            prev = self._update(sample, 2)
        else:
            prev = self._update(sample, 3)
        if self.lower_order_nums < cfg.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return prev.to(model_output.dtype)


# ---- PNDM with skip_prk_steps (PLMS): diffusers 0.33.1 ``PNDMScheduler.step_plms`` / ``_get_prev_sample`` / ``set_timesteps`` -------------
@dataclass
class PNDMConfig:
    num_train_timesteps: int = 1000
    beta_start: float = 0.0001
    beta_end: float = 0.02
    beta_schedule: str = "linear"
    skip_prk_steps: bool = True
    set_alpha_to_one: bool = False
    prediction_type: str = "epsilon"
    timestep_spacing: str = "leading"
    steps_offset: int = 0


class PNDMScheduler:
    """Stateful PLMS as the reference's per-latent copies run it: a counter, the last four model outputs, the stored first sample.
    The object's second ``step`` re-does the first transfer with the averaged output (``counter == 1``); which timestep VALUE it is
    handed does not matter to that decision (the reference hands it whatever its own index table says)."""
    init_noise_sigma = 1.0

    def __init__(self, cfg: PNDMConfig = PNDMConfig()):
        assert cfg.skip_prk_steps, "the Runge-Kutta warm-up needs four model evaluations per step"
        self.cfg = cfg
        self.alphas_cumprod = torch.cumprod(1.0 - _betas(cfg), dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg.set_alpha_to_one else self.alphas_cumprod[0]
        self.timesteps = None
        self.ets: List[torch.Tensor] = []
        self.counter = 0
        self.cur_sample = None

    def set_timesteps(self, n):
        c, last = self.cfg, self.cfg.num_train_timesteps
        self.num_inference_steps = n
        if c.timestep_spacing == "linspace":
            base = np.linspace(0, last - 1, n).round().astype(np.int64)
        elif c.timestep_spacing == "leading":
            base = (np.arange(0, n) * (last // n)).round().astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            base = np.round(np.arange(last, 0, -last / n))[::-1].astype(np.int64) - 1
        else:
            raise NotImplementedError(c.timestep_spacing)
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        self.ets, self.counter, self.cur_sample = [], 0, None
        return self.timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _prev_sample(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        if self.cfg.prediction_type == "v_prediction":
            model_output = (a_t**0.5) * model_output + (b_t**0.5) * sample
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return sample_coeff * sample - (a_p - a_t) * model_output / denom

    def step(self, model_output: torch.Tensor, t: int, sample: torch.Tensor) -> torch.Tensor:
        ratio = self.cfg.num_train_timesteps // self.num_inference_steps
        timestep, prev_timestep = int(t), int(t) - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = timestep
            timestep = timestep + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        out = self._prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return out
