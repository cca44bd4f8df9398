"""Integer bookkeeping of the sliding iterative denoiser (bit-exact with the reference).

Everything here is host-side index arithmetic restating
``pipeline_diffuman4d.py:463-472`` (step budget), ``:474-487`` (entry checks), ``:503-518`` (window
list), ``:273-278,423,542`` (timestep-index evolution) and ``:546-551`` (exit checks).  Because the
evolution of every latent's timestep index is data independent, the whole sweep is planned up front
-- the device loop then runs without a single host synchronisation (the reference syncs once per
latent per step through ``.item()``, :415).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence

import numpy as np


def steps_per_alternation(window_size: int, sliding_stride: int, bidirectional: bool, num_denoising_steps: int) -> int:
    if (window_size * num_denoising_steps) % sliding_stride != 0:
        raise ValueError(
            f"The window size ({window_size}) * num denoising steps ({num_denoising_steps}) "
            f"should be divisible by the sliding stride ({sliding_stride})"
        )
    n = window_size * num_denoising_steps // sliding_stride
    return n * 2 if bidirectional else n


def build_windows(target_indices: Sequence[int], input_indices: Sequence[int], domain: str, window_size: int,
                  sliding_stride: int, sliding_shift: int, bidirectional: bool):
    """-> (target_windows, input_windows): lists of int64 arrays (torch.roll semantics, :508)."""
    tgt = np.asarray(target_indices, dtype=np.int64)
    inp = np.asarray(input_indices, dtype=np.int64)
    tws, iws = [], []
    for direction in ((-1, 1) if bidirectional else (-1,)):
        for shift in range(sliding_shift, sliding_shift + len(tgt), sliding_stride):
            tw = np.roll(tgt, shift * direction)[:window_size]
            tws.append(tw)
            if domain == "spatial":
                iws.append(inp)
            elif domain == "temporal":
                iws.append(tw - len(inp))
            else:
                raise ValueError(f"unknown domain {domain!r}")
    return tws, iws


@dataclass
class SweepPlan:
    """One task's full window sweep, planned on the host."""

    num_inference_steps: int
    windows: List[np.ndarray]          # per UNet call: frame indices fed to the UNet (inputs first, then targets)
    is_cond: List[np.ndarray]          # per call: bool [F]
    timestep_index: List[np.ndarray]   # per call: int64 [F] index into scheduler.timesteps (cond rows 0)
    final_timestep_indices: np.ndarray  # [N] after the sweep
    target_indices: np.ndarray
    input_indices: np.ndarray


def plan_sweep(cond_flags: Sequence[bool], timestep_indices: Sequence[int], domain: str, window_size: int,
               sliding_stride: int, sliding_shift: int, bidirectional: bool, num_denoising_steps: int,
               alternation_rounds: int) -> SweepPlan:
    """cond_flags[i] is True for input (conditioning) frames, i.e. cond_masks[i,0,0,0] == 0."""
    per_alt = steps_per_alternation(window_size, sliding_stride, bidirectional, num_denoising_steps)
    num_inference_steps = per_alt * alternation_rounds
    cond = np.asarray(cond_flags, dtype=bool)
    idx = np.asarray(timestep_indices, dtype=np.int64).copy()
    target_indices = np.nonzero(~cond)[0]
    input_indices = np.nonzero(cond)[0]
    tt = idx[target_indices]
    if (tt != tt[0]).any():
        raise ValueError(f"The timestep indices should be the same for all target samples, timestep_indices = {idx}")
    if (idx[input_indices] != 0).any():
        raise ValueError(f"The timestep indices should be 0 for all input samples, timestep_indices = {idx}")
    id_end = int(tt[0]) + per_alt
    tws, iws = build_windows(target_indices, input_indices, domain, window_size, sliding_stride, sliding_shift,
                             bidirectional)
    windows, conds, tidx = [], [], []
    for tw, iw in zip(tws, iws):
        window = np.concatenate([iw, tw])
        wc = cond[window]
        local = idx[window].copy()
        for _ in range(num_denoising_steps):
            local[wc] = 0  # get_timestep, :275
            windows.append(window)
            conds.append(wc.copy())
            tidx.append(local.copy())
            local[~wc] += 1  # :423
        idx[tw] += num_denoising_steps  # :542
    if (idx[target_indices] != id_end).any():
        raise ValueError(f"The denoised timesteps of target samples mismatch the config, timestep_indices = {idx}")
    if (idx[input_indices] != 0).any():
        raise ValueError(f"Timesteps of input samples have changed, timestep_indices = {idx}")
    return SweepPlan(num_inference_steps, windows, conds, tidx, idx, target_indices, input_indices)


def history_counts(windows: Sequence[np.ndarray], is_cond: Sequence[np.ndarray]) -> List[np.ndarray]:
    """Per call: HOW MANY steps each frame of the window has taken earlier in this plan (0 for conditioning frames) -- what a
    multistep scheduler of order > 2, or one with a corrector, needs to know about its per-latent object's history
    (``lower_order_nums`` / ``last_sample`` of a scheduler copy made afresh for the call, pipeline_diffuman4d.py:500-501)."""
    taken: dict = {}
    out = []
    for w, c in zip(windows, is_cond):
        out.append(np.array([0 if ic else taken.get(int(i), 0) for i, ic in zip(w, c)], dtype=np.int64))
        for i, ic in zip(w, c):
            if not ic:
                taken[int(i)] = taken.get(int(i), 0) + 1
    return out


def history_flags(windows: Sequence[np.ndarray], is_cond: Sequence[np.ndarray]) -> List[np.ndarray]:
    """Per call: which frames of the window have been stepped EARLIER IN THIS PLAN.  The reference makes one fresh scheduler
    object per latent for every sliding_iterative_denoise call (pipeline_diffuman4d.py:500-501), so a multistep scheduler has
    a previous prediction for a latent exactly from that latent's second step of the call on."""
    stepped = set()
    out = []
    for w, c in zip(windows, is_cond):
        out.append(np.array([(not ic) and int(i) in stepped for i, ic in zip(w, c)], dtype=bool))
        stepped.update(int(i) for i, ic in zip(w, c) if not ic)
    return out
