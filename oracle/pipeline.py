"""Oracle pipeline: restatement of ``Diffuman4DPipeline`` (sliding path).  TEST INFRASTRUCTURE ONLY.

Follows ``/root/reference/src/diffusers/pipelines/diffuman4d/pipeline_diffuman4d.py``:
``encode_vae``/``decode_vae`` :47-72, ``encode_image_resizing`` :90-100, ``get_negative_latents``
:103-113, ``prepare_all_latents`` :193-263, ``get_timestep`` :273-278, ``__call__`` :289-437,
``sliding_iterative_denoise`` :439-559.  All random draws are injected.
"""
from __future__ import annotations

import copy

from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

from .ddim import DDIMScheduler
from .unet import UNetMultiviewConditionModel
from .vae import AutoencoderKL


def steps_per_alternation(window_size, sliding_stride, bidirectional, num_denoising_steps):
    """pipeline_diffuman4d.py:463-470."""
    if (window_size * num_denoising_steps) % sliding_stride != 0:
        raise ValueError(
            f"The window size ({window_size}) * num denoising steps ({num_denoising_steps}) "
            f"should be divisible by the sliding stride ({sliding_stride})"
        )
    n = window_size * num_denoising_steps // sliding_stride
    return n * 2 if bidirectional else n


def build_windows(target_indices: torch.Tensor, input_indices: torch.Tensor, domain: str, window_size: int,
                  sliding_stride: int, sliding_shift: int, bidirectional: bool):
    """pipeline_diffuman4d.py:503-518: list of (target_window, input_window) int64 tensors."""
    target_windows, input_windows = [], []
    directions = (-1, 1) if bidirectional else (-1,)
    for direction in directions:
        for shift in range(sliding_shift, sliding_shift + len(target_indices), sliding_stride):
            tw = target_indices.roll(shifts=shift * direction)[:window_size]
            target_windows.append(tw)
            if domain == "spatial":
                iw = input_indices
            elif domain == "temporal":
                iw = tw - len(input_indices)
            else:
                raise ValueError(domain)
            input_windows.append(iw)
    return target_windows, input_windows


class OraclePipeline:
    """vae / unet / scheduler triple with the reference's call semantics (CPU)."""

    def __init__(self, vae: Optional[AutoencoderKL], unet: UNetMultiviewConditionModel, scheduler: DDIMScheduler,
                 dtype=torch.float32):
        self.vae, self.unet, self.scheduler, self.dtype = vae, unet, scheduler, dtype
        self.device = torch.device("cpu")
        if vae is not None:
            self.vae.to(dtype)
        self.unet.to(dtype)

    # -- pipeline_diffuman4d.py:47-56 -------------------------------------------------
    def encode_vae(self, x, noise, batch_size=8):
        outs = []
        for xb, nb in zip(x.split(batch_size), noise.split(batch_size)):
            outs.append(self.vae.sample_posterior(self.vae.moments(xb), nb))
        return torch.cat(outs) * self.vae.cfg.scaling_factor

    # -- pipeline_diffuman4d.py:59-72, 280-285 ----------------------------------------
    def post_process(self, latents, batch_size=8):
        imgs = [self.vae.decode(lb / self.vae.cfg.scaling_factor) for lb in latents.split(batch_size)]
        return (torch.cat(imgs) / 2 + 0.5).clamp(0, 1)

    # -- pipeline_diffuman4d.py:193-263 (sliding entry: raw images in) -----------------
    def prepare_all_latents(self, pixel_values, plucker_embeds, skeletons, cond_masks, latents, noise: Dict):
        dt = self.dtype
        pv_lat = self.encode_vae(pixel_values.to(dt), noise["pixel"])
        n, _, h, w = pv_lat.shape
        pl_lat = F.interpolate(plucker_embeds, size=(h, w), mode="bilinear").to(dt)  # host fp32, then cast (:94-97)
        if self.unet.cfg.enable_pose_encoder:
            sk_lat = skeletons.to(dt)
        else:
            sk_lat = self.encode_vae(skeletons.to(dt), noise["skeleton"])
        cm_lat = F.interpolate(cond_masks, size=(h, w), mode="nearest").to(dt)
        if latents is None:
            latents = noise["latents"].to(dt)
        latents = latents * self.scheduler.init_noise_sigma
        return pv_lat, pl_lat, sk_lat, cm_lat, latents

    # -- pipeline_diffuman4d.py:289-437 ------------------------------------------------
    def denoise_window(self, pv_lat, pl_lat, sk_lat, cm_lat, latents, domains: List[str], num_inference_steps: int,
                       timesteps: torch.Tensor, timestep_indices: torch.Tensor, guidance_scale: float,
                       trace: Optional[list] = None, schedulers: Optional[list] = None):
        dt = self.dtype
        num_frames = pv_lat.shape[0]
        if schedulers is None:  # :361-363 -- one scheduler object per latent (stateful schedulers keep their history there)
            schedulers = [copy.deepcopy(self.scheduler) for _ in range(num_frames)]
        latents = latents * self.scheduler.init_noise_sigma  # :190 (second pass through prepare_latents)
        is_cond = cm_lat[:, 0, 0, 0] == 0  # :345
        do_cfg = guidance_scale > 1
        if do_cfg:  # :348-357
            neg_pv = torch.ones_like(pv_lat)
            pl_lat = torch.cat([torch.zeros_like(pl_lat), pl_lat])
            if sk_lat is not None:
                sk_lat = torch.cat([-torch.ones_like(sk_lat), sk_lat])
            cm_lat = torch.cat([cm_lat] * 2)
            domains = domains * 2
        timestep_indices = timestep_indices.clone()
        for _ in range(num_inference_steps):
            timestep_indices[is_cond] = 0  # :275
            timestep = timesteps[timestep_indices]
            timestep[is_cond] = 0  # :277
            x = self.scheduler.scale_model_input(latents)  # identity => alias
            x[is_cond] = pv_lat[is_cond]  # :379 (mutates `latents` too)
            t_in = timestep
            if do_cfg:
                t_in = torch.cat([timestep] * 2)
                neg = x.clone()
                neg[is_cond] = neg_pv[is_cond]
                x = torch.cat([neg, x])
            parts = [x, pl_lat]
            if sk_lat is not None and not self.unet.cfg.enable_pose_encoder:
                parts.append(sk_lat)
            parts.append(cm_lat)
            model_in = torch.cat(parts, dim=1)
            noise_pred = self.unet(model_in, t_in, skeletons=sk_lat, domains=domains, num_frames=num_frames)
            if trace is not None:
                trace.append({"model_in": model_in.clone(), "timestep": t_in.clone(), "noise_pred": noise_pred.clone()})
            if do_cfg:
                u, c = noise_pred.chunk(2)
                noise_pred = u + guidance_scale * (c - u)
            new = []
            for j in range(num_frames):  # :413-422
                lat = latents[j : j + 1]
                if not is_cond[j]:
                    lat = schedulers[j].step(noise_pred[j : j + 1], int(timestep[j]), lat)
                new.append(lat.to(dt))
            latents = torch.cat(new)
            timestep_indices[~is_cond] += 1
        return latents

    # -- pipeline_diffuman4d.py:439-559 ------------------------------------------------
    @torch.no_grad()
    def sliding_iterative_denoise(self, pixel_values, plucker_embeds, skeletons, cond_masks, latents, domain,
                                  timestep_indices, noise: Dict, window_size=12, sliding_stride=1, sliding_shift=0,
                                  bidirectional=True, num_denoising_steps=1, alternation_rounds=3, guidance_scale=2.0,
                                  decode=True, trace: Optional[list] = None):
        per_alt = steps_per_alternation(window_size, sliding_stride, bidirectional, num_denoising_steps)
        num_inference_steps = per_alt * alternation_rounds
        timestep_indices = timestep_indices.clone()
        target_indices = torch.where(cond_masks[:, 0, 0, 0] != 0.0)[0]
        input_indices = torch.where(cond_masks[:, 0, 0, 0] == 0.0)[0]
        tt = timestep_indices[target_indices]
        id_end = int(tt[0]) + per_alt
        if (tt != tt[0]).any():
            raise ValueError("The timestep indices should be the same for all target samples")
        if (timestep_indices[input_indices] != 0).any():
            raise ValueError("The timestep indices should be 0 for all input samples")
        pv_lat, pl_lat, sk_lat, cm_lat, latents = self.prepare_all_latents(
            pixel_values, plucker_embeds, skeletons, cond_masks, latents, noise)
        timesteps = self.scheduler.set_timesteps(num_inference_steps)
        schedulers = [copy.deepcopy(self.scheduler) for _ in range(len(latents))]  # :265-271, :500-501 ("a scheduler for each latent")
        tws, iws = build_windows(target_indices, input_indices, domain, window_size, sliding_stride, sliding_shift,
                                 bidirectional)
        for tw, iw in zip(tws, iws):
            window = torch.cat([iw, tw])
            out = self.denoise_window(pv_lat[window], pl_lat[window], sk_lat[window] if sk_lat is not None else None,
                                      cm_lat[window], latents[window], [domain], num_denoising_steps, timesteps,
                                      timestep_indices[window], guidance_scale, trace,
                                      schedulers=[schedulers[int(k)] for k in window])  # :535
            timestep_indices[tw] += num_denoising_steps  # :542
            latents[window] = out  # :543
        if (timestep_indices[target_indices] != id_end).any():
            raise ValueError("The denoised timesteps of target samples mismatch the config")
        if (timestep_indices[input_indices] != 0).any():
            raise ValueError("Timesteps of input samples have changed")
        images = self.post_process(latents) if decode else None
        return {"images": images, "latents": latents, "timestep_indices": timestep_indices,
                "fully_denoised": timestep_indices == num_inference_steps}
