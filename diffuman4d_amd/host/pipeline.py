"""Diffuman4DPipeline on libdm4d.so -- drop-in for the reference's pipeline protocol.

Mirrors ``/root/reference/src/diffusers/pipelines/diffuman4d/pipeline_diffuman4d.py``:
``sliding_iterative_denoise`` (:439-559) keeps its keyword signature, its ``ValueError``s and its
return dict; the inner ``__call__`` (:289-437) becomes ``_denoise_window`` = 3 device launches
around the UNet (pack -> UNet -> CFG+DDIM) with no host synchronisation, because the timestep
bookkeeping is planned on the host up front (``schedule.plan_sweep``).

Differences that do not change results:
  * everything stays on the device in NHWC; NCHW only at the boundary;
  * the per-window ``x[window]`` gathers / ``latents[window] = ...`` scatter are index arrays
    consumed by the pack / step kernels;
  * random draws can be injected (``noise=``) because the reference never seeds.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Callable, Dict, List, NamedTuple, Optional

import numpy as np
import torch

from . import ops
from .schedule import SweepPlan, history_counts, history_flags, plan_sweep
from .scheduler import DDIMScheduler
from .unet import UNetMultiviewConditionModel

BF16 = torch.bfloat16
F32 = torch.float32


def _identity_tqdm(it, **kw):
    return it


class PoseFeatures(NamedTuple):
    """enable_pose_encoder checkpoints: PoseEncoder outputs for the task's skeleton images and for the CFG negative."""
    feat: torch.Tensor  # [N, h, w, C0]
    neg: torch.Tensor   # [1, h, w, C0]  (all -1 image, pipeline_diffuman4d.py:352-353)


class Diffuman4DPipeline:
    def __init__(self, vae, unet: UNetMultiviewConditionModel, scheduler: DDIMScheduler, device="cuda"):
        self.vae, self.unet, self.scheduler = vae, unet, scheduler
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self._device = dev
        # precision of the arithmetic (include/dm4d.h "Parity precision" / "fp16 precision"): "fast" = bf16 tensors and MFMA operands;
        # "parity" = fp32 tensors between kernels and two-term bf16 operands, "fp16" = fp32 tensors and single-term fp16 operands: both
        # within north_star's 1e-3 of the fp32 reference path on decoded RGB
        self.precision = getattr(unet, "precision", "fast")
        if vae is not None and getattr(vae, "precision", "fast") != self.precision:
            raise ValueError("the UNet and the VAE of a pipeline must be built with the same precision")
        self.parity, self.h16 = self.precision == "parity", self.precision == "fp16"
        self.wide = self.parity or self.h16  # task tensors and everything between kernels are fp32
        self.dtype = F32 if self.wide else BF16
        ucfg = getattr(unet, "config", None)
        if ucfg is not None and hasattr(ucfg, "in_channels"):
            want = 11 if getattr(ucfg, "enable_pose_encoder", False) else 15  # [latent 4 | Pluecker 6 | skeleton latent 4 | mask 1] (:389-395)
            if ucfg.in_channels != want:
                raise ValueError(f"unet/config.json: in_channels = {ucfg.in_channels}, but this pipeline assembles {want} channels per frame "
                                 f"(enable_pose_encoder = {bool(getattr(ucfg, 'enable_pose_encoder', False))}; pipeline_diffuman4d.py:389-395)")
        self.vae_scale_factor = vae.scale_factor if vae is not None else 8
        self._vae_cache: Dict[str, dict] = {"pixel": {}, "skeleton": {}}  # encoder moments by caller-supplied key
        # extension (off = compute what the reference computes): after the last 3-D attention layer run only the rows
        # whose noise prediction the DDIM step reads, i.e. drop the conditioning frames from the per-frame tail of the UNet
        self.prune_cond_rows = False

    def clear_vae_cache(self):
        self._vae_cache = {"pixel": {}, "skeleton": {}}

    @property
    def device(self) -> torch.device:
        return self._device

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, model_dir, torch_dtype=BF16, device="cuda", precision: str = "fast") -> "Diffuman4DPipeline":
        """diffusers checkpoint directory (sampling_utils.py:28-46): model_index.json, unet/, vae/, scheduler/.
        ``torch_dtype`` bf16 | fp16 selects the checkpoint FILES the way the reference does (``*model.safetensors`` vs
        ``*model.fp16.safetensors``, :28-33).  ``precision`` (extension; configs/model/diffuman4d_mi355x.yaml) selects the arithmetic:
        "fast" = bf16 MFMA operands and bf16 tensors whatever files were read (fp16 weights are converted once at load: their 10-bit
        mantissas are rounded to 7; no SD-class weight leaves bf16's range); "fp16" = fp16 MFMA operands over fp32 tensors, on the
        weights exactly as ``torch_dtype`` holds them -- the faithful form of the reference's fp16 pipelines (:27-29) and, on bf16
        weights, the fastest arithmetic that meets north_star's 1e-3 against the fp32 reference path; "parity" = two-term bf16
        operands over fp32 tensors (1e-5); "auto" = "fp16" for fp16 pipelines, "fast" for bf16 ones."""
        from .vae import AutoencoderKL
        if torch_dtype in (BF16, "bf16"):
            variant = None
        elif torch_dtype in (torch.float16, "fp16"):
            variant = "fp16"
        else:
            raise ValueError(f"Unsupported torch_dtype: {torch_dtype}. Supported types are 'bf16' and 'fp16'.")
        if precision == "auto":
            precision = "fp16" if variant == "fp16" else "fast"
        wdt = torch.float16 if variant == "fp16" else BF16
        model_dir = Path(model_dir)
        if (model_dir / "model_index.json").exists():
            json.loads((model_dir / "model_index.json").read_text())  # class names only; import paths are ignored
        unet = UNetMultiviewConditionModel.from_pretrained(model_dir / "unet", device, variant, precision, wdt)
        vae = AutoencoderKL.from_pretrained(model_dir / "vae", device, variant, precision, wdt)
        sched = DDIMScheduler.from_pretrained(model_dir / "scheduler")
        pipe = cls(vae, unet, sched, device)
        pipe.checkpoint_variant = variant
        pipe._source = (str(model_dir), torch_dtype, precision)
        return pipe

    def to(self, device):
        """Protocol compatibility (sampling_utils.py:46-47: ``from_pretrained(dir)`` then ``.to("cuda:i")``).  Weights are
        laid out for the kernels at load time, so moving = loading the checkpoint again on the target device."""
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if dev == self._device:
            return self
        src = getattr(self, "_source", None)
        if src is None:
            raise NotImplementedError("this pipeline was assembled from in-memory components; build it on the target device")
        moved = type(self).from_pretrained(src[0], torch_dtype=src[1], device=dev, precision=src[2] if len(src) > 2 else "fast")
        moved.prune_cond_rows = self.prune_cond_rows
        self.__dict__.update(moved.__dict__)
        return self

    def set_progress_bar_config(self, **kw):
        pass

    # ------------------------------------------------------------------------------------------
    def _to_dev_nhwc(self, x: torch.Tensor, cpad: Optional[int] = None) -> torch.Tensor:
        """CPU/GPU NCHW (any float dtype) -> device NHWC bf16 via the layout kernel (wide precisions: fp32, permuted where it lies)."""
        if self.wide:
            y = x.float().permute(0, 2, 3, 1)
            if cpad is not None and cpad > y.shape[-1]:
                y = torch.nn.functional.pad(y, (0, cpad - y.shape[-1]))
            return y.contiguous().to(self._device)
        x = x.to(device=self._device, dtype=BF16).contiguous()
        return ops.nchw_to_nhwc(x, cpad)

    def prepare_all_latents(self, pixel_values, plucker_embeds, skeletons, cond_masks, latents, noise: Optional[Dict],
                            cache_keys=None, cameras: Optional[Dict] = None):
        """pipeline_diffuman4d.py:193-263 (sliding entry).  Returns NHWC bf16 device tensors.
        cache_keys: one hashable key per frame (e.g. (camera, frame) labels) -> VAE encoder moments are reused
        across tasks and alternation rounds (see AutoencoderKL.encode_scaled).
        cameras (with plucker_embeds=None): {"Ks": [N,3,3], "poses": [N,4,4] camera-to-world, "image_size": (H, W)} -> the
        Pluecker maps are evaluated on the device at latent resolution (ops.plucker_latents) instead of being built at
        image resolution on the host, shipped and resized (spatem_dataset.py:169-176 + :90-100)."""
        noise = noise or {}
        n = pixel_values.shape[0]
        ck = dict(cache=self._vae_cache["pixel"], keys=cache_keys) if cache_keys is not None else {}
        pv_lat = self.vae.encode_scaled(pixel_values, noise.get("pixel"), **ck)  # [N,h,w,4], x scaling_factor
        h, w = pv_lat.shape[1:3]
        if plucker_embeds is not None:
            pl_lat = self.vae.resize_to_nhwc(plucker_embeds, (h, w), "bilinear")
        elif cameras is not None:
            pl_lat = ops.plucker_latents(cameras["Ks"], cameras["poses"], tuple(cameras["image_size"]), (h, w), self._device,
                                         out_f32=self.wide)
        else:
            raise ValueError("plucker_embeds is None and no cameras were given")
        if self.unet.config.enable_pose_encoder:
            # the reference hands the raw skeleton images to the UNet, which re-encodes them on every call
            # (:229-231; unet_multiview_condition.py:551-552); they do not change, so encode once per task
            sk = self._to_dev_nhwc(skeletons, 4)
            neg = torch.full_like(sk[:1], -1.0)
            neg[..., 3] = 0.0  # channel 3 is padding
            sk_lat = PoseFeatures(self.unet.pose_encoder(sk), self.unet.pose_encoder(neg))
        else:
            ck = dict(cache=self._vae_cache["skeleton"], keys=cache_keys) if cache_keys is not None else {}
            sk_lat = self.vae.encode_scaled(skeletons, noise.get("skeleton"), **ck) if skeletons is not None else None
        cm_lat = self.vae.resize_to_nhwc(cond_masks, (h, w), "nearest")
        if latents is None:
            if "latents" in noise:
                latents = noise["latents"]
            else:
                latents = torch.randn((n, 4, h, w), device=self._device, dtype=torch.float32)
        lat = self._to_dev_nhwc(latents)  # init_noise_sigma == 1 for DDIM
        return pv_lat, pl_lat, sk_lat, cm_lat, lat

    def seeded_noise(self, seed: int, n: int, h: int, w: int, need_latents: bool = True) -> Dict[str, torch.Tensor]:
        """The task's random draws (VAE posterior samples of the images and the skeleton maps, initial latents) from a generator of
        this device seeded with `seed`: the same numbers on every rank that passes the same seed (frame-shard groups)."""
        g = torch.Generator(device=self._device).manual_seed(seed)
        keys = ("pixel", "skeleton") + (("latents",) if need_latents else ())
        return {k: torch.randn((n, 4, h, w), generator=g, device=self._device, dtype=torch.float32) for k in keys}

    # ------------------------------------------------------------------------------------------
    def upload_plan(self, plan: SweepPlan, guidance_scale: float, shard=None, copies: int = 1, rows_per_task: int = 0):
        """Host plan -> device index / timestep / coefficient tables (one H2D each, no syncs later).
        With `shard` (parallel.FrameShard) the per-call rows are cut down to this rank's frames of every window;
        `win_full` keeps the whole window for the latent all-gather.
        copies > 1 (task batching): the tables of `copies` tasks with THIS plan whose tensors are stacked along the frame axis
        (task k occupies rows k * rows_per_task ...): every window call then carries copies x F frames, the 3-D attention still
        folds F frames per group (`frames_per_group`), and each task's arithmetic is what it is alone (per-row GEMMs, per-sample
        GroupNorm, per-(batch, head) attention; every tile choice is bit-identical)."""
        if copies > 1 and (shard is not None or rows_per_task <= 0):
            raise ValueError("task batching needs rows_per_task and is not combined with frame sharding")
        ts = self.scheduler.set_timesteps(plan.num_inference_steps)
        cfg = 2 if guidance_scale > 1 else 1
        win = np.stack(plan.windows).astype(np.int32)              # [calls, F]
        cond = np.stack(plan.is_cond)                              # [calls, F] bool
        t = ts[np.stack(plan.timestep_index)].astype(np.int64)     # [calls, F]
        t[cond] = 0                                                # get_timestep :277
        if getattr(self.scheduler, "is_multistep", False):
            # stateful scheduler (one deep copy per latent in the reference, made afresh for this call: :265-271, :500-501): which
            # update a latent gets depends on its step index and on whether it has been stepped before IN THIS CALL -- both known
            # from the plan -- so the objects collapse to one coefficient row per (call, frame)
            tix = np.stack(plan.timestep_index)
            if getattr(self.scheduler, "general_rows", False):  # UniPC / DEIS: rows depend on how many steps a latent has taken in this call
                coef = self.scheduler.step_rows(np.where(cond, 0, tix), np.stack(history_counts(plan.windows, plan.is_cond)))  # [calls, F, 16]
            else:
                has_prev = np.stack(history_flags(plan.windows, plan.is_cond))
                coef = self.scheduler.step_rows(np.where(cond, 0, tix), has_prev)   # [calls, F, 8]
        else:
            coef = self.scheduler.step_coefficients(t)             # [calls, F, 4]
        group = win.shape[1]  # frames folded into one 3-D attention sequence
        if copies > 1:
            win = np.concatenate([win + k * rows_per_task for k in range(copies)], axis=1).astype(np.int32)
            cond, t, coef = (np.concatenate([a] * copies, axis=1) for a in (cond, t, coef))
        win_full = win
        if shard is not None:
            sl = shard.local_frames(win.shape[1])
            win, cond, t, coef = win[:, sl], cond[:, sl], t[:, sl], coef[:, sl]
            group = win.shape[1]
        t_in = np.concatenate([t] * cfg, axis=1).astype(np.float32)
        dev = self._device

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

        # rows of the CFG batch whose noise prediction is consumed (non-conditioning frames, both halves), per call
        F = win.shape[1]
        keep = [up(np.concatenate([np.nonzero(~c)[0] + h * F for h in range(cfg)]).astype(np.int64)) for c in cond]
        return dict(win=up(win), cond=up(cond.astype(np.int32)), t=up(t_in), coef=up(coef), calls=win.shape[0], cfg=cfg,
                    win_full=up(win_full.astype(np.int64)), keep=keep, frames_per_group=group, copies=copies)

    def denoise_latents(self, pv_lat, pl_lat, sk_lat, cm_lat, lat, plan: SweepPlan, domain: str, guidance_scale: float,
                        tqdm: Callable = _identity_tqdm, tables=None, shard=None):
        """The window sweep (:521-543) on device-resident NHWC tensors; `lat` is updated in place.
        shard (parallel.FrameShard): every rank holds the whole task tensors but runs only its F/P frames of each
        window through the UNet (K/V all-gather inside the 3-D attention layers), then the updated latent rows are
        all-gathered so that all copies of `lat` stay identical."""
        tb = tables or self.upload_plan(plan, guidance_scale, shard)
        use_cfg = tb["cfg"] == 2
        N, h, w, _ = lat.shape
        HW = h * w
        lat3, pv3, pl3, cm3 = lat.view(N, HW, 4), pv_lat.view(N, HW, 4), pl_lat.view(N, HW, 6), cm_lat.view(N, HW, 1)
        if isinstance(sk_lat, PoseFeatures):
            sk3 = sk_lat
        else:
            sk3 = sk_lat.view(N, HW, 4) if sk_lat is not None else None
        vpred = self.scheduler.config.prediction_type == "v_prediction"
        F = tb["win"].shape[1]  # frames of a window handled by THIS rank
        domains = [domain] * tb["cfg"]
        # multistep schedulers: the latents' previous x0 predictions, the only state the reference's per-latent scheduler copies carry
        x0_prev = None
        if getattr(self.scheduler, "general_rows", False):  # UniPC / DEIS: up to three stored tensors per latent, zero at the start of a call
            x0_prev = [torch.zeros_like(lat3) for _ in range(self.scheduler.state_slots)]
        elif getattr(self.scheduler, "is_multistep", False):
            x0_prev = torch.zeros_like(lat3)
        for i in tqdm(range(tb["calls"]), total=tb["calls"]):
            self.window_call(lat3, pv3, pl3, sk3, cm3, tb, i, h, w, domains, guidance_scale, use_cfg, vpred, shard, x0_prev)
        return lat

    def window_call(self, lat3, pv3, pl3, sk3, cm3, tb, i, h, w, domains, guidance_scale, use_cfg, vpred, shard=None,
                    x0_prev=None):
        """One window: pack -> UNet -> CFG + scheduler step (pipeline_diffuman4d.py:369-423), all on device, no host sync.
        x0_prev: [N, HW, 4] state of a multistep scheduler (None for DDIM; a list of 1-3 such tensors for UniPC / DEIS)."""
        widx, cond = tb["win"][i], tb["cond"][i]
        F, HW = widx.shape[0], h * w
        pose = None
        if isinstance(sk3, PoseFeatures):  # negative half: features of the all -1 skeleton image (:352-353)
            pose = sk3.feat.index_select(0, widx.long())
            if use_cfg:
                pose = torch.cat([sk3.neg.expand(F, -1, -1, -1), pose])
            sk3 = None
        x = ops.pack_model_input(lat3, pv3, pl3, sk3, cm3, cond, self.unet.IN_PAD, use_cfg, frame_idx=widx, h16=self.h16)
        keep = tb["keep"][i] if self.prune_cond_rows else None
        fpg = tb.get("frames_per_group", F)  # < F when several tasks share the call (upload_plan copies)
        if len(domains) * fpg != tb["cfg"] * F:
            domains = list(domains[:1]) * (tb["cfg"] * F // fpg)
        eps = self.unet(x.view(tb["cfg"] * F, h, w, -1), tb["t"][i], domains=domains, num_frames=fpg, shard=shard,
                        pose_features=pose, keep_rows=keep)
        if keep is not None:  # back to one row per CFG-batch entry; the rows left at zero are never read by the step kernel
            full = torch.zeros((tb["cfg"] * F,) + tuple(eps.shape[1:]), dtype=eps.dtype, device=eps.device)
            eps = full.index_copy_(0, keep, eps)
        if isinstance(x0_prev, list):
            ops.cfg_multistep_step(lat3, x0_prev, eps.view(tb["cfg"] * F, HW, -1), tb["coef"][i], cond, use_cfg, float(guidance_scale),
                                   frame_idx=widx)
        elif x0_prev is not None:
            ops.cfg_linear_step(lat3, x0_prev, eps.view(tb["cfg"] * F, HW, -1), tb["coef"][i], cond, use_cfg,
                                float(guidance_scale), frame_idx=widx)
        else:
            ops.cfg_ddim_step(lat3, eps.view(tb["cfg"] * F, HW, -1), tb["coef"][i], cond, use_cfg, float(guidance_scale), vpred,
                              frame_idx=widx)
        if shard is not None:  # F/P updated rows per rank -> every rank's copy of the task latents (and of the scheduler state)
            rows = shard.gather_rows(lat3.index_select(0, widx.long()))
            lat3.index_copy_(0, tb["win_full"][i], rows)
            for st in (x0_prev if isinstance(x0_prev, list) else ([x0_prev] if x0_prev is not None else [])):
                rows = shard.gather_rows(st.index_select(0, widx.long()))
                st.index_copy_(0, tb["win_full"][i], rows)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sliding_iterative_denoise(self, pixel_values=None, plucker_embeds=None, skeletons=None, cond_masks=None,
                                  latents=None, domain: str = "spatial", timestep_indices=None, window_size: int = 12,
                                  sliding_stride: int = 1, sliding_shift: int = 0, bidirectional: bool = True,
                                  num_denoising_steps: int = 1, alternation_rounds: int = 3, guidance_scale: float = 2.0,
                                  tqdm: Callable = _identity_tqdm, noise: Optional[Dict] = None, cache_keys=None,
                                  decode: str = "all", cameras: Optional[Dict] = None, shard=None, noise_seed: Optional[int] = None):
        """Same contract as pipeline_diffuman4d.py:439-559 (inputs are not mutated).
        Extensions, off by default (= the reference's behaviour): `noise` injects the random draws; `cache_keys`
        (one hashable per frame) reuses VAE encoder moments across calls; `decode="denoised"` runs the VAE decoder
        only for fully denoised rows -- the only ones the sampler saves (sampling_utils.py:103-104) -- and returns
        zero images for the rest (`decode="none"`: for no row at all -- the non-leading ranks of a frame-shard group); `cameras` (with plucker_embeds=None) evaluates the Pluecker maps on the device at latent
        resolution (see prepare_all_latents); `shard` (parallel.FrameShard): the ranks of its group run THIS task together, every
        window call split over them by frames with K/V all-gathers in the 3-D attention layers -- every rank
        passes the same arguments and gets the same result; `noise_seed`: the random draws come from a device generator seeded
        with it instead of the global one, so that the ranks of a shard group draw the same numbers."""
        self._check_sweep_call(decode)
        plan, tensors = self._prepare_sweep(pixel_values, plucker_embeds, skeletons, cond_masks, latents, domain, timestep_indices,
                                            window_size, sliding_stride, sliding_shift, bidirectional, num_denoising_steps,
                                            alternation_rounds, noise, cache_keys, cameras, noise_seed)
        lat = tensors[4]
        self.denoise_latents(*tensors, plan, domain, guidance_scale, tqdm, shard=shard)
        return self._finish_sweep(lat, plan, decode)

    def _check_sweep_call(self, decode: str) -> None:
        if decode not in ("all", "denoised", "none"):
            raise ValueError("decode must be 'all', 'denoised' or 'none'")
        if self.vae is None:
            raise RuntimeError("this pipeline was built without a VAE; use denoise_latents()")
        torch.cuda.set_device(self._device)  # worker threads inherit device 0 (sampling_runner.py:36)

    def _prepare_sweep(self, pixel_values, plucker_embeds, skeletons, cond_masks, latents, domain, timestep_indices, window_size,
                       sliding_stride, sliding_shift, bidirectional, num_denoising_steps, alternation_rounds, noise, cache_keys,
                       cameras, noise_seed):
        """Window plan + the task's device tensors (pipeline_diffuman4d.py:466-519): (plan, (pv, pl, sk, cm, lat))."""
        cond_flags = (cond_masks[:, 0, 0, 0] == 0.0).cpu().numpy()
        plan = plan_sweep(cond_flags, torch.as_tensor(timestep_indices).cpu().numpy(), domain, window_size,
                          sliding_stride, sliding_shift, bidirectional, num_denoising_steps, alternation_rounds)
        if noise_seed is not None and noise is None:
            noise = self.seeded_noise(int(noise_seed), pixel_values.shape[0], pixel_values.shape[-2] // self.vae_scale_factor,
                                      pixel_values.shape[-1] // self.vae_scale_factor, need_latents=latents is None)
        return plan, self.prepare_all_latents(pixel_values, plucker_embeds, skeletons, cond_masks, latents, noise, cache_keys, cameras)

    def _finish_sweep(self, lat, plan: SweepPlan, decode: str) -> dict:
        if self.h16 and not bool(torch.isfinite(lat).all()):
            # fp16 operands stop at 65504: the norm / conversion kernels saturate, the GEMM and convolution epilogues do not, and an inf
            # operand turns the next product into NaN.  Seeded synthetic weights never get there; a checkpoint whose activations do
            # must be told so instead of being handed a NaN image (one 0.5 MB reduction per task, next to the decode)
            raise FloatingPointError("precision 'fp16': the denoised latents are not finite -- an activation left the fp16 range (|x| > 65504) "
                                     "on its way to a matrix product; run this checkpoint with precision 'fast' (bf16 range) or 'parity'")
        tidx = torch.from_numpy(plan.final_timestep_indices)
        rows = (tidx == plan.num_inference_steps) if decode == "denoised" else (torch.zeros_like(tidx, dtype=torch.bool) if decode == "none" else None)
        images = self.vae.decode_to_images(lat, rows=rows)  # [N,3,H,W] in [0,1]
        return {
            "images": images,
            "latents": ops.nhwc_to_nchw(lat),
            "timestep_indices": tidx,
            "fully_denoised": tidx == plan.num_inference_steps,
        }

    @staticmethod
    def same_plan(a: SweepPlan, b: SweepPlan) -> bool:
        """Two tasks can share their window calls exactly when their plans agree call by call."""
        def eq(x, y):
            return len(x) == len(y) and all(np.array_equal(u, v) for u, v in zip(x, y))
        return (a.num_inference_steps == b.num_inference_steps and eq(a.windows, b.windows) and eq(a.is_cond, b.is_cond)
                and eq(a.timestep_index, b.timestep_index) and np.array_equal(a.final_timestep_indices, b.final_timestep_indices))

    @torch.no_grad()
    def sliding_iterative_denoise_stack(self, tasks: List[Dict], domain: str = "spatial", window_size: int = 12, sliding_stride: int = 1,
                                        sliding_shift: int = 0, bidirectional: bool = True, num_denoising_steps: int = 1,
                                        alternation_rounds: int = 3, guidance_scale: float = 2.0, tqdm: Callable = _identity_tqdm,
                                        decode: str = "all") -> List[Dict]:
        """Extension (runner.task_batch): several tasks of ONE alternation round through SHARED window calls.  The reference runs one
        task per pipeline call (sliding_iterative_sampler.py:201-204); tasks of a round have the same geometry and the same timestep
        indices, hence the same window plan, so their tensors can be stacked along the frame axis: every call then carries
        len(tasks) x F frames, the 3-D attention still folds F frames per task, and each task's arithmetic is what it is alone
        (per-row GEMMs, per-sample GroupNorm, per-(task, head) attention) -- the results equal, bit for bit, those of
        `sliding_iterative_denoise` called once per task in list order (random draws included: the tasks are prepared in that
        order).  `tasks[k]`: the per-task arguments of `sliding_iterative_denoise` (pixel_values, plucker_embeds, skeletons, cond_masks,
        latents, timestep_indices and, optionally, noise, cache_keys, cameras, noise_seed).  Raises ValueError when the plans differ."""
        self._check_sweep_call(decode)
        if not tasks:
            return []
        per_task = ("pixel_values", "plucker_embeds", "skeletons", "cond_masks", "latents", "timestep_indices", "noise", "cache_keys",
                    "cameras", "noise_seed")
        preps = []
        for t in tasks:
            unknown = set(t) - set(per_task)
            if unknown:
                raise TypeError(f"unexpected per-task arguments: {sorted(unknown)}")
            a = {k: t.get(k) for k in per_task}
            preps.append(self._prepare_sweep(a["pixel_values"], a["plucker_embeds"], a["skeletons"], a["cond_masks"], a["latents"], domain,
                                             a["timestep_indices"], window_size, sliding_stride, sliding_shift, bidirectional,
                                             num_denoising_steps, alternation_rounds, a["noise"], a["cache_keys"], a["cameras"],
                                             a["noise_seed"]))
        plan = preps[0][0]
        n = preps[0][1][4].shape[0]
        for other, tens in preps[1:]:
            if tens[4].shape != preps[0][1][4].shape or not self.same_plan(plan, other):
                raise ValueError("tasks of a stack need identical window plans (same rows, conditioning flags and timestep indices)")
        if len(preps) == 1:
            stacked = preps[0][1]
        else:
            cols = list(zip(*[tens for _, tens in preps]))
            sk = cols[2]
            if isinstance(sk[0], PoseFeatures):
                sk = PoseFeatures(torch.cat([f.feat for f in sk]), sk[0].neg)
            elif sk[0] is not None:
                sk = torch.cat(sk)
            else:
                sk = None
            stacked = (torch.cat(cols[0]), torch.cat(cols[1]), sk, torch.cat(cols[3]), torch.cat(cols[4]))
        tables = self.upload_plan(plan, guidance_scale, copies=len(preps), rows_per_task=n) if len(preps) > 1 else None
        self.denoise_latents(*stacked, plan, domain, guidance_scale, tqdm, tables=tables)
        lat = stacked[4]
        return [self._finish_sweep(lat[k * n:(k + 1) * n], plan, decode) for k in range(len(preps))]
