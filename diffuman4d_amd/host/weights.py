"""Checkpoint key/shape tables (diffusers naming) and synthetic weights.

There are no public weights in the build environment, so benchmarks and smoke
tests instantiate the *architecture* with seeded random weights.  The tables double as the strict
key contract a real ``unet/diffusion_pytorch_model.safetensors`` / ``vae/...`` must satisfy
(state_dict key skeleton).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

from .unet import UNetConfig
from .vae import VAEConfig

Shapes = Dict[str, Tuple[int, ...]]


def _resnet(s: Shapes, p: str, cin: int, cout: int, temb):
    s[p + "norm1.weight"] = (cin,)
    s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3)
    s[p + "conv1.bias"] = (cout,)
    if temb is not None:
        s[p + "time_emb_proj.weight"] = (cout, temb)
        s[p + "time_emb_proj.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,)
    s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3)
    s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + "conv_shortcut.bias"] = (cout,)


def _transformer(s: Shapes, p: str, c: int, linear: bool):
    s[p + "norm.weight"] = (c,)
    s[p + "norm.bias"] = (c,)
    for n in ("proj_in", "proj_out"):
        s[p + n + ".weight"] = (c, c) if linear else (c, c, 1, 1)
        s[p + n + ".bias"] = (c,)
    b = p + "transformer_blocks.0."
    for n in ("norm1", "norm3"):
        s[b + n + ".weight"] = (c,)
        s[b + n + ".bias"] = (c,)
    for n in ("to_q", "to_k", "to_v"):
        s[b + f"attn1.{n}.weight"] = (c, c)
    s[b + "attn1.to_out.0.weight"] = (c, c)
    s[b + "attn1.to_out.0.bias"] = (c,)
    s[b + "ff.net.0.proj.weight"] = (8 * c, c)
    s[b + "ff.net.0.proj.bias"] = (8 * c,)
    s[b + "ff.net.2.weight"] = (c, 4 * c)
    s[b + "ff.net.2.bias"] = (c,)


def unet_param_shapes(cfg: UNetConfig) -> Shapes:
    s: Shapes = {}
    boc = cfg.block_out_channels
    temb = boc[0] * 4
    s["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (boc[0],)
    for n in ["time_embedding"] + (["temporal_pos_embed"] if cfg.enable_tem_embeds else []):
        s[n + ".linear_1.weight"] = (temb, boc[0])
        s[n + ".linear_1.bias"] = (temb,)
        s[n + ".linear_2.weight"] = (temb, temb)
        s[n + ".linear_2.bias"] = (temb,)
    if cfg.enable_pose_encoder:  # pose_encoder.py:11-36
        from .unet import _PoseEncoder
        for n, (ci, co, k, _) in enumerate(_PoseEncoder.LAYERS):
            s[f"pose_encoder.conv_layers.{2 * n}.weight"] = (co, ci, k, k)
            s[f"pose_encoder.conv_layers.{2 * n}.bias"] = (co,)
        s["pose_encoder.final_proj.weight"] = (boc[0], 128, 1, 1)
        s["pose_encoder.final_proj.bias"] = (boc[0],)
        s["pose_encoder.scale"] = (1,)
    out_c = boc[0]
    for i, t in enumerate(cfg.down_block_types):
        in_c, out_c = out_c, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet(s, f"down_blocks.{i}.resnets.{j}.", in_c if j == 0 else out_c, out_c, temb)
            if t != "DownBlock2D":
                _transformer(s, f"down_blocks.{i}.attentions.{j}.", out_c, cfg.use_linear_projection)
        if i != len(boc) - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (out_c,)
    c = boc[-1]
    _resnet(s, "mid_block.resnets.0.", c, c, temb)
    _transformer(s, "mid_block.attentions.0.", c, cfg.use_linear_projection)
    _resnet(s, "mid_block.resnets.1.", c, c, temb)
    rboc = list(reversed(boc))
    out_c = rboc[0]
    n = cfg.layers_per_block + 1
    for i, t in enumerate(cfg.up_block_types):
        prev_c, out_c = out_c, rboc[i]
        in_c = rboc[min(i + 1, len(boc) - 1)]
        for j in range(n):
            skip = in_c if j == n - 1 else out_c
            rin = prev_c if j == 0 else out_c
            _resnet(s, f"up_blocks.{i}.resnets.{j}.", rin + skip, out_c, temb)
            if t != "UpBlock2D":
                _transformer(s, f"up_blocks.{i}.attentions.{j}.", out_c, cfg.use_linear_projection)
        if i != len(boc) - 1:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out_c,)
    s["conv_norm_out.weight"] = (boc[0],)
    s["conv_norm_out.bias"] = (boc[0],)
    s["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3)
    s["conv_out.bias"] = (cfg.out_channels,)
    return s


def vae_param_shapes(cfg: VAEConfig) -> Shapes:
    s: Shapes = {}
    boc, lc = cfg.block_out_channels, cfg.latent_channels

    def mid(p, c):
        _resnet(s, p + "resnets.0.", c, c, None)
        a = p + "attentions.0."
        s[a + "group_norm.weight"] = (c,)
        s[a + "group_norm.bias"] = (c,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            s[a + n + ".weight"] = (c, c)
            s[a + n + ".bias"] = (c,)
        _resnet(s, p + "resnets.1.", c, c, None)

    s["encoder.conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    s["encoder.conv_in.bias"] = (boc[0],)
    c = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet(s, f"encoder.down_blocks.{i}.resnets.{j}.", c if j == 0 else co, co, None)
        if i != len(boc) - 1:
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (co, co, 3, 3)
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (co,)
        c = co
    mid("encoder.mid_block.", c)
    s["encoder.conv_norm_out.weight"] = (c,)
    s["encoder.conv_norm_out.bias"] = (c,)
    s["encoder.conv_out.weight"] = (2 * lc, c, 3, 3)
    s["encoder.conv_out.bias"] = (2 * lc,)
    s["quant_conv.weight"] = (2 * lc, 2 * lc, 1, 1)
    s["quant_conv.bias"] = (2 * lc,)
    s["post_quant_conv.weight"] = (lc, lc, 1, 1)
    s["post_quant_conv.bias"] = (lc,)
    rboc = list(reversed(boc))
    s["decoder.conv_in.weight"] = (rboc[0], lc, 3, 3)
    s["decoder.conv_in.bias"] = (rboc[0],)
    mid("decoder.mid_block.", rboc[0])
    c = rboc[0]
    for i, co in enumerate(rboc):
        for j in range(cfg.layers_per_block + 1):
            _resnet(s, f"decoder.up_blocks.{i}.resnets.{j}.", c if j == 0 else co, co, None)
        if i != len(boc) - 1:
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (co, co, 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (co,)
        c = co
    s["decoder.conv_norm_out.weight"] = (c,)
    s["decoder.conv_norm_out.bias"] = (c,)
    s["decoder.conv_out.weight"] = (cfg.out_channels, c, 3, 3)
    s["decoder.conv_out.bias"] = (cfg.out_channels,)
    return s


def random_state_dict(shapes: Shapes, seed: int = 0, device="cpu", dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights that keep activations O(1): N(0, 1/fan_in) matrices, norm gamma ~ 1,
    small biases.  Generated on `device` (torch RNG is plumbing here, not compute on the hot path)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for k, shp in shapes.items():
        if len(shp) >= 2:
            fan_in = math.prod(shp[1:])
            t = torch.randn(shp, generator=g, device=device, dtype=torch.float32) * (1.0 / math.sqrt(fan_in))
        elif "norm" in k and k.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        else:
            t = 0.05 * torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        out[k] = t.to(dtype)
    return out


def load_component_state_dict(path, variant: str = None) -> Dict[str, torch.Tensor]:
    """state_dict of one diffusers component folder (``unet/`` or ``vae/``), by the file names diffusers itself uses.

    ``variant=None`` (bf16 runs): ``diffusion_pytorch_model.safetensors`` -- never a ``*.fp16.*`` file that an fp16 run
    of the reference downloaded into the same directory (sampling_utils.py:28-33 keeps both patterns in one folder).
    ``variant="fp16"``: ``diffusion_pytorch_model.fp16.safetensors``, falling back to the un-suffixed file (what
    ``from_pretrained(torch_dtype=float16)`` loads when no variant file exists).  A sharded checkpoint
    (``<name>.safetensors.index.json`` with a ``weight_map``) is merged from all of its shards; a sharded VARIANT follows
    diffusers' naming: index ``diffusion_pytorch_model.safetensors.index.fp16.json``, shards
    ``diffusion_pytorch_model.fp16-0000x-of-0000y.safetensors`` (the ``<stem>.fp16.safetensors.index.json`` spelling is
    accepted too)."""
    import json
    from pathlib import Path

    from safetensors.torch import load_file
    path = Path(path)
    stems = ["diffusion_pytorch_model"]
    if variant:
        stems.insert(0, f"diffusion_pytorch_model.{variant}")
    for stem in stems:
        single, index = path / f"{stem}.safetensors", path / f"{stem}.safetensors.index.json"
        if single.is_file():
            return load_file(str(single))
        if variant and stem.endswith(f".{variant}") and not index.is_file():
            index = path / f"diffusion_pytorch_model.safetensors.index.{variant}.json"  # diffusers' own name for a sharded variant
        if index.is_file():
            shards = sorted(set(json.loads(index.read_text())["weight_map"].values()))
            sd: Dict[str, torch.Tensor] = {}
            for shard in shards:
                sd.update(load_file(str(path / shard)))
            return sd
    raise FileNotFoundError(f"no {' / '.join(s_ + '.safetensors' for s_ in stems)} (or sharded index) under {path}")


def write_synthetic_checkpoint(model_dir, unet_cfg: UNetConfig = None, vae_cfg: VAEConfig = None, seed: int = 0,
                               prediction_type: str = "epsilon", device="cpu") -> str:
    """Write a diffusers-layout checkpoint directory with seeded random weights (the layout `load_pipelines` /
    `Diffuman4DPipeline.from_pretrained` read: sampling_utils.py:28-33):
        model_index.json, unet/{config.json, diffusion_pytorch_model.safetensors}, vae/{...}, scheduler/scheduler_config.json
    Used by tools/e2e_demo.py and the loader tests -- there is no public checkpoint in the build environment."""
    import json
    from dataclasses import asdict
    from pathlib import Path

    from safetensors.torch import save_file

    unet_cfg, vae_cfg = unet_cfg or UNetConfig(), vae_cfg or VAEConfig()
    root = Path(model_dir)
    for sub in ("unet", "vae", "scheduler"):
        (root / sub).mkdir(parents=True, exist_ok=True)
    (root / "model_index.json").write_text(json.dumps({
        "_class_name": "Diffuman4DPipeline",
        "unet": ["src.diffusers.models.unets.unet_multiview_condition", "UNetMultiviewConditionModel"],
        "vae": ["diffusers", "AutoencoderKL"], "scheduler": ["diffusers", "DDIMScheduler"]}, indent=1))
    (root / "unet" / "config.json").write_text(json.dumps(dict(asdict(unet_cfg), _class_name="UNetMultiviewConditionModel"), indent=1))
    (root / "vae" / "config.json").write_text(json.dumps(dict(asdict(vae_cfg), _class_name="AutoencoderKL"), indent=1))
    (root / "scheduler" / "scheduler_config.json").write_text(json.dumps({
        "_class_name": "DDIMScheduler", "num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012,
        "beta_schedule": "scaled_linear", "clip_sample": False, "set_alpha_to_one": False, "steps_offset": 1,
        "prediction_type": prediction_type, "timestep_spacing": "leading"}, indent=1))
    usd = {k: v.cpu().contiguous() for k, v in random_state_dict(unet_param_shapes(unet_cfg), seed, device).items()}
    save_file(usd, str(root / "unet" / "diffusion_pytorch_model.safetensors"))
    vsd = {k: v.cpu().contiguous() for k, v in random_state_dict(vae_param_shapes(vae_cfg), seed + 1, device).items()}
    save_file(vsd, str(root / "vae" / "diffusion_pytorch_model.safetensors"))
    return str(root)
