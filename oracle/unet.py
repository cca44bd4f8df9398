"""Oracle UNet: pure-torch restatement of ``UNetMultiviewConditionModel``.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows
  * ``src/diffusers/models/unets/unet_multiview_condition.py:213-443,501-598`` (wiring, forward)
  * ``src/diffusers/models/unets/unet_multiview_blocks.py:233-712``            (down / mid / up blocks)
  * ``src/diffusers/models/transformers/transformer_multiview.py:42-77,156-216``
  * ``src/diffusers/models/attention.py:38-153``                               (3-D frame folding)
  * ``src/diffusers/models/unets/pose_encoder.py``
and, for the building blocks that live in diffusers==0.33.1 (not vendored):
ResnetBlock2D, Downsample2D, Upsample2D, Transformer2DModel (continuous input),
BasicTransformerBlock(norm_type="layer_norm"), Attention + AttnProcessor2_0,
FeedForward(GEGLU), Timesteps, TimestepEmbedding.

Module / parameter names reproduce the diffusers ``state_dict`` keys so a real
checkpoint would load with ``strict=True``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn


@dataclass
class UNetConfig:
    """Mirror of the fields of ``unet/config.json`` that matter on this path
    (``unet_multiview_condition.py:149-212``)."""

    in_channels: int = 15
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = (
        "CrossAttnDownBlockMultiview",
        "CrossAttnDownBlockMultiview",
        "CrossAttnDownBlockMultiview",
        "DownBlock2D",
    )
    up_block_types: Tuple[str, ...] = (
        "UpBlock2D",
        "CrossAttnUpBlockMultiview",
        "CrossAttnUpBlockMultiview",
        "CrossAttnUpBlockMultiview",
    )
    layers_per_block: int = 2
    # upstream naming quirk (unet_multiview_condition.py:222-228): this is the NUMBER of heads
    attention_head_dim: Tuple[int, ...] = (5, 10, 20, 20)
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    cross_attention_dim: Optional[int] = None
    use_linear_projection: bool = True
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    num_3d_attn_blocks: int = 3
    enable_tem_embeds: bool = False
    enable_pose_encoder: bool = False
    mid_block_scale_factor: float = 1.0
    resnet_out_scale_factor: float = 1.0

    def heads(self, i: int) -> int:
        a = self.attention_head_dim
        return a if isinstance(a, int) else a[i]

    @staticmethod
    def tiny(**kw) -> "UNetConfig":
        """A small geometry with the same topology, for CPU-sized tests."""
        base = dict(
            block_out_channels=(64, 128, 128, 128),
            attention_head_dim=(1, 2, 2, 2),
            norm_num_groups=32,
        )
        base.update(kw)
        return UNetConfig(**base)


# --------------------------------------------------------------------------------------
# diffusers building blocks (restated)
# --------------------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int, flip_sin_to_cos: bool, freq_shift: float) -> torch.Tensor:
    """diffusers ``get_timestep_embedding`` (scale=1, max_period=10000); always fp32."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D, time_embedding_norm="default", dropout 0."""

    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5, output_scale_factor=1.0):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = output_scale_factor

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


class Downsample2D(nn.Module):
    """use_conv=True; padding=1 -> conv s2 p1; padding=0 -> F.pad(0,1,0,1) + conv s2 p0 (VAE)."""

    def __init__(self, channels, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Attention(nn.Module):
    """diffusers Attention + AttnProcessor2_0, self-attention only.

    UNet flavour: bias=False on q/k/v, out bias=True, no norm, no residual.
    """

    def __init__(self, query_dim, heads, dim_head, bias=False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, x):
        b, l, _ = x.shape
        q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)
        d = q.shape[-1] // self.heads
        q = q.view(b, l, self.heads, d).transpose(1, 2)
        k = k.view(b, l, self.heads, d).transpose(1, 2)
        v = v.view(b, l, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, l, self.heads * d).to(q.dtype)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class MultiviewTransformerBlock(nn.Module):
    """``src/diffusers/models/attention.py:17-153`` on top of BasicTransformerBlock
    (norm_type="layer_norm", cross_attention_dim=None => attn2/norm2 absent)."""

    def __init__(self, dim, heads, dim_head):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head, bias=False)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, num_frames=1):
        n = self.norm1(x)
        if num_frames > 1:  # attention.py:69-71  "(b t) hw c -> b (t hw) c"
            bt, hw, c = n.shape
            n = n.reshape(bt // num_frames, num_frames * hw, c)
        a = self.attn1(n)
        if num_frames > 1:  # attention.py:81-83
            a = a.reshape(bt, hw, c)
        x = a + x  # attention.py:90
        x = self.ff(self.norm3(x)) + x  # attention.py:129-149
        return x


class TransformerMultiviewModel(nn.Module):
    """``transformer_multiview.py:34-232`` (continuous input; GroupNorm eps is 1e-6, line 43-45)."""

    def __init__(self, heads, dim_head, in_channels, norm_num_groups=32, use_linear_projection=True, num_layers=1):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner)
            self.proj_out = nn.Linear(inner, in_channels)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, 1)
            self.proj_out = nn.Conv2d(inner, in_channels, 1)
        self.transformer_blocks = nn.ModuleList(
            [MultiviewTransformerBlock(inner, heads, dim_head) for _ in range(num_layers)]
        )

    def forward(self, x, num_frames=1):
        b, c, h, w = x.shape
        residual = x
        y = self.norm(x)
        if self.use_linear_projection:
            y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
            y = self.proj_in(y)
        else:
            y = self.proj_in(y)
            y = y.permute(0, 2, 3, 1).reshape(b, h * w, y.shape[1])
        for blk in self.transformer_blocks:
            y = blk(y, num_frames=num_frames)
        if self.use_linear_projection:
            y = self.proj_out(y)
            y = y.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous()
        else:
            y = y.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous()
            y = self.proj_out(y)
        return y + residual


class PoseEncoder(nn.Module):
    """``pose_encoder.py:11-54``."""

    def __init__(self, out_channels=320):
        super().__init__()
        chans = [(3, 3, 3, 1), (3, 16, 4, 2), (16, 16, 3, 1), (16, 32, 4, 2), (32, 32, 3, 1), (32, 64, 4, 2),
                 (64, 64, 3, 1), (64, 128, 3, 1)]
        layers: List[nn.Module] = []
        for ci, co, k, s in chans:
            layers += [nn.Conv2d(ci, co, k, stride=s, padding=1), nn.SiLU()]
        self.conv_layers = nn.Sequential(*layers)
        self.final_proj = nn.Conv2d(128, out_channels, 1)
        self.scale = nn.Parameter(torch.ones(1) * 2.0)

    def forward(self, x):
        return self.final_proj(self.conv_layers(x)) * self.scale


# --------------------------------------------------------------------------------------
# blocks (unet_multiview_blocks.py)
# --------------------------------------------------------------------------------------
class DownBlock(nn.Module):
    """CrossAttnDownBlockMultiview (:386-541) when ``heads`` is given, else DownBlock2D."""

    def __init__(self, cin, cout, temb, layers, groups, eps, heads, add_downsample, use_linear, scale):
        super().__init__()
        self.has_cross_attention = heads is not None
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps, scale) for i in range(layers)]
        )
        if self.has_cross_attention:
            self.attentions = nn.ModuleList(
                [TransformerMultiviewModel(heads, cout // heads, cout, groups, use_linear) for _ in range(layers)]
            )
        self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=1)]) if add_downsample else None

    def forward(self, x, temb, num_frames=1):
        outs = ()
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.has_cross_attention:
                x = self.attentions[i](x, num_frames=num_frames)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class MidBlock(nn.Module):
    """UNetMidBlockMultiviewCrossAttn (:233-383)."""

    def __init__(self, c, temb, groups, eps, heads, use_linear, scale):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps, scale) for _ in range(2)])
        self.attentions = nn.ModuleList([TransformerMultiviewModel(heads, c // heads, c, groups, use_linear)])

    def forward(self, x, temb, num_frames=1):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, num_frames=num_frames)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    """CrossAttnUpBlockMultiview (:544-712) when ``heads`` is given, else UpBlock2D."""

    def __init__(self, cin, cout, cprev, temb, layers, groups, eps, heads, add_upsample, use_linear, scale):
        super().__init__()
        self.has_cross_attention = heads is not None
        res = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout  # :581
            rin = cprev if i == 0 else cout  # :582
            res.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps, scale))
        self.resnets = nn.ModuleList(res)
        if self.has_cross_attention:
            self.attentions = nn.ModuleList(
                [TransformerMultiviewModel(heads, cout // heads, cout, groups, use_linear) for _ in range(layers)]
            )
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None

    def forward(self, x, res_tuple, temb, num_frames=1):
        for i, res in enumerate(self.resnets):
            skip = res_tuple[-1]
            res_tuple = res_tuple[:-1]
            x = torch.cat([x, skip], dim=1)  # :667
            x = res(x, temb)
            if self.has_cross_attention:
                x = self.attentions[i](x, num_frames=num_frames)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNetMultiviewConditionModel(nn.Module):
    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        boc = cfg.block_out_channels
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        if cfg.enable_tem_embeds:
            self.temporal_pos_embed = TimestepEmbedding(boc[0], temb)
            nn.init.zeros_(self.temporal_pos_embed.linear_2.weight)
            nn.init.zeros_(self.temporal_pos_embed.linear_2.bias)
        if cfg.enable_pose_encoder:
            self.pose_encoder = PoseEncoder(boc[0])
        g, eps, lin = cfg.norm_num_groups, cfg.norm_eps, cfg.use_linear_projection
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, t in enumerate(cfg.down_block_types):
            in_c, out_c = out_c, boc[i]
            heads = cfg.heads(i) if t != "DownBlock2D" else None
            self.down_blocks.append(
                DownBlock(in_c, out_c, temb, cfg.layers_per_block, g, eps, heads, i != len(boc) - 1, lin, 1.0)
            )
        self.mid_block = MidBlock(boc[-1], temb, g, eps, cfg.heads(len(boc) - 1), lin, cfg.mid_block_scale_factor)
        self.up_blocks = nn.ModuleList()
        rboc = list(reversed(boc))
        out_c = rboc[0]
        for i, t in enumerate(cfg.up_block_types):
            prev_c, out_c = out_c, rboc[i]
            in_c = rboc[min(i + 1, len(boc) - 1)]
            heads = cfg.heads(len(boc) - 1 - i) if t != "UpBlock2D" else None
            self.up_blocks.append(
                UpBlock(in_c, out_c, prev_c, temb, cfg.layers_per_block + 1, g, eps, heads, i != len(boc) - 1, lin, 1.0)
            )
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, sample, timestep, skeletons=None, domains: Sequence[str] = ("spatial",), num_frames: int = 1):
        cfg = self.cfg
        boc0 = cfg.block_out_channels[0]
        t_emb = timestep_embedding(timestep.expand(sample.shape[0]), boc0, cfg.flip_sin_to_cos, cfg.freq_shift)
        emb = self.time_embedding(t_emb.to(sample.dtype))
        if cfg.enable_tem_embeds:  # unet_multiview_condition.py:523-546
            if len(domains) * num_frames != len(emb):
                raise ValueError("num_frames * len(domains) != len(emb)")
            idx = []
            for d in domains:
                if d == "spatial":
                    idx.append(torch.zeros(num_frames))
                elif d == "temporal":
                    idx.append(torch.arange(num_frames // 2).repeat(2).float())
                else:
                    raise ValueError(f"Invalid domain for temporal embedding: {d}")
            f_emb = timestep_embedding(torch.cat(idx), boc0, True, 0)
            emb = emb + self.temporal_pos_embed(f_emb.to(emb.dtype))
        x = self.conv_in(sample)
        if cfg.enable_pose_encoder:
            x = x + self.pose_encoder(skeletons)
        res = (x,)
        n_down = len(self.down_blocks)
        for i, blk in enumerate(self.down_blocks):
            nf = num_frames if (blk.has_cross_attention and n_down - i - 1 < cfg.num_3d_attn_blocks) else 1  # :560
            x, outs = blk(x, emb, num_frames=nf)
            res += outs
        x = self.mid_block(x, emb, num_frames=num_frames)  # :570
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            skips, res = res[-n:], res[:-n]
            nf = num_frames if (blk.has_cross_attention and i < cfg.num_3d_attn_blocks) else 1  # :582
            x = blk(x, skips, emb, num_frames=nf)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return x


def init_unet_weights(model: nn.Module, seed: int = 0) -> None:
    """Deterministic random init that keeps activations O(1): default torch
    initialisers under a fixed seed, norm affine perturbed so that gamma/beta are exercised."""
    gen = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if p.ndim >= 2:
            fan_in = p[0].numel()
            p.data = torch.randn(p.shape, generator=gen) * (1.0 / math.sqrt(fan_in))
        elif name.endswith("scale"):
            p.data.fill_(2.0)
        elif "norm" in name and name.endswith("weight"):
            p.data = 1.0 + 0.1 * torch.randn(p.shape, generator=gen)
        else:
            p.data = 0.05 * torch.randn(p.shape, generator=gen)
