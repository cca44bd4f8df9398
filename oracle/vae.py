"""Oracle SD AutoencoderKL (diffusers==0.33.1 restatement).  TEST INFRASTRUCTURE ONLY.

Call sites in the reference: ``pipeline_diffuman4d.py:47-72`` (encode/decode, micro-batch 8),
``:52`` (``latent_dist.sample()``), ``:139`` (vae_scale_factor from block_out_channels).
Key names follow diffusers' AutoencoderKL state_dict.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .unet import Downsample2D, ResnetBlock2D, Upsample2D


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    @staticmethod
    def tiny(**kw) -> "VAEConfig":
        base = dict(block_out_channels=(32, 32, 64, 64), norm_num_groups=8)
        base.update(kw)
        return VAEConfig(**base)


class VAEAttention(nn.Module):
    """Single-head attention of the VAE mid block: GroupNorm on NCHW input, q/k/v with bias,
    ``residual_connection=True`` (AttnProcessor2_0, 4-D input path)."""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6, affine=True)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        y = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(y), self.to_k(y), self.to_v(y)
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0].to(q.dtype)
        o = self.to_out[0](o)
        return o.transpose(-1, -2).reshape(b, c, h, w) + x


class VAEMid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, None, groups, 1e-6) for _ in range(2)])
        self.attentions = nn.ModuleList([VAEAttention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=0)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, co in enumerate(boc):
            self.down_blocks.append(_EncBlock(c, co, cfg.layers_per_block, g, i != len(boc) - 1))
            c = co
        self.mid_block = VAEMid(c, g)
        self.conv_norm_out = nn.GroupNorm(g, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        rboc = list(reversed(boc))
        self.conv_in = nn.Conv2d(cfg.latent_channels, rboc[0], 3, padding=1)
        self.mid_block = VAEMid(rboc[0], g)
        self.up_blocks = nn.ModuleList()
        c = rboc[0]
        for i, co in enumerate(rboc):
            self.up_blocks.append(_DecBlock(c, co, cfg.layers_per_block + 1, g, i != len(boc) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(g, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)

    @property
    def scale_factor(self) -> int:
        return 2 ** (len(self.cfg.block_out_channels) - 1)

    def moments(self, x):
        return self.quant_conv(self.encoder(x))

    def sample_posterior(self, moments, noise):
        """DiagonalGaussianDistribution.sample with the noise injected explicitly."""
        mean, logvar = moments.chunk(2, dim=1)
        logvar = logvar.clamp(-30.0, 20.0)
        return mean + torch.exp(0.5 * logvar) * noise.to(mean.dtype)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))
