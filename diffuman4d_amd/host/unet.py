"""HIP-backed UNetMultiviewConditionModel (inference only).

Host-side mirror of the reference's model interface
(``/root/reference/src/diffusers/models/unets/unet_multiview_condition.py:49-598``): same config
fields, same ``state_dict`` key names on load, same ``forward(sample, timestep, skeletons, domains,
num_frames)`` meaning.  All arithmetic runs in libdm4d.so (``ops``); torch only owns the buffers.

Layout: activations are NHWC / token-major ``[B, H, W, C]`` bf16 end to end, so
  * the transformer's NCHW<->[B,HW,C] permutes (transformer_multiview.py:157-160,209-216) and
  * the 3-D attention frame folding + ``.contiguous()`` copies (attention.py:69-71,81-83)
are no-ops, and the up-block ``torch.cat`` (unet_multiview_blocks.py:667) is fused into the
consumers (two-source GroupNorm / split-K shortcut GEMM).
"""
from __future__ import annotations

import os

import json
from dataclasses import dataclass, fields
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops

BF16 = torch.bfloat16
F16 = torch.float16
PRECISIONS = ("fast", "parity", "fp16")
H16_CONV1_F16 = os.environ.get("DM4D_H16_CONV1_F16", "1") != "0"  # fp16 precision: conv1 -> fp16 -> norm2 (off: fp32 in between; A/B, tests)


def check_precision(precision: str) -> None:
    if precision not in PRECISIONS:
        raise ValueError(f"Unsupported precision: {precision}. Supported values are 'fast', 'parity' and 'fp16'.")


@dataclass
class UNetConfig:
    """Fields of ``unet/config.json`` used on this path (unet_multiview_condition.py:149-212)."""

    in_channels: int = 15
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlockMultiview", "CrossAttnDownBlockMultiview",
                                         "CrossAttnDownBlockMultiview", "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "CrossAttnUpBlockMultiview", "CrossAttnUpBlockMultiview",
                                       "CrossAttnUpBlockMultiview")
    layers_per_block: int = 2
    attention_head_dim: Tuple[int, ...] = (5, 10, 20, 20)  # number of heads (upstream naming quirk, :222-228)
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

    @classmethod
    def from_dict(cls, d: Dict) -> "UNetConfig":
        names = {f.name for f in fields(cls)}
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in names}
        return cls(**kw)

    def heads(self, i: int) -> int:
        a = self.attention_head_dim
        return a if isinstance(a, int) else a[i]


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class _Weights:
    """Converts a diffusers-style state_dict into kernel layouts on the device.
    parity (precision="parity", include/dm4d.h "Parity precision"): matrices are duplicated along K -- [W | W] per convolution
    tap -- to meet two-term activation operands [hi | lo]; vectors (biases, norm scales) are the same bf16 tensors.
    h16 (precision="fp16", include/dm4d.h "fp16 precision"): matrices and vectors are held in fp16 -- the values of the pipeline's
    weight dtype `wdtype` (torch_dtype: bf16 values are exact in fp16 down to 2^-14, fp16 values trivially).
    wide = parity or h16: the tensors BETWEEN kernels are fp32."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, parity: bool = False, h16: bool = False, wdtype=BF16):
        self.sd, self.device, self.parity, self.h16 = sd, device, parity, h16
        self.wide = parity or h16
        self.wdtype = wdtype if h16 else BF16  # the fast and parity precisions compute on bf16 weights whatever files were read
        self.used = set()

    def mat(self, w: torch.Tensor, taps: int = 1) -> torch.Tensor:
        """fp32 / bf16 / fp16 host matrix [N, taps * C] -> device bf16 (duplicated along K in parity precision) or fp16 (h16)."""
        w = w.to(self.wdtype)
        if self.h16:
            w = self._to_f16(w)
        elif self.parity:
            w = ops.dup_k(w, taps)
        return w.to(self.device).contiguous()

    def host_vec(self, v: torch.Tensor) -> torch.Tensor:
        """A bias / norm parameter assembled on the host -> the device vector the kernels of this precision read."""
        v = v.to(self.wdtype)
        return (self._to_f16(v) if self.h16 else v.to(BF16)).to(self.device).contiguous()

    @staticmethod
    def _to_f16(w: torch.Tensor) -> torch.Tensor:
        """The checkpoint's values in fp16.  bf16 values are exact in fp16 from 2^-14 up (below that they keep fewer bits, with an absolute
        error of at most 2^-25); a value beyond fp16's range cannot be held at all and is refused here, at load, by name of the condition
        rather than as an inf in the first matrix product."""
        h = w.to(F16)
        if not bool(torch.isfinite(h).all()):
            raise ValueError(f"precision 'fp16': a checkpoint tensor holds |w| = {float(w.float().abs().max()):.4g} > 65504, outside the fp16 "
                             "range (use model.precision=fast or parity for this checkpoint)")
        return h

    def get(self, key: str) -> torch.Tensor:
        if key not in self.sd:
            raise KeyError(f"UNet checkpoint is missing '{key}'")
        self.used.add(key)
        return self.sd[key]

    def vec(self, key: str) -> torch.Tensor:
        return self.host_vec(self.get(key))

    def linear(self, key: str) -> torch.Tensor:
        w = self.get(key)
        if w.ndim == 4:  # 1x1 conv used as a linear
            w = w.reshape(w.shape[0], w.shape[1])
        return self.mat(w)

    def conv3(self, key: str, cin_pad: Optional[int] = None, cout_pad: Optional[int] = None) -> torch.Tensor:
        """[Cout, Cin, 3, 3] -> [Cout, 9*Cin] with K ordered (ky, kx, ci); optional zero padding."""
        w = self.get(key).float()
        co, ci = w.shape[0], w.shape[1]
        cin_pad, cout_pad = cin_pad or ci, cout_pad or co
        wp = torch.zeros(cout_pad, 3, 3, cin_pad)
        wp[:co, :, :, :ci] = w.permute(0, 2, 3, 1)
        return self.mat(wp.reshape(cout_pad, 9 * cin_pad), taps=9)


class _PoseEncoder:
    """pose_encoder.py:11-54: eight thin conv+SiLU layers (direct-conv kernel), a 1x1 projection and a learned scale.
    Input: skeleton images NHWC bf16 [B, 8h, 8w, 4] (3 channels + one zero pad); output [B, h, w, C0].
    Parity precision: fp32 images, filters and layer outputs (the direct kernel's fp32 FMA chain with nothing rounded), the projection
    as a two-term GEMM, fp32 features."""

    LAYERS = ((3, 3, 3, 1), (3, 16, 4, 2), (16, 16, 3, 1), (16, 32, 4, 2), (32, 32, 3, 1), (32, 64, 4, 2),
              (64, 64, 3, 1), (64, 128, 3, 1))  # (Cin, Cout, kernel, stride), padding 1

    def __init__(self, W: _Weights, pfx: str):
        self.layers = []
        for n, (ci, co, k, s) in enumerate(self.LAYERS):
            w = W.get(pfx + f"conv_layers.{2 * n}.weight").float()
            b = W.get(pfx + f"conv_layers.{2 * n}.bias").float()
            if tuple(w.shape) != (co, ci, k, k):
                raise ValueError(f"pose_encoder.conv_layers.{2 * n}: unexpected weight shape {tuple(w.shape)}")
            cip, cop = _round_up(ci, 4), _round_up(co, 4)
            wp, bp = torch.zeros(cop, k, k, cip), torch.zeros(cop)
            wp[:co, :, :, :ci] = w.permute(0, 2, 3, 1)
            bp[:co] = b
            dt = torch.float32 if W.wide else BF16
            self.layers.append((wp.to(W.wdtype).reshape(cop, k * k * cip).to(W.device, dt).contiguous(), bp.to(W.wdtype).to(W.device, dt), k, s))
        self.wide, self.h16 = W.wide, W.h16  # wide: fp32 tensors between kernels (parity and fp16 precisions)
        self.proj_w, self.proj_b = W.linear(pfx + "final_proj.weight"), W.vec(pfx + "final_proj.bias")
        self.scale = float(W.get(pfx + "scale").to(W.wdtype).float().reshape(-1)[0])

    def __call__(self, x: torch.Tensor, batch: int = 16) -> torch.Tensor:
        if x.shape[-1] != 4:
            raise ValueError("pose encoder input must be NHWC with 4 (3 + pad) channels")
        outs = []
        for xb in x.split(batch):
            y = xb.float().contiguous() if self.wide else xb.contiguous()
            for (w, b, k, s) in self.layers:
                y = ops.conv2d_direct(y, w, ksize=k, bias=b, stride=s, pad=1, silu=True)
            B, h, wd, C = y.shape
            y = y.view(B * h * wd, C)
            if self.wide:
                y = ops.split(y, h16=self.h16)
            outs.append(ops.gemm(y, self.proj_w, bias=self.proj_b, out_scale=self.scale, out_f32=self.wide).view(B, h, wd, -1))
        return outs[0] if len(outs) == 1 else torch.cat(outs)


class _Resnet:
    def __init__(self, W: _Weights, pfx: str, groups: int, eps: float, scale: float, temb_list: List):
        self.groups, self.eps, self.scale, self.wide, self.h16 = groups, eps, scale, W.wide, W.h16
        self.n1w, self.n1b = W.vec(pfx + "norm1.weight"), W.vec(pfx + "norm1.bias")
        self.c1w, self.c1b = W.conv3(pfx + "conv1.weight"), W.vec(pfx + "conv1.bias")
        self.n2w, self.n2b = W.vec(pfx + "norm2.weight"), W.vec(pfx + "norm2.bias")
        self.c2w, self.c2b = W.conv3(pfx + "conv2.weight"), W.vec(pfx + "conv2.bias")
        self.cout = self.c1w.shape[0]
        self.has_sc = (pfx + "conv_shortcut.weight") in W.sd
        if self.has_sc:
            self.scw, self.scb = W.linear(pfx + "conv_shortcut.weight"), W.vec(pfx + "conv_shortcut.bias")
        # time_emb_proj of every resnet is batched into one GEMM per forward (see UNet._temb)
        self.t_off = sum(t[0].shape[0] for t in temb_list)
        temb_list.append((W.get(pfx + "time_emb_proj.weight"), W.get(pfx + "time_emb_proj.bias")))

    def __call__(self, x: torch.Tensor, tproj: torch.Tensor, skip: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Fast precision: bf16 tensors throughout.  Parity precision: x, skip, tproj and the result are fp32; the GroupNorm
        outputs are two-term operands (ops.groupnorm dispatches on the dtype), the convolutions take them against duplicated weights."""
        P = self.wide
        B, H, Wd = x.shape[:3]
        raw = None
        if self.h16 and self.has_sc:  # the shortcut's operand (fp16 of x | skip) comes out of the GroupNorm pass that reads the same tensor
            h, raw = ops.groupnorm(x, self.n1w, self.n1b, self.groups, self.eps, x2=skip, silu=True, raw_out=True)
        else:
            h = ops.groupnorm(x, self.n1w, self.n1b, self.groups, self.eps, x2=skip, silu=True)
        # fp16 precision: conv1's fp32 sum (+ the fp32 time-embedding row) has ONE reader, norm2, whose output is rounded to fp16 anyway:
        # it is rounded to fp16 once in the epilogue and norm2 reads two bytes per element (H16_CONV1_F16; measured on the whole-task
        # cases)
        h = ops.conv3x3(h, self.c1w, bias=self.c1b, rowbias=tproj[:, self.t_off:self.t_off + self.cout],
                        out_f32=P and not (self.h16 and H16_CONV1_F16))
        h = ops.groupnorm(h, self.n2w, self.n2b, self.groups, self.eps, silu=True)
        if self.has_sc:
            M = B * H * Wd
            if raw is not None:
                sc = ops.gemm(raw.view(M, -1), self.scw, bias=self.scb, out_f32=True)
            elif P:
                sc = ops.gemm(ops.split(x.view(M, -1), skip.view(M, -1) if skip is not None else None, h16=self.h16), self.scw, bias=self.scb,
                              out_f32=True)
            else:
                sc = ops.gemm(x.view(M, -1), self.scw, a2=skip.view(M, -1) if skip is not None else None, bias=self.scb)
            sc = sc.view(B, H, Wd, self.cout)
        else:
            assert skip is None
            sc = x
        return ops.conv3x3(h, self.c2w, bias=self.c2b, residual=sc, out_scale=1.0 / self.scale, out_f32=P)


class _TransformerBlock:
    def __init__(self, W: _Weights, pfx: str, heads: int):
        self.heads, self.wide, self.h16 = heads, W.wide, W.h16
        self.n1w, self.n1b = W.vec(pfx + "norm1.weight"), W.vec(pfx + "norm1.bias")
        q, k, v = (W.get(pfx + f"attn1.to_{n}.weight") for n in "qkv")
        self.scale = (q.shape[0] // heads) ** -0.5
        if not W.wide:
            # SDPA's q * scale (and the exp -> exp2 factor) folded into the bias-free to_q rows in fp32, before the single
            # bf16 rounding: the attention kernel then needs no per-score multiply (dm4d_attention_qscaled_kv_bf16).
            # Parity precision keeps the checkpoint's rows (exact in bf16) and scales the fp32 scores in the kernel; the fp16
            # precision keeps them too and multiplies the fp32 Q by the factor in the projection's epilogue, before its one rounding.
            q = q.float() * (self.scale * ops.LOG2E)
        self.qkv = W.mat(torch.cat([q.float(), k.float(), v.float()], dim=0))  # fused [3C, C], bias-free
        self.ow, self.ob = W.linear(pfx + "attn1.to_out.0.weight"), W.vec(pfx + "attn1.to_out.0.bias")
        self.n3w, self.n3b = W.vec(pfx + "norm3.weight"), W.vec(pfx + "norm3.bias")
        w1, b1 = W.linear(pfx + "ff.net.0.proj.weight"), W.vec(pfx + "ff.net.0.proj.bias")
        w2, b2 = W.linear(pfx + "ff.net.2.weight"), W.vec(pfx + "ff.net.2.bias")
        self.ff = (w1, b1, w2, b2) if (W.wide and not W.h16) else ops.FeedForward(w1, b1, w2, b2)
        if (pfx + "attn2.to_q.weight") in W.sd:
            raise NotImplementedError(f"unet checkpoint holds '{pfx}attn2.*' (a cross-attention block, i.e. unet/config.json: "
                                      "cross_attention_dim is not null): not supported -- the reference never passes encoder_hidden_states "
                                      "(pipeline_diffuman4d.py:398-405)")

    def __call__(self, h: torch.Tensor, batch: int, seq: int, shard=None, operand_out: bool = False) -> torch.Tensor:
        """h [M, C] token-major; attention over `batch` sequences of `seq` tokens (attention.py:68-90).
        shard (parallel.FrameShard): `seq` is this rank's share of the frame-folded sequence; K/V are all-gathered.
        operand_out (wide precisions): the result leaves as the OPERAND of the next contraction (the transformer's proj_out, when this is
        its last block) instead of an fp32 tensor: the same one rounding ops.split would apply to the stored fp32 sum, without storing it."""
        if self.wide:
            return self._call_wide(h, batch, seq, shard, operand_out)
        C = h.shape[1]
        n = ops.layernorm(h, self.n1w, self.n1b, 1e-5)
        if shard is None:
            qkv = ops.gemm(n, self.qkv)
            a = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch, self.heads, seq, q_scaled=True)
        else:
            kv = ops.gemm(n, self.qkv[C:])  # [M, 2C] contiguous so the collective needs no repack
            pending = shard.gather_kv_start(kv.view(batch, seq, 2 * C))
            q = ops.gemm(n, self.qkv[:C])  # runs while the K|V blocks travel
            kvg = shard.gather_kv_finish(pending).view(batch * shard.world * seq, 2 * C)
            a = ops.attention(q, kvg[:, :C], kvg[:, C:], batch, self.heads, seq, kv_seq=shard.world * seq, q_scaled=True)
        # attention output projection + residual, norm3, feed-forward + residual: one launch at C = 320 (level 0); gemm(residual),
        # layernorm, gemm(GEGLU), gemm(residual) elsewhere
        return self.ff.after_attention(a, self.ow, self.ob, h, (self.n3w, self.n3b, 1e-5))

    def _call_wide(self, h: torch.Tensor, batch: int, seq: int, shard=None, operand_out: bool = False) -> torch.Tensor:
        """The same block on an fp32 residual stream: LayerNorm -> operand; QKV projection -> operand planes; attention -> operand;
        output projection + fp32 residual; LayerNorm; GEGLU -> operand; projection + residual.  Parity precision: two-term operands,
        three MFMA terms per attention product.  fp16 precision: one fp16 plane each, Q pre-scaled in the projection's epilogue, the
        fast precision's attention loop on fp16 operands.  With `shard` the K | V planes are all-gathered as in the fast precision."""
        C = h.shape[1]
        n = ops.layernorm(h, self.n1w, self.n1b, 1e-5)
        if self.h16:
            qs = self.scale * ops.LOG2E
            if shard is None:
                qkv = ops.gemm(n, self.qkv, scale_cols=C, col_scale=qs)  # fp16 [M, 3C]
                a = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch, self.heads, seq, q_scaled=True)
            else:
                kv = ops.gemm(n, self.qkv[C:])  # fp16 [M, 2C]
                pending = shard.gather_kv_start(kv.view(batch, seq, 2 * C))
                q = ops.gemm(n, self.qkv[:C], scale_cols=C, col_scale=qs)
                kvg = shard.gather_kv_finish(pending).view(batch * shard.world * seq, 2 * C)
                a = ops.attention(q, kvg[:, :C], kvg[:, C:], batch, self.heads, seq, kv_seq=shard.world * seq, q_scaled=True)
        elif shard is None:
            qkv = ops.gemm(n, self.qkv, split_out=True)  # [M, 6C] = [q_hi | k_hi | v_hi | q_lo | k_lo | v_lo]
            a = ops.attention_split(qkv, batch, self.heads, seq, self.scale)
        else:
            kv = ops.gemm(n, self.qkv[C:], split_out=True)  # [M, 4C] = [k_hi | v_hi | k_lo | v_lo]
            pending = shard.gather_kv_start(kv.view(batch, seq, 4 * C))
            q = ops.gemm(n, self.qkv[:C], split_out=True)  # [M, 2C] = [q_hi | q_lo]
            kvg = shard.gather_kv_finish(pending).view(batch * shard.world * seq, 4 * C)
            a = ops.attention_split(None, batch, self.heads, seq, self.scale, q=q, kv=kvg, kv_seq=shard.world * seq)
        if self.h16:  # the block's tail: one launch at C = 320 (level 0), four elsewhere (ops.FeedForward.after_attention_f16)
            return self.ff.after_attention_f16(a, self.ow, self.ob, h, (self.n3w, self.n3b, 1e-5), out_f32=not operand_out)
        w1, b1, w2, b2 = self.ff
        h = ops.gemm(a, self.ow, bias=self.ob, residual=h, out_f32=True)
        f = ops.gemm(ops.layernorm(h, self.n3w, self.n3b, 1e-5), w1, bias=b1, geglu=True, split_out=True)
        return ops.gemm(f, w2, bias=b2, residual=h, out_f32=not operand_out, split_out=operand_out)


class _Transformer:
    """TransformerMultiviewModel (transformer_multiview.py:34-232), continuous input."""

    def __init__(self, W: _Weights, pfx: str, heads: int, groups: int):
        self.groups, self.wide, self.h16 = groups, W.wide, W.h16
        self.nw, self.nb = W.vec(pfx + "norm.weight"), W.vec(pfx + "norm.bias")
        self.piw, self.pib = W.linear(pfx + "proj_in.weight"), W.vec(pfx + "proj_in.bias")
        self.pow, self.pob = W.linear(pfx + "proj_out.weight"), W.vec(pfx + "proj_out.bias")
        self.blocks = []
        i = 0
        while (pfx + f"transformer_blocks.{i}.norm1.weight") in W.sd:
            self.blocks.append(_TransformerBlock(W, pfx + f"transformer_blocks.{i}.", heads))
            i += 1
        if (self.pob.shape[0] // heads) != 64:
            raise NotImplementedError(
                f"unet/config.json: block_out_channels / attention_head_dim give a head dimension of {self.pob.shape[0] // heads} at "
                f"'{pfx[:-1]}' ({self.pob.shape[0]} channels over {heads} heads -- `attention_head_dim` is the NUMBER of heads, "
                "unet_multiview_condition.py:222-228); the HIP attention kernels are built for head dimension 64 (SD-2.1 geometry)")

    def __call__(self, x: torch.Tensor, num_frames: int, shard=None) -> torch.Tensor:
        B, H, Wd, C = x.shape
        M, HW = B * H * Wd, H * Wd
        P = self.wide
        n = ops.groupnorm(x, self.nw, self.nb, self.groups, 1e-6, silu=False)  # eps 1e-6: transformer_multiview.py:43-45
        h = ops.gemm(n.view(M, -1), self.piw, bias=self.pib, out_f32=P)
        for i, blk in enumerate(self.blocks):  # wide precisions: the last block hands proj_out its operand directly
            h = blk(h, B // num_frames, num_frames * HW, shard, operand_out=P and i == len(self.blocks) - 1)
        if P and not self.blocks:
            h = ops.split(h, h16=self.h16)
        return ops.gemm(h, self.pow, bias=self.pob, residual=x.view(M, C), out_f32=P).view(B, H, Wd, C)


class UNetMultiviewConditionModel:
    """Inference-only, HIP-backed.  ``forward`` takes/returns NHWC bf16 (see ``Diffuman4DPipeline``)."""

    IN_PAD = 32  # conv_in input channels are zero-padded to whole 32-wide K slabs: 32, or 64 for in_channels in 33..64 (per instance)

    def __init__(self, config: UNetConfig, state_dict: Dict[str, torch.Tensor], device="cuda", precision: str = "fast",
                 weight_dtype=BF16):
        """precision: "fast" (bf16 tensors, bf16 MFMA operands), "parity" (fp32 tensors between kernels, two-term bf16 operands:
        include/dm4d.h "Parity precision") or "fp16" (fp32 tensors between kernels, single-term fp16 operands: "fp16 precision");
        the last two meet north_star's 1e-3 on decoded RGB against the fp32 reference path.
        weight_dtype (fp16 precision only): the dtype the reference would hold the weights in (torch_dtype, sampling_utils.py:27-29)."""
        cfg = self.config = config
        self.device = torch.device(device)
        check_precision(precision)
        self.precision, self.parity, self.h16 = precision, precision == "parity", precision == "fp16"
        self.wide = self.parity or self.h16  # fp32 tensors between kernels
        if cfg.cross_attention_dim is not None:
            raise NotImplementedError(
                f"unet/config.json: cross_attention_dim = {cfg.cross_attention_dim}; this path builds self-attention blocks only -- the "
                "reference's UNet forward takes no encoder_hidden_states and its pipeline never passes any "
                "(unet_multiview_condition.py:290-291, :501-509; pipeline_diffuman4d.py:398-405), so cross_attention_dim must be null")
        if cfg.in_channels > 64:
            raise NotImplementedError(f"unet/config.json: in_channels = {cfg.in_channels}; conv_in is built for up to 64 input channels "
                                      "(the reference's pipeline assembles 15, or 11 with enable_pose_encoder: pipeline_diffuman4d.py:389-395)")
        self.IN_PAD = 32 if cfg.in_channels <= 32 else 64
        W = _Weights(state_dict, self.device, self.parity, self.h16, weight_dtype)
        boc = cfg.block_out_channels
        g, eps = cfg.norm_num_groups, cfg.norm_eps
        temb_list: List = []
        self.conv_in_w = W.conv3("conv_in.weight", cin_pad=self.IN_PAD)
        self.conv_in_b = W.vec("conv_in.bias")
        self.te = [W.linear("time_embedding.linear_1.weight"), W.vec("time_embedding.linear_1.bias"),
                   W.linear("time_embedding.linear_2.weight"), W.vec("time_embedding.linear_2.bias")]
        self.pose_encoder = _PoseEncoder(W, "pose_encoder.") if cfg.enable_pose_encoder else None
        self.tpe = None
        if cfg.enable_tem_embeds:
            self.tpe = [W.linear("temporal_pos_embed.linear_1.weight"), W.vec("temporal_pos_embed.linear_1.bias"),
                        W.linear("temporal_pos_embed.linear_2.weight"), W.vec("temporal_pos_embed.linear_2.bias")]
        self.down = []
        for i, t in enumerate(cfg.down_block_types):
            p = f"down_blocks.{i}."
            has_attn = t != "DownBlock2D"
            res = [_Resnet(W, p + f"resnets.{j}.", g, eps, 1.0, temb_list)
                   for j in range(cfg.layers_per_block)]
            att = [_Transformer(W, p + f"attentions.{j}.", cfg.heads(i), g) for j in range(cfg.layers_per_block)] if has_attn else None
            ds = None
            if i != len(boc) - 1:
                ds = (W.conv3(p + "downsamplers.0.conv.weight"), W.vec(p + "downsamplers.0.conv.bias"))
            self.down.append((res, att, ds))
        mp = "mid_block."
        self.mid = ([_Resnet(W, mp + f"resnets.{j}.", g, eps, cfg.mid_block_scale_factor, temb_list) for j in range(2)],
                    _Transformer(W, mp + "attentions.0.", cfg.heads(len(boc) - 1), g))
        self.up = []
        for i, t in enumerate(cfg.up_block_types):
            p = f"up_blocks.{i}."
            has_attn = t != "UpBlock2D"
            n = cfg.layers_per_block + 1
            res = [_Resnet(W, p + f"resnets.{j}.", g, eps, 1.0, temb_list)
                   for j in range(n)]
            att = [_Transformer(W, p + f"attentions.{j}.", cfg.heads(len(boc) - 1 - i), g) for j in range(n)] if has_attn else None
            us = None
            if i != len(boc) - 1:
                us = ops.Upsampler(W.conv3(p + "upsamplers.0.conv.weight"), W.vec(p + "upsamplers.0.conv.bias"), parity=self.parity,
                                   h16=self.h16)
            self.up.append((res, att, us))
        self.no_w, self.no_b = W.vec("conv_norm_out.weight"), W.vec("conv_norm_out.bias")
        self.conv_out_w, self.conv_out_b = W.conv3("conv_out.weight"), W.vec("conv_out.bias")
        # one [sum(Cout), 4*C0] weight for all time_emb_proj layers
        self.tproj_w = W.mat(torch.cat([t[0] for t in temb_list], dim=0))
        self.tproj_b = W.host_vec(torch.cat([t[1] for t in temb_list], dim=0))
        unused = [k for k in state_dict if k not in W.used and not k.startswith("time_proj")
                  and "time_emb_proj" not in k and ".attn1.to_" not in k]
        if unused:
            raise KeyError(f"unexpected keys in UNet checkpoint (strict load): {unused[:8]}{'...' if len(unused) > 8 else ''}")

    # -- loading --------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, device="cuda", variant: Optional[str] = None, precision: str = "fast",
                        weight_dtype=BF16) -> "UNetMultiviewConditionModel":
        """``path`` = the ``unet/`` folder of a diffusers checkpoint directory; ``variant="fp16"`` reads the
        ``*.fp16.safetensors`` file (the fast and parity precisions convert the weights to bf16 either way; the fp16 precision holds
        the values of ``weight_dtype`` in fp16)."""
        from .weights import load_component_state_dict
        path = Path(path)
        cfg = UNetConfig.from_dict(json.loads((path / "config.json").read_text()))
        return cls(cfg, load_component_state_dict(path, variant), device, precision, weight_dtype)

    # -- forward --------------------------------------------------------------------------------
    def _temb(self, timestep: torch.Tensor, domains: Sequence[str], num_frames: int, shard=None) -> torch.Tensor:
        cfg, P = self.config, self.wide
        c0 = cfg.block_out_channels[0]
        op = (lambda t, **kw: ops.split(t, h16=self.h16, **kw)) if P else (lambda t, silu=False: ops.silu(t) if silu else t)  # fp32 -> operand
        t_emb = ops.timestep_embedding(timestep.to(self.device, torch.float32), c0, cfg.flip_sin_to_cos, float(cfg.freq_shift), out_f32=P)
        emb = ops.gemm(ops.gemm(op(t_emb), self.te[0], bias=self.te[1], silu=True, split_out=P), self.te[2], bias=self.te[3], out_f32=P)
        if self.tpe is not None:  # unet_multiview_condition.py:523-546
            if len(domains) * num_frames != emb.shape[0]:
                raise ValueError(f"num_frames: {num_frames} * len(domains): {len(domains)} != len(emb): {emb.shape[0]}")
            idx = []
            world = shard.world if shard is not None else 1
            for d in domains:
                if d == "spatial":
                    full = torch.zeros(num_frames * world)
                elif d == "temporal":
                    full = torch.arange(num_frames * world // 2).repeat(2).float()
                else:
                    raise ValueError(f"Invalid domain for temporal embedding: {d}")
                idx.append(full if shard is None else full[shard.local_frames(num_frames * world)])
            f_emb = ops.timestep_embedding(torch.cat(idx).to(self.device), c0, True, 0.0, out_f32=P)
            emb = ops.gemm(ops.gemm(op(f_emb), self.tpe[0], bias=self.tpe[1], silu=True, split_out=P), self.tpe[2], bias=self.tpe[3],
                           residual=emb, out_f32=P)
        return ops.gemm(op(emb, silu=True), self.tproj_w, bias=self.tproj_b, out_f32=P)  # [B, sum Cout]

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: torch.Tensor, skeletons=None, domains: Sequence[str] = ("spatial",),
                num_frames: int = 1, shard=None, pose_features: Optional[torch.Tensor] = None,
                keep_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """sample [B, h, w, 32] NHWC bf16 (channels beyond in_channels zero); timestep [B]; -> [B, h, w, out_channels].
        precision "parity": sample is the two-term operand [B, h, w, 64] = [hi(32) | lo(32)] (ops.pack_model_input / ops.split of an
        fp32 sample), the result is fp32.  precision "fp16": sample is the fp16 operand [B, h, w, 32], the result is fp32.
        keep_rows (int64 [R], optional extension): batch rows whose output is wanted.  Every layer after the last 3-D
        attention is per-frame, so from there on only these rows are computed and the result has R rows (the pipeline
        discards the noise prediction of conditioning rows, pipeline_diffuman4d.py:413-421).
        With `shard` (parallel.FrameShard) sample/timestep hold this rank's frames and num_frames is the LOCAL count.
        enable_pose_encoder checkpoints (:551-552): pass `skeletons` [B, 8h, 8w, 4] NHWC (encoded here, as the
        reference does on every call) or `pose_features` [B, h, w, C0] computed once with ``self.pose_encoder``."""
        cfg, P = self.config, self.wide
        if sample.shape[-1] != (2 * self.IN_PAD if self.parity else self.IN_PAD):
            raise ValueError(f"sample must be NHWC with {self.IN_PAD} (padded) channels" + (" as a two-term operand [hi | lo]" if self.parity else ""))
        if sample.dtype != (F16 if self.h16 else BF16):
            raise ValueError(f"sample must be {'fp16' if self.h16 else 'bf16'} for precision '{self.precision}'")
        if sample.shape[0] % num_frames != 0:
            raise ValueError("batch must be a multiple of num_frames")
        tproj = self._temb(timestep, domains, num_frames, shard)
        if self.pose_encoder is not None:
            if pose_features is None:
                if skeletons is None:
                    raise ValueError("enable_pose_encoder: skeletons or pose_features are required")
                pose_features = self.pose_encoder(skeletons)
            if pose_features.shape[:3] != sample.shape[:3]:
                raise ValueError("pose features do not match the sample's batch / latent size")
            pose_features = pose_features.contiguous()
        else:
            pose_features = None
        x = ops.conv3x3(sample, self.conv_in_w, bias=self.conv_in_b, residual=pose_features, out_f32=P)
        skips = [x]
        nd = len(self.down)
        for i, (res, att, ds) in enumerate(self.down):
            is3d = att is not None and nd - i - 1 < cfg.num_3d_attn_blocks  # :560
            for j, r in enumerate(res):
                x = r(x, tproj)
                if att is not None:
                    x = att[j](x, num_frames if is3d else 1, shard if is3d else None)
                skips.append(x)
            if ds is not None:
                x = ops.conv3x3(ops.split(x, h16=self.h16) if P else x, ds[0], bias=ds[1], stride=2, pad=1, out_f32=P)
                skips.append(x)
        x = self.mid[0][0](x, tproj)
        x = self.mid[1](x, num_frames, shard)  # :570
        x = self.mid[0][1](x, tproj)
        def prune(x, tproj, skips):  # rows past the last frame-mixing layer that nobody reads are dropped
            return (x.index_select(0, keep_rows), tproj.index_select(0, keep_rows),
                    [sk.index_select(0, keep_rows) for sk in skips])

        last3d = max((i for i, (_, att, _) in enumerate(self.up) if att is not None and i < cfg.num_3d_attn_blocks), default=-1)
        if keep_rows is not None and last3d < 0:
            x, tproj, skips = prune(x, tproj, skips)
        for i, (res, att, us) in enumerate(self.up):
            is3d = att is not None and i < cfg.num_3d_attn_blocks  # :582
            for j, r in enumerate(res):
                x = r(x, tproj, skip=skips.pop())
                if att is not None:
                    x = att[j](x, num_frames if is3d else 1, shard if is3d else None)
            if keep_rows is not None and i == last3d:
                x, tproj, skips = prune(x, tproj, skips)
            if us is not None:
                x = us(x)
        x = ops.groupnorm(x, self.no_w, self.no_b, cfg.norm_num_groups, cfg.norm_eps, silu=True)
        return ops.conv3x3(x, self.conv_out_w, bias=self.conv_out_b, out_f32=P)

    __call__ = forward
