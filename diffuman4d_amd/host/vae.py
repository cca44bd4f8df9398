"""HIP-backed SD AutoencoderKL (encode / decode tiles of the pipeline).

Replaces the ``diffusers.AutoencoderKL`` calls of ``pipeline_diffuman4d.py:47-72`` (micro-batches of
8 images, posterior ``sample()``, ``scaling_factor``) and the ``VaeImageProcessor`` denormalisation
(:280-285).  Same kernels as the UNet: NHWC implicit-GEMM conv3x3, GroupNorm+SiLU, MFMA GEMM.  The
single-head d=512 mid-block attention is expressed with the GEMM kernel (QK^T, PV) + a row softmax.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, fields
from pathlib import Path
from typing import Dict, Optional, Tuple

import torch

from . import ops
from .unet import _Weights, check_precision

BF16 = torch.bfloat16
F32 = torch.float32
PAD = 32


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    @classmethod
    def from_dict(cls, d: Dict) -> "VAEConfig":
        names = {f.name for f in fields(cls)}
        return cls(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in names})


class _Res:
    """ResnetBlock2D without time embedding, eps 1e-6."""

    def __init__(self, W: _Weights, pfx: str, groups: int):
        self.g, self.wide, self.h16 = groups, W.wide, W.h16
        self.n1w, self.n1b = W.vec(pfx + "norm1.weight"), W.vec(pfx + "norm1.bias")
        self.c1w, self.c1b = W.conv3(pfx + "conv1.weight"), W.vec(pfx + "conv1.bias")
        self.n2w, self.n2b = W.vec(pfx + "norm2.weight"), W.vec(pfx + "norm2.bias")
        self.c2w, self.c2b = W.conv3(pfx + "conv2.weight"), W.vec(pfx + "conv2.bias")
        self.has_sc = (pfx + "conv_shortcut.weight") in W.sd
        if self.has_sc:
            self.scw, self.scb = W.linear(pfx + "conv_shortcut.weight"), W.vec(pfx + "conv_shortcut.bias")

    def __call__(self, x):
        P = self.wide  # fp32 tensors, operands out of the GroupNorms (ops dispatches on the dtypes)
        B, H, Wd, C = x.shape
        raw = None
        if self.h16 and self.has_sc:  # the shortcut's fp16 operand out of the GroupNorm pass that reads the same tensor
            h, raw = ops.groupnorm(x, self.n1w, self.n1b, self.g, 1e-6, silu=True, raw_out=True)
        else:
            h = ops.groupnorm(x, self.n1w, self.n1b, self.g, 1e-6, silu=True)
        h = ops.conv3x3(h, self.c1w, bias=self.c1b, out_f32=P)
        h = ops.groupnorm(h, self.n2w, self.n2b, self.g, 1e-6, silu=True)
        sc = x
        if self.has_sc:
            a = raw.view(-1, C) if raw is not None else (ops.split(x.view(-1, C), h16=self.h16) if P else x.view(-1, C))
            sc = ops.gemm(a, self.scw, bias=self.scb, out_f32=P).view(B, H, Wd, -1)
        return ops.conv3x3(h, self.c2w, bias=self.c2b, residual=sc, out_f32=P)


class _MidAttn:
    """Single-head attention, GroupNorm on the input, q/k/v with bias, residual connection."""

    QBLOCK_BYTES = 128 << 20  # fp32 logits of one query block

    def __init__(self, W: _Weights, pfx: str, groups: int):
        self.g, self.parity, self.h16 = groups, W.parity, W.h16
        self.nw, self.nb = W.vec(pfx + "group_norm.weight"), W.vec(pfx + "group_norm.bias")
        ws = [W.get(pfx + f"to_{n}.weight") for n in "qkv"]
        bs = [W.get(pfx + f"to_{n}.bias") for n in "qkv"]
        ws = [w.reshape(w.shape[0], w.shape[1]) for w in ws]  # legacy checkpoints store 1x1 convs
        self.qkv_w = W.mat(torch.cat(ws, 0))
        self.qkv_b = W.host_vec(torch.cat(bs, 0))
        self.ow, self.ob = W.linear(pfx + "to_out.0.weight"), W.vec(pfx + "to_out.0.bias")

    def _call_parity(self, x):
        """fp32 in / out.  Both factors of q k^T and of p v are activations, so each product takes three bf16 terms,
        hi hi + lo hi + hi lo: the left factor as planes [hi | lo | hi], the right one as [hi | hi | lo] (ops.split patterns 1 / 2)."""
        B, H, Wd, C = x.shape
        L = H * Wd
        Lp = (L + 31) // 32 * 32
        n = ops.groupnorm(x, self.nw, self.nb, self.g, 1e-6, silu=False)
        qkv = ops.gemm(n.view(B * L, 2 * C), self.qkv_w, bias=self.qkv_b, out_f32=True)  # [B*L, 3C] fp32
        o = torch.empty((B * L, C), dtype=F32, device=x.device)
        scale = float(C) ** -0.5
        qb = max(32, min(L, (self.QBLOCK_BYTES // (4 * Lp)) // 32 * 32))
        self.last_block_bytes = 4 * Lp * qb
        s = torch.empty((qb, Lp), dtype=F32, device=x.device)
        for b in range(B):
            rows = slice(b * L, (b + 1) * L)
            k3 = ops.split(qkv[rows, C:2 * C], pattern=2)                                # [L, 3C]  = [k_hi | k_hi | k_lo]
            vt3 = ops.split(qkv[rows, 2 * C:], cpad=Lp, pattern=2, transposed=True)      # [C, 3Lp] = [v_hi^T | v_hi^T | v_lo^T]
            for q0 in range(0, L, qb):
                q1 = min(L, q0 + qb)
                q3 = ops.split(qkv[b * L + q0: b * L + q1, :C], pattern=1)               # [q, 3C]  = [q_hi | q_lo | q_hi]
                ops.gemm(q3, k3, out=s[: q1 - q0, :L], out_f32=True)
                p3 = ops.softmax_rows_split(s[: q1 - q0], scale, n=L)                    # [q, 3Lp] = [p_hi | p_lo | p_hi]
                ops.gemm(p3, vt3, out=o[b * L + q0: b * L + q1], out_f32=True)
        return ops.gemm(ops.split(o), self.ow, bias=self.ob, residual=x.view(B * L, C), out_f32=True).view(B, H, Wd, C)

    def _call_h16(self, x):
        """fp32 in / out, one fp16 plane per operand: q k^T and p v are one MFMA per product on fp16 q / k / p / v^T, logits and
        soft-max in fp32 (the fast form below with fp16 in place of bf16 and fp32 tensors around it)."""
        B, H, Wd, C = x.shape
        L = H * Wd
        Lp = (L + 31) // 32 * 32
        n = ops.groupnorm(x, self.nw, self.nb, self.g, 1e-6, silu=False)
        qkv = ops.gemm(n.view(B * L, C), self.qkv_w, bias=self.qkv_b, out_f32=True)  # [B*L, 3C] fp32
        o = torch.empty((B * L, C), dtype=F32, device=x.device)
        scale = float(C) ** -0.5
        qb = max(32, min(L, (self.QBLOCK_BYTES // (4 * Lp)) // 32 * 32))
        self.last_block_bytes = 4 * Lp * qb
        s = torch.empty((qb, Lp), dtype=F32, device=x.device)
        for b in range(B):
            rows = slice(b * L, (b + 1) * L)
            k = ops.split(qkv[rows, C:2 * C], h16=True)                                  # [L, C]
            vt = ops.split(qkv[rows, 2 * C:], cpad=Lp, transposed=True, h16=True)        # [C, Lp], columns behind L zero
            for q0 in range(0, L, qb):
                q1 = min(L, q0 + qb)
                q = ops.split(qkv[b * L + q0: b * L + q1, :C], h16=True)                 # [q, C]
                ops.gemm(q, k, out=s[: q1 - q0, :L], out_f32=True)
                p = ops.softmax_rows_split(s[: q1 - q0], scale, n=L, h16=True)           # [q, Lp]
                ops.gemm(p, vt, out=o[b * L + q0: b * L + q1], out_f32=True)
        return ops.gemm(ops.split(o, h16=True), self.ow, bias=self.ob, residual=x.view(B * L, C), out_f32=True).view(B, H, Wd, C)

    def __call__(self, x):
        if self.h16:
            return self._call_h16(x)
        if self.parity:
            return self._call_parity(x)
        B, H, Wd, C = x.shape
        L = H * Wd
        Lp = (L + 31) // 32 * 32  # key axis padded to the GEMM's K granularity (images whose latent area is not a multiple of 32)
        n = ops.groupnorm(x, self.nw, self.nb, self.g, 1e-6, silu=False)
        qkv = ops.gemm(n.view(B * L, C), self.qkv_w, bias=self.qkv_b)  # [B*L, 3C]
        o = torch.empty((B * L, C), dtype=BF16, device=x.device)
        scale = float(C) ** -0.5
        # S = q k^T stays in fp32 from the MFMA accumulator to the softmax (SDPA's logits are fp32; a bf16 round trip would
        # cost 2^-9 |logit| in the exponent at d = 512).  Query rows go in blocks so that the fp32 logits of a block stay
        # below QBLOCK_BYTES whatever the image size: 1024^2 images have L = 16 384, i.e. 1 GB per image unblocked.
        qb = max(32, min(L, (self.QBLOCK_BYTES // (4 * Lp)) // 32 * 32))
        self.last_block_bytes = 4 * Lp * qb  # what the largest logits block of this call occupies (tests assert the bound)
        if Lp != L:
            # padded keys: zero rows behind L in K (their logits are never read: the softmax stops at column L) and zero
            # columns behind L in P and V^T (they add 0 to P V); the buffers are made once per call
            kp = torch.zeros((Lp, C), dtype=BF16, device=x.device)
            vtp = torch.zeros((C, Lp), dtype=BF16, device=x.device)
            pp = torch.zeros((qb, Lp), dtype=BF16, device=x.device)
        for b in range(B):
            rows = slice(b * L, (b + 1) * L)
            k, v = qkv[rows, C:2 * C], qkv[rows, 2 * C:]
            vt = _transpose(v, C)  # [C, L]
            if Lp != L:
                kp[:L].copy_(k)
                vtp[:, :L].copy_(vt)
                k, vt = kp, vtp
            for q0 in range(0, L, qb):
                q1 = min(L, q0 + qb)
                s = ops.gemm(qkv[b * L + q0: b * L + q1, :C], k, out_f32=True)  # [q1 - q0, Lp] fp32
                if Lp != L:
                    p = ops.softmax_rows(s, scale, n=L, out=pp[: q1 - q0])
                else:
                    p = ops.softmax_rows(s, scale)
                ops.gemm(p, vt, out=o[b * L + q0: b * L + q1])
        return ops.gemm(o, self.ow, bias=self.ob, residual=x.view(B * L, C)).view(B, H, Wd, C)


def _transpose(v: torch.Tensor, C: int) -> torch.Tensor:
    """v [L, C] row-strided view -> contiguous [C, L] via the NHWC->NCHW kernel."""
    from . import lib as _l
    lib = _l.load()
    L = v.shape[0]
    y = torch.empty((C, L), dtype=BF16, device=v.device)
    _l.check(lib.dm4d_nhwc_to_nchw_bf16(torch.cuda.current_stream().cuda_stream, v.data_ptr(), y.data_ptr(), 1, C, L,
                                        v.stride(0)), "dm4d_nhwc_to_nchw_bf16")
    return y


class AutoencoderKL:
    def __init__(self, config: VAEConfig, state_dict: Dict[str, torch.Tensor], device="cuda", precision: str = "fast",
                 weight_dtype=BF16):
        """precision "parity": fp32 tensors between kernels, two-term bf16 operands; "fp16": fp32 tensors, single-term fp16 operands on
        the weights as `weight_dtype` holds them (see host/unet.py, include/dm4d.h)."""
        cfg = self.config = config
        self.device = torch.device(device)
        check_precision(precision)
        self.precision, self.parity, self.h16 = precision, precision == "parity", precision == "fp16"
        self.wide = self.parity or self.h16
        W = _Weights(state_dict, self.device, self.parity, self.h16, weight_dtype)
        self._W = W  # host_vec / mat for the padded vectors below
        g, boc, lc = cfg.norm_num_groups, cfg.block_out_channels, cfg.latent_channels
        if cfg.in_channels > PAD or 2 * lc > PAD:
            raise NotImplementedError("channel counts above 32 at the VAE boundary")
        # ---- encoder ----
        self.e_in_w, self.e_in_b = W.conv3("encoder.conv_in.weight", cin_pad=PAD), W.vec("encoder.conv_in.bias")
        self.e_down = []
        for i in range(len(boc)):
            p = f"encoder.down_blocks.{i}."
            res = [_Res(W, p + f"resnets.{j}.", g) for j in range(cfg.layers_per_block)]
            ds = None
            if i != len(boc) - 1:
                ds = (W.conv3(p + "downsamplers.0.conv.weight"), W.vec(p + "downsamplers.0.conv.bias"))
            self.e_down.append((res, ds))
        self.e_mid = (_Res(W, "encoder.mid_block.resnets.0.", g), _MidAttn(W, "encoder.mid_block.attentions.0.", g),
                      _Res(W, "encoder.mid_block.resnets.1.", g))
        self.e_nw, self.e_nb = W.vec("encoder.conv_norm_out.weight"), W.vec("encoder.conv_norm_out.bias")
        # conv_out writes a 32-wide row (channels >= 2*lc are zero) so quant_conv is one K=32 GEMM
        self.e_out_w = W.conv3("encoder.conv_out.weight", cout_pad=PAD)
        self.e_out_b = W.host_vec(_pad_vec(W.get("encoder.conv_out.bias"), PAD))
        self.quant_w = W.mat(_pad_mat(W.get("quant_conv.weight"), 2 * lc, PAD, "cpu"))
        self.quant_b = W.vec("quant_conv.bias")
        # ---- decoder ----
        self.pq_w = W.mat(_pad_mat(W.get("post_quant_conv.weight"), PAD, PAD, "cpu"))  # out rows >= lc are zero
        self.pq_b = W.host_vec(_pad_vec(W.get("post_quant_conv.bias"), PAD))
        self.d_in_w, self.d_in_b = W.conv3("decoder.conv_in.weight", cin_pad=PAD), W.vec("decoder.conv_in.bias")
        self.d_mid = (_Res(W, "decoder.mid_block.resnets.0.", g), _MidAttn(W, "decoder.mid_block.attentions.0.", g),
                      _Res(W, "decoder.mid_block.resnets.1.", g))
        self.d_up = []
        for i in range(len(boc)):
            p = f"decoder.up_blocks.{i}."
            res = [_Res(W, p + f"resnets.{j}.", g) for j in range(cfg.layers_per_block + 1)]
            us = None
            if i != len(boc) - 1:
                us = ops.Upsampler(W.conv3(p + "upsamplers.0.conv.weight"), W.vec(p + "upsamplers.0.conv.bias"), parity=self.parity,
                                   h16=self.h16)
            self.d_up.append((res, us))
        self.d_nw, self.d_nb = W.vec("decoder.conv_norm_out.weight"), W.vec("decoder.conv_norm_out.bias")
        self.d_out_w, self.d_out_b = W.conv3("decoder.conv_out.weight"), W.vec("decoder.conv_out.bias")
        unused = [k for k in state_dict if k not in W.used]
        if unused:
            raise KeyError(f"unexpected keys in VAE checkpoint (strict load): {unused[:8]}")

    @classmethod
    def from_pretrained(cls, path, device="cuda", variant: Optional[str] = None, precision: str = "fast", weight_dtype=BF16) -> "AutoencoderKL":
        from .weights import load_component_state_dict
        path = Path(path)
        cfg = VAEConfig.from_dict(json.loads((path / "config.json").read_text()))
        return cls(cfg, load_component_state_dict(path, variant), device, precision, weight_dtype)

    @property
    def scale_factor(self) -> int:
        return 2 ** (len(self.config.block_out_channels) - 1)

    def mid_attention_block_bytes(self) -> int:
        """Bytes of the largest fp32 logits block the two mid-block attentions allocated in their last calls."""
        return max(getattr(self.e_mid[1], "last_block_bytes", 0), getattr(self.d_mid[1], "last_block_bytes", 0))

    def micro_batch(self, height: int, width: int, limit: int = 8) -> int:
        """Images per VAE pass: the reference's 8 (pipeline_diffuman4d.py:47,59), reduced for large images so that the
        widest full-resolution activation ([B, H, W, 2*C0] in the decoder) stays below the kernels' 2^31-element limit."""
        c = (4 if self.parity else 2) * self.config.block_out_channels[0]  # parity: the widest tensor is a two-term operand
        return max(1, min(limit, ((1 << 31) - 1) // (height * width * c)))

    # ---- encode --------------------------------------------------------------------------------
    def moments(self, x_nhwc32: torch.Tensor) -> torch.Tensor:
        """x [B,H,W,32] (3 image channels + zero pad) -> moments [B,h,w,2*lc] (mean | logvar).
        precision "parity": x is the two-term operand [B,H,W,64] of the fp32 image ("fp16": its fp16 operand [B,H,W,32]), the moments fp32."""
        P = self.wide
        op = (lambda t: ops.split(t, h16=self.h16)) if P else (lambda t: t)  # fp32 tensor -> operand of the next contraction
        x = ops.conv3x3(x_nhwc32, self.e_in_w, bias=self.e_in_b, out_f32=P)
        for res, ds in self.e_down:
            for r in res:
                x = r(x)
            if ds is not None:
                x = ops.conv3x3(op(x), ds[0], bias=ds[1], stride=2, pad=0, pad_hi=1, out_f32=P)  # F.pad(0,1,0,1) + conv s2 p0
        x = self.e_mid[2](self.e_mid[1](self.e_mid[0](x)))
        x = ops.groupnorm(x, self.e_nw, self.e_nb, self.config.norm_num_groups, 1e-6, silu=True)
        x = ops.conv3x3(x, self.e_out_w, bias=self.e_out_b, out_f32=P)  # [B,h,w,32]
        B, h, w, _ = x.shape
        return ops.gemm(op(x.view(-1, PAD)), self.quant_w, bias=self.quant_b, out_f32=P).view(B, h, w, -1)

    def encode_scaled(self, images: torch.Tensor, noise: Optional[torch.Tensor], batch_size: int = 8,
                      cache: Optional[dict] = None, keys=None) -> torch.Tensor:
        """pipeline_diffuman4d.py:47-56: images NCHW in [-1,1] (CPU or GPU) -> latents NHWC * scaling_factor.
        `noise` NCHW [N, lc, h, w] (any device) is the posterior draw; drawn on the device if None.
        `cache` / `keys` (one hashable key per image): the encoder's posterior MOMENTS are deterministic, so images
        seen before (the reference re-encodes all 2N images of every task in every alternation round) skip the
        encoder; the stochastic draw is still fresh for every call, i.e. the sampled distribution is unchanged."""
        lc = self.config.latent_channels
        n = images.shape[0]
        batch_size = self.micro_batch(images.shape[-2], images.shape[-1], batch_size)
        if cache is None:
            todo = list(range(n))
        else:
            if keys is None or len(keys) != n:
                raise ValueError("encode_scaled: one cache key per image is required")
            todo = [i for i in range(n) if keys[i] not in cache]
        fresh = {}
        for j in range(0, len(todo), batch_size):
            idx = todo[j:j + batch_size]
            sel = images[idx[0]:idx[-1] + 1] if idx == list(range(idx[0], idx[-1] + 1)) else images[idx]
            m = self.moments(self._image_operand(sel))
            for k, i in enumerate(idx):
                fresh[i] = m[k]
        if cache is not None:
            on_gpu = self.device.type == "cuda"
            cur = torch.cuda.current_stream(self.device) if on_gpu else None
            if fresh:
                # The cache is shared by the runner's task streams (gpu_streams > 1): an entry may be read by another stream as
                # soon as it is in the dict.  It is published together with an EVENT recorded behind the kernels that write it;
                # a reader on another stream makes its stream wait for that event.  (Round 2 drained the writing stream on the
                # host instead: one host-side stall per task of the first alternation round, with the GPU idling behind it.)
                ev = None
                clones = {i: m.clone() for i, m in fresh.items()}
                if on_gpu:
                    ev = torch.cuda.Event()
                    ev.record(cur)
                for i, c in clones.items():
                    cache[keys[i]] = (c, ev, cur)
            rows, waited = [], set()
            for i in range(n):
                c, ev, st = cache[keys[i]]
                if ev is not None and st != cur and id(ev) not in waited:  # written on another stream: order this stream behind it
                    cur.wait_event(ev)
                    waited.add(id(ev))
                rows.append(c)
        else:
            rows = [fresh[i] for i in range(n)]
        outs = []
        for i in range(0, n, batch_size):
            m = torch.stack(rows[i:i + batch_size])
            B, h, w, _ = m.shape
            if noise is not None and self.wide:
                nb = noise[i:i + batch_size].float().permute(0, 2, 3, 1).contiguous().to(self.device)  # layout change on the host
            elif noise is not None:
                nb = ops.nchw_to_nhwc(noise[i:i + batch_size].to(self.device, BF16).contiguous())
            else:
                nb = torch.randn((B, h, w, lc), device=self.device, dtype=torch.float32)
                nb = nb if self.wide else nb.to(BF16)
            outs.append(ops.vae_sample(m, nb, lc, self.config.scaling_factor))
        return torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]

    def _image_operand(self, images: torch.Tensor) -> torch.Tensor:
        """NCHW images (any device / float dtype) -> what conv_in reads: NHWC bf16 padded to 32 channels, or (parity) the two-term
        operand [B,H,W,64] of the fp32 image (the NCHW -> NHWC permutation of a host tensor is made on the host)."""
        if not self.wide:
            return ops.nchw_to_nhwc(images.to(self.device, BF16).contiguous(), PAD)
        x = images.float().permute(0, 2, 3, 1).contiguous().to(self.device)
        return ops.split(x, cpad=PAD, h16=self.h16)

    # ---- decode --------------------------------------------------------------------------------
    def decode(self, z_nhwc: torch.Tensor) -> torch.Tensor:
        """z [B,h,w,lc] (already divided by scaling_factor, padded to 32) -> image NHWC [B,H,W,3].
        precision "parity": z is the two-term operand [B,h,w,64] ("fp16": the fp16 operand [B,h,w,32]), the image fp32."""
        P = self.wide
        B, h, w, _ = z_nhwc.shape
        x = ops.gemm(z_nhwc.view(B * h * w, -1), self.pq_w, bias=self.pq_b, out_f32=P).view(B, h, w, PAD)
        x = ops.conv3x3(ops.split(x, h16=self.h16) if P else x, self.d_in_w, bias=self.d_in_b, out_f32=P)
        x = self.d_mid[2](self.d_mid[1](self.d_mid[0](x)))
        for res, us in self.d_up:
            for r in res:
                x = r(x)
            if us is not None:
                x = us(x)
        x = ops.groupnorm(x, self.d_nw, self.d_nb, self.config.norm_num_groups, 1e-6, silu=True)
        return ops.conv3x3(x, self.d_out_w, bias=self.d_out_b, out_f32=P)

    def decode_to_images(self, lat_nhwc: torch.Tensor, batch_size: int = 8, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """pipeline_diffuman4d.py:59-72,280-285: latents NHWC -> images NCHW in [0,1].
        `rows` (bool [N]): decode only these latents; the other output images are zero."""
        f = self.scale_factor
        batch_size = self.micro_batch(lat_nhwc.shape[1] * f, lat_nhwc.shape[2] * f, batch_size)
        if rows is None:
            sel = lat_nhwc
        else:
            idx = torch.nonzero(rows.to(lat_nhwc.device)).flatten()
            sel = lat_nhwc.index_select(0, idx)
        outs = []
        for i in range(0, sel.shape[0], batch_size):
            if self.wide:
                z = ops.split(sel[i:i + batch_size].contiguous(), cpad=PAD, scale=1.0 / self.config.scaling_factor, h16=self.h16)
            else:
                z = ops.scale_pad(sel[i:i + batch_size].contiguous(), PAD, 1.0 / self.config.scaling_factor)
            outs.append(ops.postprocess_images(self.decode(z), self.config.out_channels))
        if rows is None:
            return torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]
        if not outs:  # nothing to decode (an early alternation round): a zero-stride host tensor, no device traffic
            return torch.zeros(1).expand(lat_nhwc.shape[0], self.config.out_channels, lat_nhwc.shape[1] * f,
                                         lat_nhwc.shape[2] * f)
        full = torch.zeros((lat_nhwc.shape[0], self.config.out_channels, lat_nhwc.shape[1] * f, lat_nhwc.shape[2] * f),
                           dtype=outs[0].dtype, device=lat_nhwc.device)
        full.index_copy_(0, idx, torch.cat(outs, dim=0) if len(outs) > 1 else outs[0])
        return full

    # ---- conditioning resize (encode_image_resizing, :90-100) -----------------------------------
    def resize_to_nhwc(self, images: torch.Tensor, size, mode: str, batch_size: int = 16) -> torch.Tensor:
        outs = []
        for i in range(0, images.shape[0], batch_size):
            xb = images[i:i + batch_size].to(self.device, torch.float32).contiguous()
            outs.append(ops.resize_to_nhwc(xb, size, mode, out_f32=self.wide))
        return torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]


def _pad_vec(v: torch.Tensor, n: int) -> torch.Tensor:
    """Host vector zero-padded to n entries (fp32; _Weights.host_vec makes the device vector of the precision)."""
    out = torch.zeros(n)
    out[: v.shape[0]] = v.float()
    return out


def _pad_mat(w: torch.Tensor, rows: int, cols: int, device) -> torch.Tensor:
    w = w.float().reshape(w.shape[0], w.shape[1])
    out = torch.zeros(rows, cols)
    out[: w.shape[0], : w.shape[1]] = w
    return out.contiguous()  # fp32 on the host: _Weights.mat rounds to the precision's weight dtype
