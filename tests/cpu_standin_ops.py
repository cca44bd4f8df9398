#!/usr/bin/env python
"""TEST / DEV INFRASTRUCTURE (never imported by the product; tests/test_host_wiring.py installs it for its own process and removes it
again): a torch-CPU stand-in for `diffuman4d_amd.host.ops`,
so that the HOST side of a precision mode -- weight layouts, operand planes, which tensor feeds which launch -- can be exercised
without a GPU before a `gpurun` call is spent on it.  It mimics the documented semantics of every wrapper (rounding points
included: bf16 outputs are rounded once, fp32 accumulation is emulated in fp32 / fp64), not the kernels; the kernels are
checked on the GPU by tests/opcheck.py.

    python tests/cpu_standin_ops.py            # tiny UNet / VAE / pipeline in the fast, parity and fp16 precisions, vs the fp32 oracle
"""
from __future__ import annotations

import math
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
BF, F16, F32 = torch.bfloat16, torch.float16, torch.float32


def _out(v, f32=False, split=False, h16=False):
    """fp32 values -> what a kernel stores: fp32, a two-term bf16 operand [hi | lo], or ONE plane of bf16 / (h16) fp16."""
    if f32:
        return v.float()
    if h16:
        return v.to(F16)
    if split:
        hi = v.to(BF)
        return torch.cat([hi, (v - hi.float()).to(BF)], dim=-1)
    return v.to(BF)


def gemm(a, w, *, a2=None, bias=None, rowbias=None, rows_per_rowbias=1, residual=None, geglu=False, silu=False, out_scale=1.0,
         out=None, out_f32=False, split_out=False, scale_cols=0, col_scale=1.0):
    h16 = w.dtype == F16  # precision "fp16": fp16 operands and bias, one fp16 plane (or fp32) out
    assert a.dtype == w.dtype and w.dtype in (BF, F16) and (bias is None or bias.dtype == w.dtype)
    assert scale_cols == 0 or h16
    x = a.double() if a2 is None else torch.cat([a, a2], dim=1).double()
    assert x.shape[1] == w.shape[1], (x.shape, w.shape)
    v = x @ w.double().t()
    if bias is not None:
        v = v + bias.double()
    if geglu:
        h, g = v.chunk(2, dim=-1)
        v = h * F.gelu(g)
    if silu:
        v = F.silu(v)
    side = [t for t in (rowbias, residual) if t is not None]
    if side:
        assert all(t.dtype == side[0].dtype for t in side) and side[0].dtype in (w.dtype, F32)
        assert (side[0].dtype == F32) == (out_f32 or split_out) or side[0].dtype == w.dtype
    if rowbias is not None:
        v = v + rowbias.double().repeat_interleave(rows_per_rowbias, dim=0)[: v.shape[0]]
    if residual is not None:
        v = v + residual.double()
    v = v * out_scale
    if scale_cols:
        v = torch.cat([v[:, :scale_cols] * col_scale, v[:, scale_cols:]], dim=1)
    r = _out(v.float(), out_f32, split_out, h16)
    if out is not None:
        out[:, : r.shape[1]].copy_(r)
        return out
    return r


def conv_out_hw(h, w, stride, pad, upsample, pad_hi=None):
    if upsample:
        return 2 * h, 2 * w
    ph = pad if pad_hi is None else pad_hi
    return (h + pad + ph - 3) // stride + 1, (w + pad + ph - 3) // stride + 1


def conv3x3(x, wt, *, bias=None, rowbias=None, residual=None, stride=1, pad=1, pad_hi=None, upsample=False, out_scale=1.0, out_f32=False):
    h16 = wt.dtype == F16
    assert x.dtype == wt.dtype and wt.dtype in (BF, F16) and (bias is None or bias.dtype == wt.dtype)
    B, H, W, Cin = x.shape
    Cout = wt.shape[0]
    assert wt.shape[1] == 9 * Cin, (wt.shape, Cin)
    w4 = wt.double().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    xi = x.double().permute(0, 3, 1, 2)
    if upsample:
        xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    ph = pad if pad_hi is None else pad_hi
    v = F.conv2d(F.pad(xi, (pad, ph, pad, ph)), w4, bias.double() if bias is not None else None, stride=stride)
    sides = [t for t in (rowbias, residual) if t is not None]
    side = F32 if (out_f32 or (h16 and sides and sides[0].dtype == F32)) else wt.dtype
    if rowbias is not None:
        assert rowbias.dtype == side
        v = v + rowbias.double()[:, :, None, None]
    v = v.permute(0, 2, 3, 1)
    if residual is not None:
        assert residual.dtype == side
        v = v + residual.double().reshape(v.shape)
    return _out((v * out_scale).float(), out_f32, h16=h16).contiguous()


def conv2d_direct(x, wt, *, ksize, bias=None, stride=1, pad=1, silu=False):
    B, H, W, Cin = x.shape
    w = wt.double().view(-1, ksize, ksize, Cin).permute(0, 3, 1, 2)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w, None if bias is None else bias.double(), stride=stride, padding=pad)
    y = F.silu(y) if silu else y
    return y.permute(0, 2, 3, 1).contiguous().to(x.dtype)


def dup_k(w, taps=1, times=2):
    n, k = w.shape
    c = k // taps
    return w.reshape(n, taps, 1, c).expand(n, taps, times, c).reshape(n, taps * times * c).contiguous()


def split(x, x2=None, *, cpad=None, silu=False, scale=1.0, pattern=0, transposed=False, h16=False):
    assert x.dtype == F32
    if transposed:
        x = x.t()
    lead = tuple(x.shape[:-1])
    v = x.reshape(-1, x.shape[-1])
    if x2 is not None:
        v = torch.cat([v, x2.reshape(-1, x2.shape[-1])], dim=1)
    if silu:
        v = F.silu(v)
    v = v * scale
    Cp = cpad or v.shape[1]
    v = F.pad(v, (0, Cp - v.shape[1]))
    if h16:
        return v.to(F16).view(lead + (Cp,))
    hi = v.to(BF)
    lo = (v - hi.float()).to(BF)
    planes = {0: [hi, lo], 1: [hi, lo, hi], 2: [hi, hi, lo]}[pattern]
    return torch.cat(planes, dim=1).view(lead + (len(planes) * Cp,))


def groupnorm(x1, gamma, beta, groups, eps, *, x2=None, silu=False, raw_out=False):
    f32 = x1.dtype == F32
    x = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
    if raw_out:
        assert f32 and gamma.dtype == F16
        return groupnorm(x1, gamma, beta, groups, eps, x2=x2, silu=silu), x.to(F16)
    B, C = x.shape[0], x.shape[-1]
    v = F.group_norm(x.double().reshape(B, -1, C).permute(0, 2, 1), groups, gamma.double(), beta.double(), eps).permute(0, 2, 1)
    if silu:
        v = F.silu(v)
    v = v.reshape(x.shape).float().contiguous()
    assert gamma.dtype == beta.dtype and (gamma.dtype == BF or f32 or x1.dtype == F16)
    if x1.dtype == F16:  # fp16 precision, fp16 in (conv1 -> norm2)
        assert gamma.dtype == F16
        return v.to(F16)
    return _out(v, split=True, h16=gamma.dtype == F16) if f32 else v.to(BF)


def layernorm(x, gamma, beta, eps=1e-5):
    v = F.layer_norm(x.double(), (x.shape[-1],), gamma.double(), beta.double(), eps).float()
    return _out(v, split=True, h16=gamma.dtype == F16) if x.dtype == F32 else v.to(BF)


LOG2E = 1.4426950408889634


def attention(q, k, v, batch, heads, seq, scale=None, out=None, kv_seq=None, q_scaled=False):
    kv_seq = kv_seq or seq
    hv = lambda t, L: t.double().reshape(batch, L, heads, 64).transpose(1, 2)  # noqa: E731
    s = hv(q, seq) @ hv(k, kv_seq).transpose(-1, -2)
    p = torch.softmax(s * (math.log(2.0) if q_scaled else (0.125 if scale is None else scale)), dim=-1)
    assert q.dtype == k.dtype == v.dtype and (q.dtype == BF or q_scaled)
    o = p @ hv(v, kv_seq)
    return o.transpose(1, 2).reshape(batch * seq, heads * 64).to(q.dtype)


def attention_split(qkv, batch, heads, seq, scale=None, *, q=None, kv=None, kv_seq=None):
    C = heads * 64
    if qkv is not None:
        assert qkv.shape == (batch * seq, 6 * C) and qkv.dtype == BF
        val = qkv[:, :3 * C].double() + qkv[:, 3 * C:].double()
        qv, kval, vval, kv_seq = val[:, :C], val[:, C:2 * C], val[:, 2 * C:], seq
    else:  # frame-sharded form: q [.., 2C] = [q_hi | q_lo], kv [.., 4C] = [k_hi | v_hi | k_lo | v_lo]
        kv_seq = kv_seq or seq
        assert q.shape == (batch * seq, 2 * C) and kv.shape == (batch * kv_seq, 4 * C) and q.dtype == kv.dtype == BF
        qv = q[:, :C].double() + q[:, C:].double()
        kval, vval = kv[:, :C].double() + kv[:, 2 * C:3 * C].double(), kv[:, C:2 * C].double() + kv[:, 3 * C:].double()
    hv = lambda t, L: t.reshape(batch, L, heads, 64).transpose(1, 2)  # noqa: E731
    o = torch.softmax(hv(qv, seq) @ hv(kval, kv_seq).transpose(-1, -2) * (0.125 if scale is None else scale), dim=-1) @ hv(vval, kv_seq)
    return _out(o.transpose(1, 2).reshape(batch * seq, C).float(), split=True)


def softmax_rows(s, scale, n=None, out=None):
    N = s.shape[1] if n is None else n
    p = torch.softmax(s[:, :N].double() * scale, dim=-1).to(BF)
    if out is None:
        out = torch.zeros(s.shape, dtype=BF)
    out[:, :N] = p
    return out


def softmax_rows_split(s, scale, n=None, h16=False):
    M, Np = s.shape
    N = Np if n is None else n
    p = F.pad(torch.softmax(s[:, :N].double() * scale, dim=-1).float(), (0, Np - N))
    if h16:
        return p.to(F16)
    hi = p.to(BF)
    return torch.cat([hi, (p - hi.float()).to(BF), hi], dim=1)


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0, out_f32=False):
    half = dim // 2
    e = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=F32) / (half - freq_shift))
    arg = t.float()[:, None] * e[None]
    emb = torch.cat([torch.cos(arg), torch.sin(arg)] if flip_sin_to_cos else [torch.sin(arg), torch.cos(arg)], dim=1)
    return emb if out_f32 else emb.to(BF)


def silu(x):
    return F.silu(x.float()).to(BF)


def pack_model_input(latents, pv_lat, plucker, skel, mask, is_cond, cpad, use_cfg, frame_idx=None, h16=False):
    f32 = latents.dtype == F32
    rows = frame_idx.long() if frame_idx is not None else torch.arange(latents.shape[0])
    c = is_cond.bool()
    latents[rows[c]] = pv_lat[rows[c]]
    x = latents[rows].float()
    parts = [x, plucker[rows].float()] + ([skel[rows].float()] if skel is not None else []) + [mask[rows].float()]
    pos = torch.cat(parts, dim=-1)
    outs = [pos]
    if use_cfg:
        neg = [torch.where(c[:, None, None], torch.ones_like(x), x), torch.zeros_like(plucker[rows].float())]
        neg += ([-torch.ones_like(skel[rows].float())] if skel is not None else []) + [mask[rows].float()]
        outs = [torch.cat(neg, dim=-1), pos]
    v = F.pad(torch.cat(outs), (0, cpad - pos.shape[-1]))
    return _out(v, split=True, h16=h16) if f32 else v.to(BF)


def cfg_ddim_step(latents, noise_pred, coef, is_cond, use_cfg, guidance_scale, v_prediction, frame_idx=None):
    rows = frame_idx.long() if frame_idx is not None else torch.arange(latents.shape[0])
    Fn = is_cond.shape[0]
    npd = noise_pred.double()[..., :4]
    e = npd[:Fn] + guidance_scale * (npd[Fn:] - npd[:Fn]) if use_cfg else npd
    x = latents[rows].double()
    sa, sb, sap, sbp = (coef[:, i].double()[:, None, None] for i in range(4))
    x0, ee = ((sa * x - sb * e, sa * e + sb * x) if v_prediction else ((x - sb * e) / sa, e))
    new = (sap * x0 + sbp * ee).to(latents.dtype)
    keep = ~is_cond.bool()
    latents[rows[keep]] = new[keep]
    return latents


def cfg_linear_step(latents, x0_prev, noise_pred, coef, is_cond, use_cfg, guidance_scale, frame_idx=None):
    rows = frame_idx.long() if frame_idx is not None else torch.arange(latents.shape[0])
    Fn = is_cond.shape[0]
    npd = noise_pred.double()[..., :4]
    m = npd[:Fn] + guidance_scale * (npd[Fn:] - npd[:Fn]) if use_cfg else npd
    a, b, c, d, e = (coef[:, i].double()[:, None, None] for i in range(5))
    x, p = latents[rows].double(), x0_prev[rows].double()
    keep = ~is_cond.bool()
    latents[rows[keep]] = (a * x + b * m + c * p).to(latents.dtype)[keep]
    x0_prev[rows[keep]] = (d * x + e * m).to(latents.dtype)[keep]
    return latents


def cfg_multistep_step(latents, states, noise_pred, coef, is_cond, use_cfg, guidance_scale, frame_idx=None):
    rows = frame_idx.long() if frame_idx is not None else torch.arange(latents.shape[0])
    Fn = is_cond.shape[0]
    npd = noise_pred.double()[..., :4]
    m = npd[:Fn] + guidance_scale * (npd[Fn:] - npd[:Fn]) if use_cfg else npd
    k = [coef[:, i].double()[:, None, None] for i in range(13)]
    z = torch.zeros_like(latents[rows].double())
    x = latents[rows].double()
    s1, s2, s3 = (states[j][rows].double() if j < len(states) else z for j in range(3))
    conv = k[0] * x + k[1] * m
    xc = k[2] * x + k[3] * s3 + k[4] * s1 + k[5] * s2 + k[6] * conv
    xn = k[7] * xc + k[8] * conv + k[9] * s1 + k[10] * s2 + k[11] * s3
    keep = ~is_cond.bool()
    dt = latents.dtype
    latents[rows[keep]] = xn.to(dt)[keep]
    mode = coef[:, 12].long()  # 0: (conv, s1, xc);  1: stored tensors kept;  2: shifted (conv, s1, s2)
    upd = keep & (mode != 1)
    if len(states) >= 3:
        states[2][rows[upd]] = torch.where((mode == 2)[:, None, None], s2, xc).to(dt)[upd]
    if len(states) >= 2:
        states[1][rows[upd]] = s1.to(dt)[upd]
    states[0][rows[upd]] = conv.to(dt)[upd]
    return latents


def nchw_to_nhwc(x, cpad=None):
    y = x.permute(0, 2, 3, 1)
    return F.pad(y, (0, (cpad or x.shape[1]) - x.shape[1])).contiguous()


def nhwc_to_nchw(x, C=None):
    return x[..., : (C or x.shape[-1])].permute(0, 3, 1, 2).contiguous()


def vae_sample(moments, noise, channels, scale):
    m = moments.double()
    v = (m[..., :channels] + torch.exp(0.5 * m[..., channels:2 * channels].clamp(-30, 20)) * noise.double()) * scale
    return v.to(moments.dtype)


def scale_pad(x, cpad, scale):
    return F.pad((x.float() * scale), (0, cpad - x.shape[-1])).to(BF)


def resize_to_nhwc(x, size, mode, out_f32=False):
    y = F.interpolate(x.float(), size=size, mode=mode).permute(0, 2, 3, 1).contiguous()
    return y if out_f32 else y.to(BF)


def postprocess_images(x, channels=3):
    y = (x[..., :channels].float() / 2 + 0.5).clamp(0, 1).permute(0, 3, 1, 2).contiguous()
    return y if x.dtype == F32 else y.to(BF)


class FeedForward:
    def __init__(self, w1, b1, w2, b2):
        self.w1, self.b1, self.w2, self.b2 = w1, b1, w2, b2

    def after_attention(self, a, wo, bo, x, ln):
        h = gemm(a, wo, bias=bo, residual=x)
        f = gemm(layernorm(h, *ln), self.w1, bias=self.b1, geglu=True)
        return gemm(f, self.w2, bias=self.b2, residual=h)

    def after_attention_f16(self, a, wo, bo, x, ln, out_f32):
        h = gemm(a, wo, bias=bo, residual=x, out_f32=True)
        f = gemm(layernorm(h, *ln), self.w1, bias=self.b1, geglu=True)
        return gemm(f, self.w2, bias=self.b2, residual=h, out_f32=out_f32, split_out=not out_f32)


class Upsampler:
    def __init__(self, wt, bias, parity=False, h16=False):
        self.wt, self.bias, self.parity, self.h16 = wt, bias, parity or h16, h16

    def __call__(self, x):
        if self.parity:
            return conv3x3(split(x, h16=self.h16), self.wt, bias=self.bias, upsample=True, out_f32=True)
        return conv3x3(x, self.wt, bias=self.bias, upsample=True)


_NAMES = ("gemm", "conv_out_hw", "conv3x3", "conv2d_direct", "dup_k", "split", "groupnorm", "layernorm", "attention", "attention_split", "softmax_rows",
          "softmax_rows_split", "timestep_embedding", "silu", "pack_model_input", "cfg_ddim_step", "cfg_linear_step", "cfg_multistep_step", "nchw_to_nhwc",
          "nhwc_to_nchw", "vae_sample", "scale_pad", "resize_to_nhwc", "postprocess_images", "FeedForward", "Upsampler")
_SAVED = {}


def install():
    """Replace the wrappers of diffuman4d_amd.host.ops with the functions of this module (this process only; `uninstall` undoes it)."""
    from diffuman4d_amd.host import ops, vae
    me = sys.modules[__name__]
    if not _SAVED:
        _SAVED.update({"ops": {n: getattr(ops, n) for n in _NAMES}, "transpose": vae._transpose, "set_device": torch.cuda.set_device})
    for name in _NAMES:
        setattr(ops, name, getattr(me, name))
    vae._transpose = lambda v, C: v.t().contiguous()
    torch.cuda.set_device = lambda *a, **k: None


def uninstall():
    from diffuman4d_amd.host import ops, vae
    if _SAVED:
        for n, f in _SAVED["ops"].items():
            setattr(ops, n, f)
        vae._transpose, torch.cuda.set_device = _SAVED["transpose"], _SAVED["set_device"]
        _SAVED.clear()


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    install()
    import modelcheck as mc
    from dataclasses import asdict
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    if sys.argv[1:2] == ["golden"]:  # python tests/cpu_standin_ops.py golden <modelcheck case> ...: a golden-pipeline case on the stand-in
        mc.hip_unet = lambda cfg, om, precision="fast": UNetMultiviewConditionModel(UNetConfig.from_dict(asdict(cfg)), om.state_dict(), "cpu", precision)
        mc.hip_vae = lambda cfg, om, precision="fast": AutoencoderKL(VAEConfig.from_dict(asdict(cfg)), om.state_dict(), "cpu", precision)
        import diffuman4d_amd.host.pipeline as hp_
        orig = hp_.Diffuman4DPipeline
        hp_.Diffuman4DPipeline = lambda v, u, s, dev: orig(v, u, s, "cpu")
        for name in sys.argv[2:]:
            fn, kw = mc.CASES[name]
            print(name, fn(**kw), flush=True)
        return
    from dataclasses import asdict
    from diffuman4d_amd.host import ops
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMConfig as HC, DDIMScheduler as HS
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
    from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
    from oracle.ddim import DDIMConfig, DDIMScheduler
    from oracle.pipeline import OraclePipeline
    for tem, domain in ((False, "spatial"), (True, "temporal")):
        cfg, om = mc.make_unet(0, enable_tem_embeds=tem)
        if tem:
            g = torch.Generator().manual_seed(5)
            with torch.no_grad():
                for p in om.temporal_pos_embed.parameters():
                    p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(BF).float())
        g = torch.Generator().manual_seed(1)
        x = torch.randn(8, cfg.in_channels, 16, 8, generator=g).to(BF)
        t = torch.randint(0, 1000, (8,), generator=g)
        with torch.no_grad():
            ref = om(x.float(), t, domains=[domain] * 2, num_frames=4)
        for prec in ("fast", "parity", "fp16"):
            hm = UNetMultiviewConditionModel(UNetConfig.from_dict(asdict(cfg)), om.state_dict(), "cpu", prec)
            xin = ops.split(x.float().permute(0, 2, 3, 1).contiguous(), cpad=hm.IN_PAD, h16=prec == "fp16") if prec != "fast" else ops.nchw_to_nhwc(x, hm.IN_PAD)
            out = ops.nhwc_to_nchw(hm(xin, t.float(), domains=[domain] * 2, num_frames=4))
            print(f"unet {domain} tem={tem} {prec}: rel_l2 vs fp32 oracle = {rel_l2(out, ref):.3e}", flush=True)
    for hw in ((64, 64), (264, 328)):
        cfgv, ov = mc.make_vae(1)
        g = torch.Generator().manual_seed(2)
        img = (torch.rand(2, 3, *hw, generator=g) * 2 - 1).to(BF)
        noise = torch.randn(2, 4, hw[0] // 8, hw[1] // 8, generator=g).to(BF)
        with torch.no_grad():
            z_ref = ov.sample_posterior(ov.moments(img.float()), noise.float()) * cfgv.scaling_factor
            im_ref = (ov.decode(z_ref.to(BF).float() / cfgv.scaling_factor) / 2 + 0.5).clamp(0, 1)
        for prec in ("fast", "parity", "fp16"):
            hv = AutoencoderKL(VAEConfig.from_dict(asdict(cfgv)), ov.state_dict(), "cpu", prec)
            z = ops.nhwc_to_nchw(hv.encode_scaled(img, noise))
            zin = z_ref.to(BF)
            lat = zin.float().permute(0, 2, 3, 1).contiguous() if prec != "fast" else ops.nchw_to_nhwc(zin)
            im = hv.decode_to_images(lat)
            print(f"vae {hw} {prec}: latents {rel_l2(z, z_ref):.3e} images {rel_l2(im, im_ref):.3e}", flush=True)
    cfg_u, ou = mc.make_unet(11)
    cfg_v, ov = mc.make_vae(12)
    for domain, n, inputs, kw in (("spatial", 8, [1, 5], dict(window_size=4, sliding_stride=2)), ("temporal", 8, [0, 1, 2, 3], dict(window_size=4, sliding_stride=1))):
        pv, pl, sk, cm = mc.synthetic_task(n, 64, 64, inputs, 11)
        g = torch.Generator().manual_seed(13)
        noise = {k: torch.randn(n, 4, 8, 8, generator=g).to(BF) for k in ("pixel", "skeleton", "latents")}
        tidx = torch.zeros(n, dtype=torch.int64)
        kw = dict(kw, sliding_shift=0, bidirectional=False, num_denoising_steps=1, alternation_rounds=1, guidance_scale=2.0)
        ref = OraclePipeline(ov, ou, DDIMScheduler(DDIMConfig()), torch.float32).sliding_iterative_denoise(
            pv, pl, sk, cm, None, domain, tidx, {k: v.float() for k, v in noise.items()}, **kw)
        for prec in ("fast", "parity", "fp16"):
            hp = Diffuman4DPipeline(AutoencoderKL(VAEConfig.from_dict(asdict(cfg_v)), ov.state_dict(), "cpu", prec),
                                    UNetMultiviewConditionModel(UNetConfig.from_dict(asdict(cfg_u)), ou.state_dict(), "cpu", prec), HS(HC()), "cpu")
            out = hp.sliding_iterative_denoise(pixel_values=pv, plucker_embeds=pl, skeletons=sk, cond_masks=cm, latents=None, domain=domain,
                                               timestep_indices=tidx, noise=noise, **kw)
            exact = torch.equal(out["timestep_indices"], ref["timestep_indices"])
            print(f"pipeline {domain} {prec}: latents {rel_l2(out['latents'], ref['latents']):.3e} images {rel_l2(out['images'], ref['images']):.3e} "
                  f"bookkeeping_exact={exact} dtype={out['latents'].dtype}", flush=True)


if __name__ == "__main__":
    main()
