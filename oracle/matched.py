"""Rounding-matched oracle of the FAST precision.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The fp32 oracle (``oracle/unet.py``) is the truth the fast HIP path is judged against through a loose, self-referential bound
(1.15 x the bf16-oracle yardstick, about 1e-2): a defect worth 3e-3 -- a wrong eps, a mis-rounded epilogue -- passes it.  This
module restates the SAME forward pass with a bf16 rounding at exactly the points where the HIP path rounds -- every tensor it stores
between launches, nothing else -- so that the HIP output can be compared with it DIRECTLY under a fixed, tight bound (tests/modelcheck.py
``*_matched`` cases).  What remains between the two is fp32 summation order, the hardware's exp2 / rcp, and the bf16 roundings those flip.

Rounding points of the fast path (diffuman4d_amd/host/unet.py over libdm4d.so; every contraction accumulates in fp32 and applies its
epilogue -- bias, SiLU / GEGLU, time-embedding row bias, residual, output scale -- in fp32 BEFORE the single rounding of its output):
  * time embedding: sinusoid -> bf16; linear_1 + SiLU -> bf16; linear_2 (+ temporal embedding residual) -> bf16; SiLU -> bf16;
    every resnet's time_emb_proj -> bf16
  * conv_in, Downsample2D conv, conv_out -> bf16
  * ResnetBlock2D: GroupNorm + SiLU -> bf16; conv1 + bias + temb -> bf16; GroupNorm + SiLU -> bf16; conv_shortcut (its own GEMM) ->
    bf16; (conv2 + bias + shortcut) / scale -> bf16
  * TransformerMultiviewModel: GroupNorm -> bf16; proj_in -> bf16; proj_out + residual -> bf16
  * MultiviewTransformerBlock: LayerNorm -> bf16; fused QKV projection -> bf16, the to_q rows carrying scale * log2(e) folded into the
    WEIGHTS before their bf16 rounding (deviation 2 of DESIGN.md section 3); attention: S in fp32, P = exp2(S - m) with m the row
    maximum over the FIRST 64 keys (the kernel's optimistic pass), row sums of the unrounded P, P -> bf16 for P V, O / l -> bf16;
    to_out + residual -> bf16; LayerNorm -> bf16; GEGLU hidden -> bf16; ff output + residual -> bf16
  * Upsample2D: the four 2x2 phase kernels, weights = fp32 sums of the 3x3 taps rounded to bf16 ONCE MORE (deviation 1) -> bf16
Follows the module tree of oracle/unet.py (itself citing unet_multiview_condition.py:501-598, unet_multiview_blocks.py,
transformer_multiview.py:157-216, attention.py:68-149).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn.functional as F

from . import up2x
from .unet import UNetMultiviewConditionModel, timestep_embedding

BF = torch.bfloat16
LOG2E = 1.4426950408889634


def r(x: torch.Tensor) -> torch.Tensor:
    """One bf16 rounding (round-to-nearest-even), value kept in fp32."""
    return x.to(BF).float()


def _lin(mod, x):
    return F.linear(x, mod.weight.float(), None if mod.bias is None else mod.bias.float())


def _conv(mod, x, **kw):
    return F.conv2d(x, mod.weight.float(), None if mod.bias is None else mod.bias.float(), **kw)


def _gn(mod, x):
    return F.group_norm(x, mod.num_groups, mod.weight.float(), mod.bias.float(), mod.eps)


def _ln(mod, x):
    return F.layer_norm(x, mod.normalized_shape, mod.weight.float(), mod.bias.float(), mod.eps)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, first_tile: int = 64, row_chunk: int = 2048) -> torch.Tensor:
    """q (already carrying scale * log2 e), k, v: [b, L, heads * d] bf16-valued fp32 -> O [b, L, heads * d], rounded.
    csrc/attention.hip: exp2 of (S - m), m = max over the first 64 keys; l from the unrounded P; P rounded for P V; O / l rounded."""
    b, L, C = q.shape
    d = C // heads
    qh, kh, vh = (t.view(b, L, heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    out = torch.empty_like(qh)
    for bi in range(b):
        for hi in range(heads):
            kk, vv = kh[bi, hi], vh[bi, hi]
            for s0 in range(0, L, row_chunk):
                s = qh[bi, hi, s0:s0 + row_chunk] @ kk.T
                m = s[:, :first_tile].amax(dim=-1, keepdim=True)
                p = torch.exp2(s - m)
                out[bi, hi, s0:s0 + row_chunk] = (r(p) @ vv) / p.sum(dim=-1, keepdim=True)
    return r(out.permute(0, 2, 1, 3).reshape(b, L, C))


def resnet(mod, x, tp):
    """x: the (concatenated) input, bf16-valued; tp: this resnet's rounded time-embedding projection [B, Cout]."""
    h = r(F.silu(_gn(mod.norm1, x)))
    h = r(_conv(mod.conv1, h, padding=1) + tp[:, :, None, None])
    h = r(F.silu(_gn(mod.norm2, h)))
    sc = r(_conv(mod.conv_shortcut, x)) if mod.conv_shortcut is not None else x
    return r((_conv(mod.conv2, h, padding=1) + sc) / mod.output_scale_factor)


def block(mod, y, num_frames, prescaled_q=True):
    n = r(_ln(mod.norm1, y))
    a1 = mod.attn1
    d = a1.to_q.weight.shape[0] // a1.heads
    if prescaled_q:
        wq = r(a1.to_q.weight.float() * (d ** -0.5 * LOG2E))
        q = r(F.linear(n, wq))
    else:  # the checkpoint's rows; the scale is applied to the fp32 scores (the kernel's un-folded entry)
        q = r(_lin(a1.to_q, n)) * (d ** -0.5 * LOG2E)
    k, v = r(_lin(a1.to_k, n)), r(_lin(a1.to_v, n))
    if num_frames > 1:  # attention.py:69-71
        bt, hw, c = n.shape
        q, k, v = (t.reshape(bt // num_frames, num_frames * hw, c) for t in (q, k, v))
    a = attention(q, k, v, a1.heads)
    if num_frames > 1:
        a = a.reshape(bt, hw, c)
    h = r(_lin(a1.to_out[0], a) + y)
    n3 = r(_ln(mod.norm3, h))
    u, g = _lin(mod.ff.net[0].proj, n3).chunk(2, dim=-1)
    hid = r(u * F.gelu(g))
    return r(_lin(mod.ff.net[2], hid) + h)


def transformer(mod, x, num_frames, prescaled_q=True):
    assert mod.use_linear_projection, "the HIP path loads linear-projection checkpoints"
    b, c, h, w = x.shape
    y = r(_gn(mod.norm, x)).permute(0, 2, 3, 1).reshape(b, h * w, c)
    y = r(_lin(mod.proj_in, y))
    for blk in mod.transformer_blocks:
        y = block(blk, y, num_frames, prescaled_q)
    y = _lin(mod.proj_out, y).reshape(b, h, w, -1).permute(0, 3, 1, 2)
    return r(y + x)


def upsample(mod, x, summed_weights=True):
    w, b = mod.conv.weight.float(), mod.conv.bias.float()
    cin, cout = w.shape[1], w.shape[0]
    if summed_weights and cin % 64 == 0 and cout % 8 == 0:  # ops.conv_up2x_supported: the phase kernels
        wp = r(up2x.phase_weights(w))
        B, _, H, W = x.shape
        y = x.new_zeros((B, cout, 2 * H, 2 * W))
        for py in (0, 1):
            for px in (0, 1):
                y[:, :, py::2, px::2] = F.conv2d(F.pad(x, (1 - px, px, 1 - py, py)), wp[py, px])
        return r(y + b[None, :, None, None])
    return r(F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1))


@torch.no_grad()
def unet_forward(m: UNetMultiviewConditionModel, sample: torch.Tensor, timestep: torch.Tensor, domains: Sequence[str] = ("spatial",),
                 num_frames: int = 1, prescaled_q: bool = True, summed_upsample_weights: bool = True) -> torch.Tensor:
    """oracle/unet.py::UNetMultiviewConditionModel.forward with the fast path's roundings.  `m` holds fp32 parameters whose values
    are bf16-representable (the checkpoint as the HIP path loads it); `sample` is rounded on entry (the packer writes bf16)."""
    cfg = m.cfg
    if cfg.enable_pose_encoder:
        raise NotImplementedError("matched oracle: enable_pose_encoder checkpoints are not covered")
    boc0 = cfg.block_out_channels[0]
    sample = r(sample.float())
    t_emb = r(timestep_embedding(timestep.expand(sample.shape[0]), boc0, cfg.flip_sin_to_cos, cfg.freq_shift))
    te = m.time_embedding
    emb = _lin(te.linear_2, r(F.silu(_lin(te.linear_1, t_emb))))
    if cfg.enable_tem_embeds:
        if len(domains) * num_frames != len(emb):
            raise ValueError("num_frames * len(domains) != len(emb)")
        idx = []
        for dmn in domains:
            if dmn == "spatial":
                idx.append(torch.zeros(num_frames))
            elif dmn == "temporal":
                idx.append(torch.arange(num_frames // 2).repeat(2).float())
            else:
                raise ValueError(f"Invalid domain for temporal embedding: {dmn}")
        f_emb = r(timestep_embedding(torch.cat(idx), boc0, True, 0))
        tp_ = m.temporal_pos_embed
        emb = _lin(tp_.linear_2, r(F.silu(_lin(tp_.linear_1, f_emb)))) + r(emb)  # the residual enters the epilogue as a stored tensor
    emb = r(emb)
    semb = r(F.silu(emb))

    def tproj(res):
        return r(_lin(res.time_emb_proj, semb))

    x = r(_conv(m.conv_in, sample, padding=1))
    skips = [x]
    n_down = len(m.down_blocks)
    for i, blk in enumerate(m.down_blocks):
        nf = num_frames if (blk.has_cross_attention and n_down - i - 1 < cfg.num_3d_attn_blocks) else 1
        for j, res in enumerate(blk.resnets):
            x = resnet(res, x, tproj(res))
            if blk.has_cross_attention:
                x = transformer(blk.attentions[j], x, nf, prescaled_q)
            skips.append(x)
        if blk.downsamplers is not None:
            ds = blk.downsamplers[0]
            x = r(_conv(ds.conv, x, stride=2, padding=1))
            skips.append(x)
    mb = m.mid_block
    x = resnet(mb.resnets[0], x, tproj(mb.resnets[0]))
    x = transformer(mb.attentions[0], x, num_frames, prescaled_q)
    x = resnet(mb.resnets[1], x, tproj(mb.resnets[1]))
    for i, blk in enumerate(m.up_blocks):
        nf = num_frames if (blk.has_cross_attention and i < cfg.num_3d_attn_blocks) else 1
        for j, res in enumerate(blk.resnets):
            x = resnet(res, torch.cat([x, skips.pop()], dim=1), tproj(res))
            if blk.has_cross_attention:
                x = transformer(blk.attentions[j], x, nf, prescaled_q)
        if blk.upsamplers is not None:
            x = upsample(blk.upsamplers[0], x, summed_upsample_weights)
    x = r(F.silu(_gn(m.conv_norm_out, x)))
    return r(_conv(m.conv_out, x, padding=1))


# ------------------------------------------------------------------------------------------------------------------------------
# AutoencoderKL and the pipeline around the UNet, with the fast path's roundings (diffuman4d_amd/host/vae.py, pipeline.py)
# ------------------------------------------------------------------------------------------------------------------------------
def vae_resnet(mod, x):
    h = r(F.silu(_gn(mod.norm1, x)))
    h = r(_conv(mod.conv1, h, padding=1))
    h = r(F.silu(_gn(mod.norm2, h)))
    sc = r(_conv(mod.conv_shortcut, x)) if mod.conv_shortcut is not None else x
    return r((_conv(mod.conv2, h, padding=1) + sc) / mod.output_scale_factor)


def vae_attention(mod, x, q_block: int = 2048):
    """host/vae.py::_MidAttn: GroupNorm -> bf16; fused q/k/v projection -> bf16; LOGITS STAY fp32 (DM4D_EPI_F32OUT); the normalised
    probabilities -> bf16 (dm4d_softmax_rows_f32in_bf16); P V -> bf16; output projection + residual -> bf16."""
    b, c, h, w = x.shape
    y = r(_gn(mod.group_norm, x)).view(b, c, h * w).transpose(1, 2)
    q, k, v = r(_lin(mod.to_q, y)), r(_lin(mod.to_k, y)), r(_lin(mod.to_v, y))
    o = torch.empty_like(q)
    for bi in range(b):
        for s0 in range(0, h * w, q_block):
            p = r(torch.softmax((q[bi, s0:s0 + q_block] @ k[bi].T) * (float(c) ** -0.5), dim=-1))
            o[bi, s0:s0 + q_block] = r(p @ v[bi])
    out = _lin(mod.to_out[0], o)
    return r(out.transpose(-1, -2).reshape(b, c, h, w) + x)


def vae_mid(mod, x):
    return vae_resnet(mod.resnets[1], vae_attention(mod.attentions[0], vae_resnet(mod.resnets[0], x)))


@torch.no_grad()
def vae_moments(v, images):
    """oracle/vae.py::AutoencoderKL.moments with the fast path's roundings; images NCHW in [-1, 1]."""
    e = v.encoder
    x = r(_conv(e.conv_in, r(images.float()), padding=1))
    for blk in e.down_blocks:
        for res in blk.resnets:
            x = vae_resnet(res, x)
        if blk.downsamplers is not None:
            x = r(_conv(blk.downsamplers[0].conv, F.pad(x, (0, 1, 0, 1)), stride=2))
    x = vae_mid(e.mid_block, x)
    x = r(_conv(e.conv_out, r(F.silu(_gn(e.conv_norm_out, x))), padding=1))
    return r(_conv(v.quant_conv, x))


@torch.no_grad()
def vae_encode_scaled(v, images, noise):
    mean, logvar = vae_moments(v, images).chunk(2, dim=1)
    return r((mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * r(noise.float())) * v.cfg.scaling_factor)


@torch.no_grad()
def vae_decode_to_images(v, latents):
    """latents (x scaling_factor, bf16-valued) -> images in [0, 1] as decode_to_images returns them (bf16-valued)."""
    d = v.decoder
    z = r(r(latents.float()) * float(1.0 / v.cfg.scaling_factor))
    x = r(_conv(d.conv_in, r(_conv(v.post_quant_conv, z)), padding=1))
    x = vae_mid(d.mid_block, x)
    for blk in d.up_blocks:
        for res in blk.resnets:
            x = vae_resnet(res, x)
        if blk.upsamplers is not None:
            x = upsample(blk.upsamplers[0], x)
    x = r(_conv(d.conv_out, r(F.silu(_gn(d.conv_norm_out, x))), padding=1))
    return r((x * 0.5 + 0.5).clamp(0.0, 1.0))


class MatchedPipeline:
    """oracle/pipeline.py::OraclePipeline.sliding_iterative_denoise with the fast HIP pipeline's roundings: VAE and UNet as above; the
    conditioning maps resized in fp32 and rounded once; per window call the packer writes bf16, and ONE kernel forms the guided
    prediction and the DDIM update in fp32 from the bf16 noise prediction and latents and rounds the new latents once
    (csrc/elementwise.hip::cfg_ddim_kernel; pipeline_diffuman4d.py:408-422)."""

    def __init__(self, vae, unet, scheduler):
        self.vae, self.unet, self.scheduler = vae, unet, scheduler
        self.device, self.dtype = torch.device("cpu"), torch.float32

    def _encode(self, x, noise, batch_size=8):
        return torch.cat([vae_encode_scaled(self.vae, xb, nb) for xb, nb in zip(x.split(batch_size), noise.split(batch_size))])

    def post_process(self, latents, batch_size=8):
        return torch.cat([vae_decode_to_images(self.vae, lb) for lb in latents.split(batch_size)])

    @torch.no_grad()
    def sliding_iterative_denoise(self, pixel_values, plucker_embeds, skeletons, cond_masks, latents, domain, timestep_indices, noise,
                                  window_size=12, sliding_stride=1, sliding_shift=0, bidirectional=True, num_denoising_steps=1,
                                  alternation_rounds=3, guidance_scale=2.0, decode=True, trace=None):
        from .pipeline import build_windows, steps_per_alternation
        per_alt = steps_per_alternation(window_size, sliding_stride, bidirectional, num_denoising_steps)
        num_inference_steps = per_alt * alternation_rounds
        timestep_indices = timestep_indices.clone()
        target_indices = torch.where(cond_masks[:, 0, 0, 0] != 0.0)[0]
        input_indices = torch.where(cond_masks[:, 0, 0, 0] == 0.0)[0]
        id_end = int(timestep_indices[target_indices][0]) + per_alt
        pv_lat = self._encode(pixel_values, noise["pixel"])
        n, _, h, w = pv_lat.shape
        pl_lat = r(F.interpolate(plucker_embeds.float(), size=(h, w), mode="bilinear"))
        sk_lat = self._encode(skeletons, noise["skeleton"])
        cm_lat = r(F.interpolate(cond_masks.float(), size=(h, w), mode="nearest"))
        lat = r((noise["latents"] if latents is None else latents).float())
        timesteps = self.scheduler.set_timesteps(num_inference_steps)
        is_cond_all = cm_lat[:, 0, 0, 0] == 0
        tws, iws = build_windows(target_indices, input_indices, domain, window_size, sliding_stride, sliding_shift, bidirectional)
        sch, do_cfg = self.scheduler, guidance_scale > 1
        for tw, iw in zip(tws, iws):
            win = torch.cat([iw, tw])
            cond = is_cond_all[win]
            tidx = timestep_indices[win].clone()
            for _ in range(num_denoising_steps):
                tidx[cond] = 0
                t = timesteps[tidx].clone()
                t[cond] = 0
                lat[win[cond]] = pv_lat[win[cond]]  # the packer's aliasing side effect (:375-379)
                x = lat[win]
                pos = torch.cat([x, pl_lat[win], sk_lat[win], cm_lat[win]], dim=1)
                if do_cfg:
                    neg = torch.cat([torch.where(cond[:, None, None, None], torch.ones_like(x), x), torch.zeros_like(pl_lat[win]),
                                     -torch.ones_like(sk_lat[win]), cm_lat[win]], dim=1)
                    eps = unet_forward(self.unet, torch.cat([neg, pos]), torch.cat([t, t]), domains=[domain] * 2, num_frames=len(win))
                    u, c = eps.chunk(2)
                    e = u + guidance_scale * (c - u)
                else:
                    e = unet_forward(self.unet, pos, t, domains=[domain], num_frames=len(win))
                new = x.clone()
                for j in range(len(win)):
                    if not cond[j]:
                        a_t, a_prev = (float(v) for v in sch.coefficients(int(t[j])))
                        sa, sb, sap, sbp = (torch.tensor(v, dtype=torch.float32).sqrt() for v in (a_t, 1 - a_t, a_prev, 1 - a_prev))
                        if sch.cfg.prediction_type == "epsilon":
                            x0, ee = (x[j] - sb * e[j]) / sa, e[j]
                        else:
                            x0, ee = sa * x[j] - sb * e[j], sa * e[j] + sb * x[j]
                        new[j] = r(sap * x0 + sbp * ee)
                lat[win] = new
                tidx[~cond] += 1
            timestep_indices[tw] += num_denoising_steps
        if (timestep_indices[target_indices] != id_end).any() or (timestep_indices[input_indices] != 0).any():
            raise ValueError("matched pipeline: the timestep bookkeeping went wrong")
        images = self.post_process(lat) if decode else None
        return {"images": images, "latents": lat, "timestep_indices": timestep_indices,
                "fully_denoised": timestep_indices == num_inference_steps}


def _self_check():  # python -m oracle.matched: the un-rounded walk of this module equals oracle/unet.py's forward
    global r
    from .unet import UNetConfig, init_unet_weights
    cfg = UNetConfig.tiny(enable_tem_embeds=True)
    m = UNetMultiviewConditionModel(cfg).eval()
    init_unet_weights(m, 0)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in m.temporal_pos_embed.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    x = torch.randn(8, cfg.in_channels, 16, 8, generator=g)
    t = torch.randint(0, 1000, (8,), generator=g)
    with torch.no_grad():
        ref = m(x, t, domains=["temporal"] * 2, num_frames=4)
    keep, r = r, (lambda v: v)
    try:
        out = unet_forward(m, x, t, domains=["temporal"] * 2, num_frames=4, summed_upsample_weights=True)
    finally:
        r = keep
    err = float((out - ref).norm() / ref.norm())
    print(f"matched walk without roundings vs oracle forward: rel-L2 {err:.3e}")
    assert err < 1e-5, err
    out = unet_forward(m, x, t, domains=["temporal"] * 2, num_frames=4)
    print(f"matched (fast-path roundings) vs fp32 oracle: rel-L2 {float((out - ref).norm() / ref.norm()):.3e}")


if __name__ == "__main__":
    _self_check()
    del math
