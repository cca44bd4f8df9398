#!/usr/bin/env python
"""Error budget of the judged UNet call (SD-2.1 geometry, 72x40, F = 16, CFG batch 32): WHICH bf16 rounding points carry the
distance between a bf16-MFMA implementation and the fp32 oracle?  CPU only, test infrastructure (imports `oracle/`).

The fp32 oracle is re-run with bf16 roundings injected at chosen classes of tensors and compared with the plain fp32 oracle
output stored in tests/golden/sd21_72x40.pt (`unet_f16_spatial`):

  operands   every tensor that is an MFMA operand in the HIP path is rounded to bf16 right before its contraction and nothing
             else is: inputs of every Linear / Conv2d (= outputs of GroupNorm+SiLU, LayerNorm, GEGLU, attention), Q / K / V, and
             the un-normalised softmax probabilities P before P.V.  Accumulation, bias, residual adds, norms, softmax and the
             residual stream stay fp32.  This is the FLOOR of any path that feeds bf16 operands to the matrix unit: fp32
             activations kept between kernels cannot go below it.
  +stream    additionally every tensor the HIP path STORES in bf16 between kernels: the residual stream (resnet output, both
             transformer residual adds, proj_in / proj_out), conv1's output (GroupNorm 2's input), conv_in, down / up-sampler
             outputs.  = an emulation of the shipped HIP path (compare with the measured HIP error).
  -attn      `operands` without the attention-internal roundings (Q / K / V, P): what the 3-D / 2-D attention adds.

  fp16       `operands` with the operands rounded to fp16 (3 more mantissa bits, one MFMA per product): the budget of
             precision "fp16" (round 5) -- fp32 tensors between kernels, single-term fp16 MFMA operands.
  fp16noattn the same without the attention-internal roundings.

    python tools/error_budget.py [operands] [stream] [noattn] [fp16] [fp16noattn]   # default: first three; ~1-4 min each on 8 cores

Result of the run recorded in DESIGN.md section 3 / profiles/r03_error_budget.log.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
BF = torch.bfloat16
OPERAND_DTYPE = BF  # the type MFMA operands are rounded to: bf16 (fast precision) or fp16 (precision "fp16", the `fp16` / `fp16noattn` rows)


def r(x):
    return x.to(OPERAND_DTYPE).float()


class Policy:
    operands = False
    attn = False
    stream = False


P = Policy()


def install():
    import oracle.unet as ou

    def pre_round(_m, args):
        return (r(args[0]),) + tuple(args[1:]) if P.operands else None

    def attention_forward(self, x):
        b, l, _ = x.shape
        q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)
        d = q.shape[-1] // self.heads
        if P.operands and P.attn:
            q, k, v = r(q), r(k), r(v)
        q = q.view(b, l, self.heads, d).transpose(1, 2)
        k = k.view(b, l, self.heads, d).transpose(1, 2)
        v = v.view(b, l, self.heads, d).transpose(1, 2)
        if not (P.operands and P.attn):
            o = F.scaled_dot_product_attention(q, k, v)
        else:  # P rounded to bf16 before P.V (un-normalised, as the kernel holds it), row sums from the fp32 values
            o = torch.empty_like(q)
            scale = d ** -0.5
            step = max(1, (1 << 27) // max(l, 1))  # <= 512 MB of scores per chunk
            for bi in range(b):
                for hi in range(self.heads):
                    kk, vv = k[bi, hi], v[bi, hi]
                    for s0 in range(0, l, step):
                        s = (q[bi, hi, s0:s0 + step] @ kk.T) * scale
                        p = torch.exp(s - s.amax(dim=-1, keepdim=True))
                        o[bi, hi, s0:s0 + step] = (r(p) @ vv) / p.sum(dim=-1, keepdim=True)
        o = o.transpose(1, 2).reshape(b, l, self.heads * d)
        return self.to_out[0](o)

    def st(x):
        return r(x) if P.stream else x

    def resnet_forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = st(h)  # stored bf16 (conv1 epilogue adds bias + time embedding in fp32, then rounds)
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)  # fused into conv2's launch as a split-K partner: never stored on its own
        return st((x + h) / self.output_scale_factor)

    def block_forward(self, x, num_frames=1):
        n = self.norm1(x)
        if num_frames > 1:
            bt, hw, c = n.shape
            n = n.reshape(bt // num_frames, num_frames * hw, c)
        a = self.attn1(n)
        if num_frames > 1:
            a = a.reshape(bt, hw, c)
        x = st(a + x)
        x = st(self.ff(self.norm3(x)) + x)
        return x

    def transformer_forward(self, x, num_frames=1):
        b, c, h, w = x.shape
        residual = x
        y = self.norm(x)
        y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
        y = st(self.proj_in(y))
        for blk in self.transformer_blocks:
            y = blk(y, num_frames=num_frames)
        y = self.proj_out(y)
        y = y.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous()
        return st(y + residual)

    ou.Attention.forward = attention_forward
    ou.ResnetBlock2D.forward = resnet_forward
    ou.MultiviewTransformerBlock.forward = block_forward
    ou.TransformerMultiviewModel.forward = transformer_forward
    down_f, up_f = ou.Downsample2D.forward, ou.Upsample2D.forward
    ou.Downsample2D.forward = lambda self, x: st(down_f(self, x))
    ou.Upsample2D.forward = lambda self, x: st(up_f(self, x))
    return pre_round


def main():
    import make_golden_sd21 as mk
    which = sys.argv[1:] or ["operands", "stream", "noattn"]
    g = torch.load(ROOT / "tests" / "golden" / "sd21_72x40.pt")["unet_f16_spatial"]
    pre_round = install()
    cfg, m, _ = mk.build_unet()
    assert cfg.use_linear_projection
    for mod in m.modules():
        if isinstance(mod, (nn.Linear, nn.Conv2d)):
            mod.register_forward_pre_hook(pre_round)
    conv_in_f = m.conv_in.forward
    m.conv_in.forward = lambda x: (r(conv_in_f(x)) if P.stream else conv_in_f(x))
    x, t = mk.unet_inputs(g["num_frames"], g["n_cond"], g["seed"])
    ref = g["out"].float()
    modes = {"operands": (True, True, False), "stream": (True, True, True), "noattn": (True, False, False),
             "none": (False, False, False), "fp16": (True, True, False), "fp16noattn": (True, False, False)}
    print(f"reference: fp32 oracle output of tests/golden/sd21_72x40.pt (stored fp16); bf16-oracle yardstick {g['yard_bf16']:.3e}", flush=True)
    for name in which:
        global OPERAND_DTYPE
        OPERAND_DTYPE = torch.float16 if name.startswith("fp16") else BF
        P.operands, P.attn, P.stream = modes[name]
        t0 = time.time()
        with torch.no_grad():
            out = m(x.float(), t, domains=[g["domain"]] * 2, num_frames=g["num_frames"])
        if P.stream:
            out = r(out)
        e = float((out - ref).norm() / ref.norm())
        print(f"{name:9s} rel-L2 vs fp32 oracle = {e:.3e}   ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
