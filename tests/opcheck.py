"""Operator-level parity checks: HIP kernels (through the C ABI) vs a plain fp32 PyTorch reference
computed on the CPU from the same bf16-rounded inputs.

Used two ways: `pytest -m gpu` (tests/test_ops_gpu.py parametrises over CASES) and
`python tests/opcheck.py` on a GPU box, which runs every case, never stops at the first failure and
prints one line per case (handy because GPU round-trips are expensive).

Tolerance: outputs are bf16 (8 mantissa bits), accumulation fp32.  rel-L2 <= 4e-3 (about one bf16
ulp rms) unless the case says otherwise; max-abs is reported for information.
"""
from __future__ import annotations

import math
import sys
import traceback
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

BF = torch.bfloat16
TOL = 4e-3


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _rnd(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(BF)


def case_gemm(M, N, K, bias=True, rowbias=False, residual=False, geglu=False, silu=False, split=0, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    a = _rnd((M, K), g)
    w = _rnd(((2 * N) if geglu else N, K), g, 1.0 / math.sqrt(K))
    b = _rnd(((2 * N) if geglu else N,), g, 0.5) if bias else None
    rpr = 7
    rb = _rnd(((M + rpr - 1) // rpr, N), g) if rowbias else None
    res = _rnd((M, N), g) if residual else None
    ref = a.float() @ w.float().t()
    if b is not None:
        ref = ref + b.float()
    if geglu:
        h, gate = ref.chunk(2, dim=-1)
        ref = h * F.gelu(gate)
    if silu:
        ref = F.silu(ref)
    if rb is not None:
        ref = ref + rb.float().repeat_interleave(rpr, dim=0)[:M]
    if res is not None:
        ref = ref + res.float()
    d = "cuda"
    kw = dict(bias=b.to(d) if b is not None else None, rowbias=rb.to(d) if rb is not None else None,
              rows_per_rowbias=rpr, residual=res.to(d) if res is not None else None, geglu=geglu, silu=silu)
    if split:
        out = ops.gemm(a[:, :split].contiguous().to(d), w.to(d), a2=a[:, split:].contiguous().to(d), **kw)
    else:
        out = ops.gemm(a.to(d), w.to(d), **kw)
    return rel_l2(out, ref), float((out.float().cpu() - ref).abs().max())


def case_ff_fused(M, C=320, hidden=1280, bias=True, seed=0, strided=False, ln=False):
    """ops.FeedForward: the fused one-launch feed-forward (LayerNorm output -> GEGLU projection -> output projection + residual)
    against (a) the two-GEMM form on the same inputs -- required BIT-IDENTICAL: same products in the same order, same rounding
    of the hidden tensor -- and (b) the fp32 reference of attention.py:129-149's `ff(n) + x`."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    n = _rnd((M, C), g)
    x = _rnd((M, C), g)
    w1, w2 = _rnd((2 * hidden, C), g, 1.0 / math.sqrt(C)), _rnd((C, hidden), g, 1.0 / math.sqrt(hidden))
    b1, b2 = (_rnd((2 * hidden,), g, 0.5), _rnd((C,), g, 0.5)) if bias else (None, None)
    gam, bet = (1.0 + 0.1 * torch.randn(C, generator=g)).to(BF), (0.1 * torch.randn(C, generator=g)).to(BF)
    nn_ = F.layer_norm(n.float(), (C,), gam.float(), bet.float(), 1e-5).to(BF).float() if ln else n.float()  # norm3 folded in
    pre = nn_ @ w1.float().t() + (b1.float() if bias else 0.0)
    h, gate = pre.chunk(2, dim=-1)
    hid = (h * F.gelu(gate)).to(BF).float()  # the hidden tensor is bf16 between the two products in both forms
    ref = hid @ w2.float().t() + (b2.float() if bias else 0.0) + x.float()
    d = "cuda"
    dev = lambda t: None if t is None else t.to(d)  # noqa: E731
    ff = ops.FeedForward(dev(w1), dev(b1), dev(w2), dev(b2))
    assert ff.packed is not None, "fused feed-forward not built for this shape"
    nd, xd = dev(n), dev(x)
    if strided:  # row-strided views (column slices of wider tensors), as the model may pass
        nd = torch.cat([nd, nd], dim=1)[:, :C]
        xd = torch.cat([xd, xd], dim=1)[:, C:]
    lnp = (dev(gam), dev(bet), 1e-5) if ln else None
    old = ops.FF_FUSED
    try:
        ops.FF_FUSED = True
        fused = ff(nd, xd, ln=lnp)
        ops.FF_FUSED = False
        two = ff(nd, xd, ln=lnp)
    finally:
        ops.FF_FUSED = old
    worst = float((fused.float() - two.float()).abs().max())
    assert torch.equal(fused, two), f"fused feed-forward differs from the two-GEMM form (max abs {worst:.3e})"
    return rel_l2(fused, ref), float((fused.float().cpu() - ref).abs().max())


def case_ff_proj_fused(M, C=320, hidden=1280, bias=True, seed=0, strided=False):
    """FeedForward.after_attention: attention output projection + residual, norm3, feed-forward + residual in ONE launch against
    (a) the launches it replaces -- gemm(residual), then the fused LayerNorm + feed-forward -- to the order of fp32 additions (round 6; bit-identical
    in rounds 3-5), and (b) the fp32 reference of attention.py:88-90 + :129-149 with the same two bf16 roundings (h, the hidden tensor)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    a, x = _rnd((M, C), g), _rnd((M, C), g)
    wo = _rnd((C, C), g, 1.0 / math.sqrt(C))
    w1, w2 = _rnd((2 * hidden, C), g, 1.0 / math.sqrt(C)), _rnd((C, hidden), g, 1.0 / math.sqrt(hidden))
    bo, b1, b2 = (_rnd((C,), g, 0.5), _rnd((2 * hidden,), g, 0.5), _rnd((C,), g, 0.5)) if bias else (None, None, None)
    gam, bet = (1.0 + 0.1 * torch.randn(C, generator=g)).to(BF), (0.1 * torch.randn(C, generator=g)).to(BF)
    h = (a.float() @ wo.float().t() + (bo.float() if bias else 0.0) + x.float()).to(BF).float()
    nn_ = F.layer_norm(h, (C,), gam.float(), bet.float(), 1e-5).to(BF).float()
    pre = nn_ @ w1.float().t() + (b1.float() if bias else 0.0)
    u, gate = pre.chunk(2, dim=-1)
    hid = (u * F.gelu(gate)).to(BF).float()
    ref = hid @ w2.float().t() + (b2.float() if bias else 0.0) + h
    d = "cuda"
    dev = lambda t: None if t is None else t.to(d)  # noqa: E731
    ff = ops.FeedForward(dev(w1), dev(b1), dev(w2), dev(b2))
    assert ff.packed is not None, "fused feed-forward not built for this shape"
    ad, xd = dev(a), dev(x)
    if strided:  # row-strided views, as the model passes them (the attention output is a plain tensor, the residual may be a view)
        ad = torch.cat([ad, ad], dim=1)[:, :C]
        xd = torch.cat([xd, xd], dim=1)[:, C:]
    lnp = (dev(gam), dev(bet), 1e-5)
    old = ops.FF_PROJ_FUSED
    try:
        ops.FF_PROJ_FUSED = True
        one = ff.after_attention(ad, dev(wo), dev(bo), xd, lnp)
        ops.FF_PROJ_FUSED = False
        three = ff.after_attention(ad, dev(wo), dev(bo), xd, lnp)
    finally:
        ops.FF_PROJ_FUSED = old
    # Round 6: the one-launch form takes norm3 from the accumulators and starts the second product's accumulators from h (no h round trip):
    # same products and rounding points as the launches it replaces, fp32 sums in another order -- a bf16 rounding may fall the other way on
    # isolated elements (one ulp = 2^-8 .. 2^-7 relative), nothing more
    between = rel_l2(one.float().cpu(), three.float().cpu())
    flips = float((one != three).float().mean())
    assert between <= 1.5e-3 and flips <= 0.25, f"projection-fused launch against gemm + fused feed-forward: rel-L2 {between:.3e}, {100 * flips:.1f} % of the elements differ"
    return rel_l2(one, ref), float((one.float().cpu() - ref).abs().max())


def case_conv(B, H, W, Cin, Cout, stride=1, pad=1, pad_hi=None, upsample=False, rowbias=False, residual=False, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = _rnd((B, Cin, H, W), g)
    w = _rnd((Cout, Cin, 3, 3), g, 1.0 / math.sqrt(9 * Cin))
    b = _rnd((Cout,), g, 0.5)
    xin = x.float()
    if upsample:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    if pad_hi is not None:
        xin = F.pad(xin, (pad, pad_hi, pad, pad_hi))
        ref = F.conv2d(xin, w.float(), b.float(), stride=stride, padding=0)
    else:
        ref = F.conv2d(xin, w.float(), b.float(), stride=stride, padding=pad)
    rb = _rnd((B, Cout), g) if rowbias else None
    if rb is not None:
        ref = ref + rb.float()[:, :, None, None]
    res = _rnd(tuple(ref.shape), g) if residual else None
    if res is not None:
        ref = ref + res.float()
    d = "cuda"
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(d)
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(d)
    out = ops.conv3x3(x_nhwc, wt, bias=b.to(d), rowbias=rb.to(d) if rb is not None else None,
                      residual=res.permute(0, 2, 3, 1).contiguous().to(d) if res is not None else None, stride=stride,
                      pad=pad, pad_hi=pad_hi, upsample=upsample)
    out = out.permute(0, 3, 1, 2)
    assert tuple(out.shape) == tuple(ref.shape), (out.shape, ref.shape)
    return rel_l2(out, ref), float((out.float().cpu() - ref).abs().max())


def case_conv_up2x(B, H, W, Cin, Cout, bias=True, seed=0):
    """Upsample2D as four 2x2 phase convolutions of the low-resolution input (ops.conv_up2x).  Checked three ways:
    the phase kernels are exactly the fp32 sums of the 3x3 taps rounded once to bf16; the output against the fp32
    nearest-x2 + conv2d of the ORIGINAL weights (the reported error: it contains the one extra bf16 rounding of the
    summed weights); and against the gather kernel it replaces (both within the same bound of the fp32 result)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = _rnd((B, Cin, H, W), g)
    w = _rnd((Cout, Cin, 3, 3), g, 1.0 / math.sqrt(9 * Cin))
    b = _rnd((Cout,), g, 0.5) if bias else None
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), b.float() if bias else None, padding=1)
    d = "cuda"
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(d)
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(d)
    wp = ops.conv_up2x_prepare(wt)
    # expected phase kernels: rows py = 0: {0} | {1, 2}, py = 1: {0, 1} | {2}; same along x
    sets = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    w4 = w.float().permute(0, 2, 3, 1)  # [Cout, ky, kx, ci]
    for py in (0, 1):
        for px in (0, 1):
            exp = torch.stack([torch.stack([sum(w4[:, ky, kx] for ky in sets[py][dy] for kx in sets[px][dx])
                                            for dx in (0, 1)], dim=1) for dy in (0, 1)], dim=1)  # [Cout, dy, dx, ci]
            exp = exp.reshape(Cout, 4 * Cin).to(torch.bfloat16)
            assert torch.equal(wp[2 * py + px].cpu(), exp), f"phase kernel ({py},{px}) is not the rounded fp32 tap sum"
    out = ops.conv_up2x(x_nhwc, wp, bias=b.to(d) if bias else None).permute(0, 3, 1, 2)
    assert tuple(out.shape) == tuple(ref.shape), (out.shape, ref.shape)
    gather = ops.conv3x3(x_nhwc, wt, bias=b.to(d) if bias else None, upsample=True).permute(0, 3, 1, 2)
    e_gather = rel_l2(gather, ref)
    err = rel_l2(out, ref)
    assert e_gather <= TOL, f"gather kernel off: {e_gather}"
    return err, float((out.float().cpu() - ref).abs().max())


def case_conv_batch_invariance(B, H, W, Cin, Cout, seed=0):
    """A frame-sharded rank convolves fewer images per call: every image's result must not depend on how many
    images share the launch (tile configuration, split-K decision), BITWISE."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = _rnd((B, H, W, Cin), g).cuda()
    wt = _rnd((Cout, 9 * Cin), g, 1.0 / math.sqrt(9 * Cin)).cuda()
    b = _rnd((Cout,), g, 0.5).cuda()
    full = ops.conv3x3(x, wt, bias=b)
    worst = 0.0
    for n in (1, 2, B // 2):
        part = ops.conv3x3(x[:n].contiguous(), wt, bias=b)
        worst = max(worst, float((part.float() - full[:n].float()).abs().max()))
    return worst, worst


def case_conv_direct(B, H, W, Cin, Cout, k, stride, silu=True, seed=0, f32=False):
    """PoseEncoder layers (pose_encoder.py:14-31): thin direct convolution, channels zero-padded to multiples of 4.
    f32: the parity precision's variant (fp32 image, filters, bias, result; inputs NOT bf16-representable)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    BF = torch.float32 if f32 else torch.bfloat16
    x = torch.randn((B, Cin, H, W), generator=g) if f32 else _rnd((B, Cin, H, W), g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / math.sqrt(k * k * Cin) if f32 else _rnd((Cout, Cin, k, k), g, 1.0 / math.sqrt(k * k * Cin))
    b = _rnd((Cout,), g, 0.5).float() * (1.0 + 2.0 ** -12) if f32 else _rnd((Cout,), g, 0.5)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1).float()
    if silu:
        ref = F.silu(ref)
    cip, cop = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4
    xp = torch.zeros(B, H, W, cip, dtype=BF)
    xp[..., :Cin] = x.permute(0, 2, 3, 1)
    wp = torch.zeros(cop, k, k, cip, dtype=BF)
    wp[:Cout, :, :, :Cin] = w.permute(0, 2, 3, 1)
    bp = torch.zeros(cop, dtype=BF)
    bp[:Cout] = b
    out = ops.conv2d_direct(xp.cuda(), wp.reshape(cop, -1).contiguous().cuda(), ksize=k, bias=bp.cuda(), stride=stride,
                            pad=1, silu=silu)
    pad_ok = bool((out[..., Cout:] == 0).all())
    out = out[..., :Cout].permute(0, 3, 1, 2)
    assert tuple(out.shape) == tuple(ref.shape), (out.shape, ref.shape)
    return (rel_l2(out, ref) if pad_ok else 1.0), float((out.float().cpu() - ref).abs().max())


def case_attention(batch, heads, L, seed=0, spike=False, ramp=False, q_scaled=False, threads=None):
    """q_scaled: the kernel gets Q' = bf16(Q * scale * log2 e) (what the model's scaled to_q rows produce) through
    dm4d_attention_qscaled_kv_bf16; the reference is SDPA on Q'/(scale * log2 e), i.e. the same numbers.
    threads: torch CPU threads for the fp32 reference of the large shapes (the 256-thread GPU hosts are slower with all)."""
    from diffuman4d_amd.host import ops
    if threads:
        torch.set_num_threads(min(threads, torch.get_num_threads()))
    g = torch.Generator().manual_seed(seed)
    C = heads * 64
    qkv = _rnd((batch * L, 3 * C), g)
    fac = 0.125 * ops.LOG2E
    if q_scaled:
        qkv[:, :C] = (qkv[:, :C].float() * fac).to(qkv.dtype)
    if spike:  # force large online-softmax rescales (one key dominates late in the sequence)
        qkv[L - 3, C:2 * C] *= 8.0
    if ramp:  # logits far above anything in the first 64-key tile (> 2^60 in exp2 terms for a good share of the rows):
        qkv[100, C:2 * C] *= 64.0  # the optimistic first-tile max must be abandoned for the exact running-max loop
        qkv[L - 70, C:2 * C] *= 48.0
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]

    def heads_view(t):
        return t.float().view(batch, L, heads, 64).transpose(1, 2)

    q_ref = heads_view(q) / fac if q_scaled else heads_view(q)
    ref = F.scaled_dot_product_attention(q_ref, heads_view(k), heads_view(v))
    ref = ref.transpose(1, 2).reshape(batch * L, C)
    dq = qkv.to("cuda")
    out = ops.attention(dq[:, :C], dq[:, C:2 * C], dq[:, 2 * C:], batch, heads, L, q_scaled=q_scaled)
    return rel_l2(out, ref), float((out.float().cpu() - ref).abs().max())


def case_attention_kv_split(batch, heads, L, parts, seed=0, q_scaled=False):
    """Frame-sharded 3-D attention: each rank's queries against the all-gathered K/V must reproduce the
    unsharded result BITWISE (same key order, same tile boundaries)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    C = heads * 64
    qkv = _rnd((batch * L, 3 * C), g).to("cuda")
    full = ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch, heads, L, q_scaled=q_scaled).view(batch, L, C)
    ls = L // parts
    kv = qkv[:, C:].contiguous()  # [batch*L, 2C] = what the all-gather assembles
    worst = 0.0
    for r in range(parts):
        q_loc = qkv[:, :C].view(batch, L, C)[:, r * ls:(r + 1) * ls].reshape(batch * ls, C).contiguous()
        out = ops.attention(q_loc, kv[:, :C], kv[:, C:], batch, heads, ls, kv_seq=L, q_scaled=q_scaled).view(batch, ls, C)
        worst = max(worst, float((out.float() - full[:, r * ls:(r + 1) * ls].float()).abs().max()))
    return worst, worst


def case_groupnorm(B, HW, C1, C2, groups, silu, eps=1e-5, seed=0, two_launch=False, offset=0.5):
    """two_launch=True forces the statistics + apply kernel pair for a shape the single-launch kernel would take."""
    from diffuman4d_amd.host import lib, ops
    if two_launch:
        lib.load().dm4d_tune_set_groupnorm_resident(0)
    try:
        return _groupnorm(ops, B, HW, C1, C2, groups, silu, eps, seed, offset)
    finally:
        lib.load().dm4d_tune_set_groupnorm_resident(1)


def case_groupnorm_batch_invariant(B, HW, C1, C2, groups, seed=0):
    """Sample k of a batch-B call == the same sample normalised alone, bit for bit (both kernels); the single-launch
    and the two-launch kernels agree to rounding (different fp32 summation order of the statistics)."""
    from diffuman4d_amd.host import lib, ops
    g = torch.Generator().manual_seed(seed)
    x1 = (_rnd((B, HW, C1), g) + 0.5).cuda()
    x2 = (_rnd((B, HW, C2), g) * 2.0 - 0.25).cuda() if C2 else None
    gamma, beta = (_rnd((C1 + C2,), g) + 1.0).cuda(), _rnd((C1 + C2,), g, 0.3).cuda()
    outs = []
    try:
        for resident in (1, 0):
            lib.load().dm4d_tune_set_groupnorm_resident(resident)
            full = ops.groupnorm(x1, gamma, beta, groups, 1e-5, x2=x2, silu=True)
            for k in (0, B - 1):
                one = ops.groupnorm(x1[k:k + 1].contiguous(), gamma, beta, groups, 1e-5,
                                    x2=x2[k:k + 1].contiguous() if C2 else None, silu=True)
                assert torch.equal(one[0], full[k]), f"groupnorm depends on the batch (resident={resident}, sample {k})"
            outs.append(full.float())
    finally:
        lib.load().dm4d_tune_set_groupnorm_resident(1)
    return rel_l2(outs[0], outs[1].cpu()), float((outs[0] - outs[1]).abs().max())


def _groupnorm(ops, B, HW, C1, C2, groups, silu, eps, seed, offset=0.5):
    g = torch.Generator().manual_seed(seed)
    x1 = _rnd((B, HW, C1), g) + offset  # offset = the groups' mean in units of their standard deviation
    x2 = (_rnd((B, HW, C2), g) * 2.0 - 0.25) if C2 else None
    C = C1 + C2
    gamma, beta = _rnd((C,), g) + 1.0, _rnd((C,), g, 0.3)
    x = torch.cat([x1, x2], dim=-1) if C2 else x1
    ref = F.group_norm(x.double().permute(0, 2, 1), groups, gamma.double(), beta.double(), eps).permute(0, 2, 1).float()
    if silu:
        ref = F.silu(ref)
    d = "cuda"
    out = ops.groupnorm(x1.to(d), gamma.to(d), beta.to(d), groups, eps, x2=x2.to(d) if C2 else None, silu=silu)
    return rel_l2(out, ref), float((out.float().cpu() - ref).abs().max())


def case_layernorm(M, C, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = _rnd((M, C), g) * 2 + 0.3
    gamma, beta = _rnd((C,), g) + 1.0, _rnd((C,), g, 0.3)
    ref = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5)
    out = ops.layernorm(x.to("cuda"), gamma.to("cuda"), beta.to("cuda"), 1e-5)
    return rel_l2(out, ref), float((out.float().cpu() - ref).abs().max())


def case_softmax(M, N, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    s = _rnd((M, N), g, 4.0)
    ref = torch.softmax(s.float() * 0.3, dim=-1)
    out = ops.softmax_rows(s.to("cuda"), 0.3)
    return rel_l2(out, ref), float((out.float().cpu() - ref).abs().max())


def case_logits_softmax_f32(M, N, K, seed=0):
    """VAE mid-block attention pieces: fp32 logits straight from the GEMM (DM4D_EPI_F32OUT) and the softmax that reads
    them.  The logits must equal an fp32 matmul of the bf16 operands to accumulation order (no bf16 rounding)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    q, k = _rnd((M, K), g), _rnd((N, K), g)
    ref_s = q.float() @ k.float().t()
    s = ops.gemm(q.cuda(), k.cuda(), out_f32=True)
    assert s.dtype == torch.float32
    e_logits = rel_l2(s, ref_s)
    scale = K ** -0.5
    p = ops.softmax_rows(s, scale)
    e_p = rel_l2(p, torch.softmax(ref_s * scale, dim=-1))
    return (e_p if e_logits <= 2e-6 else 1.0), e_logits


def case_logits_softmax_f32_padded(M, L, K, seed=0):
    """The same pieces for a key count that is not a multiple of 32 (VAE images whose latent area is odd): keys padded with zero
    rows to the GEMM's K granularity, softmax over the first L columns only, the probabilities behind L left at the caller's zeros."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    q, k = _rnd((M, K), g), _rnd((L, K), g)
    Lp = (L + 31) // 32 * 32
    kp = torch.zeros(Lp, K, dtype=torch.bfloat16)
    kp[:L] = k
    ref_s = q.float() @ k.float().t()
    s = ops.gemm(q.cuda(), kp.cuda(), out_f32=True)
    e_logits = rel_l2(s[:, :L], ref_s)
    scale = K ** -0.5
    out = torch.zeros(M, Lp, dtype=torch.bfloat16, device="cuda")
    p = ops.softmax_rows(s, scale, n=L, out=out)
    assert p.data_ptr() == out.data_ptr() and float(p[:, L:].float().abs().max()) == 0.0
    e_p = rel_l2(p[:, :L], torch.softmax(ref_s * scale, dim=-1))
    return (e_p if e_logits <= 2e-6 else 1.0), e_logits


def case_plucker(n, H, W, h, w, seed=0):
    """Pluecker maps at latent resolution from the cameras (one launch) vs the reference path restated in
    oracle/plucker.py: full-resolution fp32 maps on the CPU -> F.interpolate(bilinear) -> bf16.  Both round the same
    fp32 values (to a few ulp) to bf16, so they may differ by one bf16 ulp on isolated elements."""
    from diffuman4d_amd.host import ops
    from oracle import plucker as op
    from test_plucker import cameras
    Ks, poses = cameras(n, H, W, seed)
    poses = op.calc_relative_poses(poses)
    ref = op.plucker_latents(H, W, Ks, poses, (h, w)).permute(0, 2, 3, 1)  # NHWC bf16
    out = ops.plucker_latents(Ks, poses, (H, W), (h, w), "cuda")
    assert tuple(out.shape) == (n, h, w, 6)
    diff = (out.float().cpu() - ref.float()).abs()
    ulp_ok = bool((diff <= 2.0 ** -7 * ref.float().abs().clamp(min=0.5)).all())  # <= 1 bf16 ulp at the element's magnitude
    frac_exact = float((diff == 0).float().mean())
    return (rel_l2(out, ref) if ulp_ok and frac_exact > 0.97 else 1.0), float(diff.max())


def case_temb(B, dim, seed=0):
    from diffuman4d_amd.host import ops
    from oracle.unet import timestep_embedding
    t = torch.tensor([0, 1, 56, 111, 500, 936, 999, 3][:B], dtype=torch.float32)
    ref = timestep_embedding(t, dim, True, 0)
    out = ops.timestep_embedding(t.to("cuda"), dim, True, 0.0)
    return rel_l2(out, ref), float((out.float().cpu() - ref).abs().max())


def case_silu(n, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = _rnd((n,), g, 3.0)
    out = ops.silu(x.to("cuda"))
    ref = F.silu(x.float())
    return rel_l2(out, ref), float((out.float().cpu() - ref).abs().max())


def case_layout(B, C, H, W, cpad, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = _rnd((B, C, H, W), g)
    y = ops.nchw_to_nhwc(x.to("cuda"), cpad).cpu()
    ref = torch.zeros(B, H, W, cpad, dtype=BF)
    ref[..., :C] = x.permute(0, 2, 3, 1)
    e1 = float((y.float() - ref.float()).abs().max())
    z = ops.nhwc_to_nchw(y.to("cuda"), C).cpu()
    e2 = float((z.float() - x.float()).abs().max())
    return max(e1, e2), max(e1, e2)


def case_pack_ddim(F_, HW, use_cfg, vpred, skel=True, seed=0):
    """pack_model_input + cfg_ddim_step vs a literal transcription of pipeline_diffuman4d.py:345-422."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    lat, pv = _rnd((F_, HW, 4), g), _rnd((F_, HW, 4), g)
    pl, sk = _rnd((F_, HW, 6), g, 0.5), (_rnd((F_, HW, 4), g) if skel else None)
    is_cond = torch.zeros(F_, dtype=torch.int32)
    is_cond[: max(1, F_ // 4)] = 1
    mask = (1 - is_cond).to(BF)[:, None, None].expand(F_, HW, 1).contiguous()
    cpad = 32
    d = "cuda"
    lat_d = lat.to(d)
    out = ops.pack_model_input(lat_d, pv.to(d), pl.to(d), sk.to(d) if skel else None, mask.to(d), is_cond.to(d), cpad,
                               use_cfg).cpu()
    c = is_cond.bool()
    x = lat.clone()
    x[c] = pv[c]
    parts_pos = [x, pl] + ([sk] if skel else []) + [mask]
    pos = torch.cat(parts_pos, dim=-1)
    nch = pos.shape[-1]
    ref = torch.zeros((2 if use_cfg else 1) * F_, HW, cpad, dtype=BF)
    ref[-F_:, :, :nch] = pos
    if use_cfg:
        neg_x = x.clone()
        neg_x[c] = 1.0
        neg = torch.cat([neg_x, torch.zeros_like(pl)] + ([-torch.ones_like(sk)] if skel else []) + [mask], dim=-1)
        ref[:F_, :, :nch] = neg
    e_pack = float((out.float() - ref.float()).abs().max())
    e_alias = float((lat_d.cpu().float() - x.float()).abs().max())
    # DDIM step
    npred = _rnd(((2 if use_cfg else 1) * F_, HW, 8), g)
    a_t = torch.rand(F_, generator=g) * 0.9 + 0.05
    a_p = torch.rand(F_, generator=g) * 0.9 + 0.05
    coef = torch.stack([a_t.sqrt(), (1 - a_t).sqrt(), a_p.sqrt(), (1 - a_p).sqrt()], dim=1).float().contiguous()
    gs = 2.0
    e = npred[..., :4].float()
    if use_cfg:
        e = e[:F_] + gs * (e[F_:] - e[:F_])
    xf = x.float()
    sa, sb, sap, sbp = [coef[:, i][:, None, None] for i in range(4)]
    if vpred:
        x0, eps = sa * xf - sb * e, sa * e + sb * xf
    else:
        x0, eps = (xf - sb * e) / sa, e
    new = sap * x0 + sbp * eps
    new[c] = xf[c]
    ops.cfg_ddim_step(lat_d, npred.to(d), coef.to(d), is_cond.to(d), use_cfg, gs, vpred)
    e_step = rel_l2(lat_d, new)
    ok = (e_pack == 0.0) and (e_alias == 0.0)
    return (e_step if ok else 1.0), max(e_pack, e_alias)


def case_linear_step(F_, HW, use_cfg, seed=0):
    """dm4d_cfg_linear_step_bf16 vs its definition in fp32: m = u + s (c - u); x' = a x + b m + c p; p' = d x + e m on the
    target rows (scattered through frame_idx), conditioning rows and rows outside the window untouched."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    N, gs = F_ + 3, 2.5
    lat, prev = _rnd((N, HW, 4), g), _rnd((N, HW, 4), g)
    cfg = 2 if use_cfg else 1
    npred = _rnd((cfg * F_, HW, 8), g)  # ldn = 8: only the first four channels are read
    coef = torch.randn((F_, 8), generator=g)
    coef[0, 2] = 0.0  # a first step: the stored prediction is not read
    is_cond = torch.zeros(F_, dtype=torch.int32)
    is_cond[1] = 1
    fidx = torch.randperm(N, generator=g)[:F_].to(torch.int32)
    m = npred[..., :4].float()
    m = m[:F_] + gs * (m[F_:] - m[:F_]) if use_cfg else m
    x, p = lat.float()[fidx.long()], prev.float()[fidx.long()]
    a, b, c, d, e = (coef[:, k].view(F_, 1, 1) for k in range(5))
    ref_lat, ref_prev = lat.float().clone(), prev.float().clone()
    tgt = (is_cond == 0)
    ref_lat[fidx.long()[tgt]] = (a * x + b * m + c * p)[tgt]
    ref_prev[fidx.long()[tgt]] = (d * x + e * m)[tgt]
    dl, dp = lat.cuda(), prev.cuda()
    ops.cfg_linear_step(dl, dp, npred.cuda(), coef.cuda(), is_cond.cuda(), use_cfg, gs, frame_idx=fidx.cuda())
    untouched = torch.ones(N, dtype=torch.bool)
    untouched[fidx.long()[tgt]] = False
    assert torch.equal(dl.cpu()[untouched], lat[untouched]) and torch.equal(dp.cpu()[untouched], prev[untouched]), "rows outside the targets changed"
    err = max(rel_l2(dl, ref_lat), rel_l2(dp, ref_prev))
    return err, float((dl.float().cpu() - ref_lat).abs().max())



# ---- parity precision (include/dm4d.h "Parity precision"): fp32 tensors, two-term bf16 operands -------------------------------------
# References are computed in fp64 from the SAME fp32 inputs; the bound is what two bf16 terms leave (|x - hi - lo| <= 2^-17 |x|, about
# 3e-6 rms per operand) plus fp32 accumulation: TOL_PAR.  Attention multiplies two split operands twice: its own bound.
TOL_PAR = 2e-5
TOL_PAR_ATTN = 1e-4


def _join(op, planes=2):
    """two-term operand [..., planes * C] -> fp64 value hi + lo (pattern 0)."""
    C = op.shape[-1] // planes
    return op[..., :C].double().cpu() + op[..., C:2 * C].double().cpu()


def case_par_split(M=300, C1=96, C2=0, cpad=None, silu=False, scale=1.0, pattern=0, transposed=False, seed=0):
    """ops.split: every plane must hold exactly bf16(x) / bf16(x - bf16(x)) of act(x) * scale in the documented order."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, C1, generator=g) * 3
    x2 = torch.randn(M, C2, generator=g) if C2 else None
    src = x.t().contiguous() if transposed else x
    op = ops.split(src.cuda(), x2.cuda() if C2 else None, cpad=cpad, silu=silu, scale=scale, pattern=pattern, transposed=transposed).cpu()
    Cp = cpad or (C1 + C2)
    v = torch.cat([x, x2], dim=1) if C2 else x
    if silu:
        v = F.silu(v)
    v = F.pad(v * scale, (0, Cp - v.shape[1]))
    hi = v.to(BF)
    lo = (v - hi.float()).to(BF)
    want = {0: [hi, lo], 1: [hi, lo, hi], 2: [hi, hi, lo]}[pattern]
    want = torch.cat(want, dim=1)
    assert op.shape == want.shape, (op.shape, want.shape)
    # SiLU on the device uses the hardware exp / rcp: compare the recombined value; without it the planes must match bit for bit
    if not silu:
        assert torch.equal(op, want), f"planes differ: max abs {float((op.float() - want.float()).abs().max()):.3e}"
        return 0.0, 0.0
    rec, ref = op[:, :Cp].double() + op[:, Cp:2 * Cp].double(), v.double()
    return rel_l2(rec, ref), float((rec - ref).abs().max())


def case_par_gemm(M, N, K, bias=True, rowbias=False, residual=False, geglu=False, silu=False, split_out=False, seed=0):
    """dm4d_gemm_bf16 on a two-term operand against K-duplicated weights, fp32 side inputs, fp32 or two-term output."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g)
    w = _rnd(((2 * N) if geglu else N, K), g, 1.0 / math.sqrt(K))
    b = _rnd(((2 * N) if geglu else N,), g, 0.5) if bias else None
    rpr = 7
    rb = torch.randn((M + rpr - 1) // rpr, N, generator=g) if rowbias else None
    res = torch.randn(M, N, generator=g) if residual else None
    ref = a.double() @ w.double().t()
    if b is not None:
        ref = ref + b.double()
    if geglu:
        h, gate = ref.chunk(2, dim=-1)
        ref = h * F.gelu(gate)
    if silu:
        ref = F.silu(ref)
    if rb is not None:
        ref = ref + rb.double().repeat_interleave(rpr, dim=0)[:M]
    if res is not None:
        ref = ref + res.double()
    d = "cuda"
    out = ops.gemm(ops.split(a.to(d)), ops.dup_k(w).to(d), bias=b.to(d) if b is not None else None,
                   rowbias=rb.to(d) if rb is not None else None, rows_per_rowbias=rpr, residual=res.to(d) if res is not None else None,
                   geglu=geglu, silu=silu, out_f32=not split_out, split_out=split_out)
    got = _join(out) if split_out else out.double().cpu()
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_par_conv(B, H, W, Cin, Cout, stride=1, pad=1, pad_hi=None, upsample=False, bias=True, rowbias=False, residual=False,
                  scale=1.0, seed=0):
    """dm4d_conv3x3_nhwc_bf16_flags on a two-term operand (channels [hi | lo]) against weights duplicated per tap."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = _rnd((Cout, Cin, 3, 3), g, 1.0 / math.sqrt(9 * Cin))
    b = _rnd((Cout,), g, 0.5) if bias else None
    xi = F.interpolate(x.double(), scale_factor=2, mode="nearest") if upsample else x.double()
    ph = pad if pad_hi is None else pad_hi
    ref = F.conv2d(F.pad(xi, (pad, ph, pad, ph)), w.double(), b.double() if bias else None, stride=stride)
    Ho, Wo = ref.shape[-2:]
    rb = torch.randn(B, Cout, generator=g) if rowbias else None
    res = torch.randn(B, Ho, Wo, Cout, generator=g) if residual else None
    if rb is not None:
        ref = ref + rb.double()[:, :, None, None]
    if res is not None:
        ref = ref + res.double().permute(0, 3, 1, 2)
    ref = ref * scale
    d = "cuda"
    wt = ops.dup_k(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous(), taps=9).to(d)
    out = ops.conv3x3(ops.split(x.permute(0, 2, 3, 1).contiguous().to(d)), wt, bias=b.to(d) if bias else None,
                      rowbias=rb.to(d) if rb is not None else None, residual=res.to(d) if res is not None else None, stride=stride,
                      pad=pad, pad_hi=pad_hi, upsample=upsample, out_scale=scale, out_f32=True)
    got = out.double().cpu().permute(0, 3, 1, 2)
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_par_groupnorm(B, HW, C1, C2=0, groups=32, silu=True, eps=1e-5, mean_shift=0.0, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(B, HW, C1, generator=g) * 2 + mean_shift
    x2 = torch.randn(B, HW, C2, generator=g) if C2 else None
    C = C1 + C2
    gam, bet = (1.0 + 0.1 * torch.randn(C, generator=g)).to(BF), (0.1 * torch.randn(C, generator=g)).to(BF)
    x = torch.cat([x1, x2], dim=-1) if C2 else x1
    ref = F.group_norm(x.double().permute(0, 2, 1), groups, gam.double(), bet.double(), eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    out = ops.groupnorm(x1.cuda(), gam.cuda(), bet.cuda(), groups, eps, x2=x2.cuda() if C2 else None, silu=silu)
    got = _join(out)
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_par_layernorm(M, C, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, C, generator=g) * 3 + 0.5
    gam, bet = (1.0 + 0.1 * torch.randn(C, generator=g)).to(BF), (0.1 * torch.randn(C, generator=g)).to(BF)
    ref = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-5)
    got = _join(ops.layernorm(x.cuda(), gam.cuda(), bet.cuda(), 1e-5))
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_par_softmax(M, N, Np, scale=0.05, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(M, Np, generator=g) * 40
    ref = torch.softmax(s[:, :N].double() * scale, dim=-1)
    p = ops.softmax_rows_split(s.cuda(), scale, n=N).cpu()
    assert p.shape == (M, 3 * Np) and torch.equal(p[:, :Np], p[:, 2 * Np:]), "planes [hi | lo | hi]"
    assert bool((p[:, N:Np] == 0).all()) and bool((p[:, Np + N:2 * Np] == 0).all()), "padded columns must be zero"
    got = p[:, :N].double() + p[:, Np:Np + N].double()
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_par_attention(batch, heads, L, seed=0, spike=False, qk_scale=1.0):
    """dm4d_attention_split_bf16 on the hi / lo planes a fused QKV projection leaves ([q_hi | k_hi | v_hi | q_lo | k_lo | v_lo]),
    against fp64 SDPA on the same two-term values.  spike: one late key dominates every row (forces the lazy rescale)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    C = heads * 64
    qkv = torch.randn(batch * L, 3 * C, generator=g)
    qkv[:, :2 * C] *= qk_scale
    if spike:
        kk = qkv[:, C:2 * C].view(batch, L, C)
        qq = qkv[:, :C].view(batch, L, C)
        kk[:, L - 7] = qq.mean(dim=1) * 6 + kk[:, L - 7]
    hi = qkv.to(BF)
    lo = (qkv - hi.float()).to(BF)
    val = hi.double() + lo.double()

    def hv(t):
        return t.view(batch, L, heads, 64).transpose(1, 2)
    q, k, v = (hv(val[:, i * C:(i + 1) * C]) for i in range(3))
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v
    ref = ref.transpose(1, 2).reshape(batch * L, C)
    out = ops.attention_split(torch.cat([hi, lo], dim=1).cuda(), batch, heads, L, 0.125)
    got = _join(out)
    return rel_l2(got, ref), float((got - ref).abs().max())


# ---- fp16 precision (include/dm4d.h "fp16 precision"): fp32 tensors, single-term fp16 operands ---------------------------------------
# Operands are drawn fp16-representable, references are fp64 on the SAME numbers: what is left is fp32 accumulation (outputs stored in
# fp32: TOL_H16_F32) or the one fp16 rounding of an operand output (2^-12 relative, 1.4e-4 rms: TOL_H16).  Attention rounds P as well.
F16 = torch.float16
TOL_H16_F32 = 2e-5
TOL_H16 = 4e-4
TOL_H16_ATTN = 6e-4


def _rndh(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(F16)


def case_h16_split(M=300, C1=96, C2=0, cpad=None, silu=False, scale=1.0, transposed=False, seed=0):
    """ops.split(h16=True): the plane must hold exactly fp16(act(x) * scale), columns behind C1 + C2 zero."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, C1, generator=g) * 3
    x2 = torch.randn(M, C2, generator=g) if C2 else None
    src = x.t().contiguous() if transposed else x
    op = ops.split(src.cuda(), x2.cuda() if C2 else None, cpad=cpad, silu=silu, scale=scale, transposed=transposed, h16=True).cpu()
    Cp = cpad or (C1 + C2)
    v = torch.cat([x, x2], dim=1) if C2 else x
    if silu:
        v = F.silu(v)
    v = F.pad(v * scale, (0, Cp - v.shape[1]))
    assert op.shape == v.shape and op.dtype == F16, (op.shape, op.dtype)
    if not silu:
        assert torch.equal(op, v.to(F16)), f"plane differs: max abs {float((op.float() - v).abs().max()):.3e}"
        return 0.0, 0.0
    return rel_l2(op, v.double()), float((op.double() - v.double()).abs().max())


def case_h16_gemm(M, N, K, bias=True, rowbias=False, residual=False, geglu=False, silu=False, out_f32=True, a2=0, scale_cols=0, seed=0):
    """dm4d_gemm_f16: fp16 A (optionally two sources) / W / bias, fp32 row bias / residual, fp32 or fp16 result, column-range scale."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    a = _rndh((M, K), g)
    w = _rndh(((2 * N) if geglu else N, K), g, 1.0 / math.sqrt(K))
    b = _rndh(((2 * N) if geglu else N,), g, 0.5) if bias else None
    rpr = 7
    rb = torch.randn((M + rpr - 1) // rpr, N, generator=g) if rowbias else None
    res = torch.randn(M, N, generator=g) if residual else None
    ref = a.double() @ w.double().t()
    if b is not None:
        ref = ref + b.double()
    if geglu:
        h, gate = ref.chunk(2, dim=-1)
        ref = h * F.gelu(gate)
    if silu:
        ref = F.silu(ref)
    if rb is not None:
        ref = ref + rb.double().repeat_interleave(rpr, dim=0)[:M]
    if res is not None:
        ref = ref + res.double()
    cs = 0.125 * 1.4426950408889634
    if scale_cols:
        ref = torch.cat([ref[:, :scale_cols] * cs, ref[:, scale_cols:]], dim=1)
    d = "cuda"
    k1 = K - a2
    out = ops.gemm(a[:, :k1].contiguous().to(d) if a2 else a.to(d), w.to(d), a2=a[:, k1:].contiguous().to(d) if a2 else None,
                   bias=b.to(d) if b is not None else None, rowbias=rb.to(d) if rb is not None else None, rows_per_rowbias=rpr,
                   residual=res.to(d) if res is not None else None, geglu=geglu, silu=silu, out_f32=out_f32, scale_cols=scale_cols,
                   col_scale=cs)
    assert out.dtype == (torch.float32 if out_f32 else F16)
    got = out.double().cpu()
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_h16_conv(B, H, W, Cin, Cout, stride=1, pad=1, pad_hi=None, upsample=False, bias=True, rowbias=False, residual=False,
                  scale=1.0, out_f32=True, seed=0, f32side=False):
    """dm4d_conv3x3_nhwc_f16 (incl. the split over the kernel rows on small images with a deep K).
    f32side with out_f32=False: fp32 row bias into an fp16 result (conv1 of a resnet in front of norm2)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = _rndh((B, Cin, H, W), g)
    w = _rndh((Cout, Cin, 3, 3), g, 1.0 / math.sqrt(9 * Cin))
    b = _rndh((Cout,), g, 0.5) if bias else None
    xi = F.interpolate(x.double(), scale_factor=2, mode="nearest") if upsample else x.double()
    ph = pad if pad_hi is None else pad_hi
    ref = F.conv2d(F.pad(xi, (pad, ph, pad, ph)), w.double(), b.double() if bias else None, stride=stride)
    Ho, Wo = ref.shape[-2:]
    sdt = torch.float32 if (out_f32 or f32side) else F16
    rb = torch.randn(B, Cout, generator=g).to(sdt) if rowbias else None
    res = torch.randn(B, Ho, Wo, Cout, generator=g).to(sdt) if residual else None
    if rb is not None:
        ref = ref + rb.double()[:, :, None, None]
    if res is not None:
        ref = ref + res.double().permute(0, 3, 1, 2)
    ref = ref * scale
    d = "cuda"
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(d)
    out = ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().to(d), wt, bias=b.to(d) if bias else None,
                      rowbias=rb.to(d) if rb is not None else None, residual=res.to(d) if res is not None else None, stride=stride,
                      pad=pad, pad_hi=pad_hi, upsample=upsample, out_scale=scale, out_f32=out_f32)
    assert out.dtype == (torch.float32 if out_f32 else F16)
    got = out.double().cpu().permute(0, 3, 1, 2)
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_h16_conv_up2x(B, H, W, Cin, Cout, bias=True, seed=0):
    """Upsample2D as four 2x2 phase convolutions in the fp16 precision: the phase kernels are exactly the fp32 tap sums rounded once to
    fp16, and the fp32 output against fp64 nearest-x2 + conv2d of the ORIGINAL fp16 weights (the reported error is that one extra
    rounding of the summed weights, 2^-12 relative per weight: TOL_H16 applies, not TOL_H16_F32)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = _rndh((B, Cin, H, W), g)
    w = _rndh((Cout, Cin, 3, 3), g, 1.0 / math.sqrt(9 * Cin))
    b = _rndh((Cout,), g, 0.5) if bias else None
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), w.double(), b.double() if bias else None, padding=1)
    d = "cuda"
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(d)
    wt = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(d)
    wp = ops.conv_up2x_prepare(wt)
    assert wp.dtype == F16
    sets = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    w4 = w.float().permute(0, 2, 3, 1)  # [Cout, ky, kx, ci]
    for py in (0, 1):
        for px in (0, 1):
            exp = torch.stack([torch.stack([sum(w4[:, ky, kx] for ky in sets[py][dy] for kx in sets[px][dx])
                                            for dx in (0, 1)], dim=1) for dy in (0, 1)], dim=1)
            assert torch.equal(wp[2 * py + px].cpu(), exp.reshape(Cout, 4 * Cin).to(F16)), f"phase kernel ({py},{px}) is not the rounded fp32 tap sum"
    out = ops.conv_up2x(x_nhwc, wp, bias=b.to(d) if bias else None, out_f32=True)
    assert out.dtype == torch.float32
    got = out.double().cpu().permute(0, 3, 1, 2)
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_h16_groupnorm(B, HW, C1, C2=0, groups=32, silu=True, eps=1e-5, mean_shift=0.0, seed=0, raw=False, f16in=False):
    """raw: the second output (fp16 of the un-normalised concat, the shortcut convolution's operand) must be bit for bit what ops.split makes.
    f16in: the input is an fp16 tensor (dm4d_groupnorm_nhwc_f16_f16: conv1's output in front of norm2); reference on the same fp16 numbers."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(B, HW, C1, generator=g) * 2 + mean_shift
    x2 = torch.randn(B, HW, C2, generator=g) if C2 else None
    if f16in:
        x1, x2 = x1.to(F16), (x2.to(F16) if C2 else None)
    C = C1 + C2
    gam, bet = (1.0 + 0.1 * torch.randn(C, generator=g)).to(F16), (0.1 * torch.randn(C, generator=g)).to(F16)
    x = torch.cat([x1, x2], dim=-1) if C2 else x1
    ref = F.group_norm(x.double().permute(0, 2, 1), groups, gam.double(), bet.double(), eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    out = ops.groupnorm(x1.cuda(), gam.cuda(), bet.cuda(), groups, eps, x2=x2.cuda() if C2 else None, silu=silu, raw_out=raw)
    if raw:
        out, rawv = out
        plain = ops.groupnorm(x1.cuda(), gam.cuda(), bet.cuda(), groups, eps, x2=x2.cuda() if C2 else None, silu=silu)
        assert torch.equal(out, plain), "the normalised plane must not depend on the second output"
        assert rawv.dtype == F16 and torch.equal(rawv.cpu(), x.to(F16)), "raw plane is not fp16(x1 | x2)"
    assert out.dtype == F16 and out.shape[-1] == C
    got = out.double().cpu()
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_h16_layernorm(M, C, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, C, generator=g) * 3 + 0.5
    gam, bet = (1.0 + 0.1 * torch.randn(C, generator=g)).to(F16), (0.1 * torch.randn(C, generator=g)).to(F16)
    ref = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-5)
    out = ops.layernorm(x.cuda(), gam.cuda(), bet.cuda(), 1e-5)
    assert out.dtype == F16 and out.shape == (M, C)
    got = out.double().cpu()
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_h16_ff_proj_fused(M, C=320, hidden=1280, bias=True, out_f32=False, seed=0, strided=False):
    """FeedForward.after_attention_f16 (dm4d_attn_out_ff_geglu_fused_f16): the tail of a transformer block of the fp16 precision in ONE
    launch against (a) the four launches it replaces -- the same products and rounding points, fp32 sums in another order: agreement to
    fp32 rounding (1e-6 on an fp32 result; an fp16 result may differ by one fp16 ulp where the fp32 sums straddle a rounding boundary)
    -- and (b) the fp64 reference of attention.py:88-90 + :129-149 with the two fp16 roundings (norm3's output, the hidden tensor)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    a = _rndh((M, C), g)
    x = torch.randn(M, C, generator=g) * 2
    wo = _rndh((C, C), g, 1.0 / math.sqrt(C))
    w1, w2 = _rndh((2 * hidden, C), g, 1.0 / math.sqrt(C)), _rndh((C, hidden), g, 1.0 / math.sqrt(hidden))
    bo, b1, b2 = (_rndh((C,), g, 0.5), _rndh((2 * hidden,), g, 0.5), _rndh((C,), g, 0.5)) if bias else (None, None, None)
    gam, bet = (1.0 + 0.1 * torch.randn(C, generator=g)).to(F16), (0.1 * torch.randn(C, generator=g)).to(F16)
    h = a.double() @ wo.double().t() + (bo.double() if bias else 0.0) + x.double()
    nn_ = F.layer_norm(h, (C,), gam.double(), bet.double(), 1e-5).to(F16).double()
    pre = nn_ @ w1.double().t() + (b1.double() if bias else 0.0)
    u, gate = pre.chunk(2, dim=-1)
    hid = (u * F.gelu(gate)).to(F16).double()
    ref = hid @ w2.double().t() + (b2.double() if bias else 0.0) + h
    d = "cuda"
    dev = lambda t: None if t is None else t.to(d)  # noqa: E731
    ff = ops.FeedForward(dev(w1), dev(b1), dev(w2), dev(b2))
    assert ff.packed is not None, "fused feed-forward not built for this shape"
    ad, xd = dev(a), dev(x)
    if strided:
        ad = torch.cat([ad, ad], dim=1)[:, :C]
        xd = torch.cat([xd, xd], dim=1)[:, C:]
    lnp = (dev(gam), dev(bet), 1e-5)
    old = ops.FF_PROJ_FUSED
    try:
        ops.FF_PROJ_FUSED = True
        one = ff.after_attention_f16(ad, dev(wo), dev(bo), xd, lnp, out_f32)
        ops.FF_PROJ_FUSED = False
        four = ff.after_attention_f16(ad, dev(wo), dev(bo), xd, lnp, out_f32)
    finally:
        ops.FF_PROJ_FUSED = old
    assert one.dtype == (torch.float32 if out_f32 else F16) and one.shape == (M, C)
    between = rel_l2(one.double().cpu(), four.double().cpu())
    # the hidden tensor's fp16 rounding can flip on an fp32-order difference of norm3's statistics: isolated elements, far below the
    # distance either form has to the fp64 reference
    assert between <= (2e-5 if out_f32 else 1e-4), f"one launch against four: rel-L2 {between:.3e}"
    got = one.double().cpu()
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_h16_softmax(M, N, Np, scale=0.05, seed=0):
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    s = torch.randn(M, Np, generator=g) * 40
    ref = torch.softmax(s[:, :N].double() * scale, dim=-1)
    p = ops.softmax_rows_split(s.cuda(), scale, n=N, h16=True).cpu()
    assert p.shape == (M, Np) and p.dtype == F16 and bool((p[:, N:] == 0).all()), "one fp16 plane, padded columns zero"
    got = p[:, :N].double()
    return rel_l2(got, ref), float((got - ref).abs().max())


def case_h16_attention(batch, heads, L, seed=0, spike=False, ramp=False, flat=False, threads=None, parts=0):
    """dm4d_attention_qscaled_kv_f16 against fp64/fp32 SDPA on the same fp16 numbers (Q carries scale * log2 e).  spike / ramp: late keys
    outgrow the first tile's maximum (ramp: by more than 2^24, so the workgroup must redo its rows with the exact loop); flat: every
    score equal -- the row sum is L times the largest probability.  parts > 0: queries of each part against all keys (frame sharding)
    must reproduce the unsharded result bitwise."""
    from diffuman4d_amd.host import ops
    if threads:
        torch.set_num_threads(min(threads, torch.get_num_threads()))
    g = torch.Generator().manual_seed(seed)
    C = heads * 64
    qkv = _rndh((batch * L, 3 * C), g)
    fac = 0.125 * ops.LOG2E
    if flat:
        qkv[:, :C] = 0
    if spike:
        qkv[L - 3, C:2 * C] *= 8.0
    if ramp:
        qkv[100, C:2 * C] *= 64.0
        qkv[L - 70, C:2 * C] *= 48.0
    qkv[:, :C] = (qkv[:, :C].float() * fac).to(F16)

    def hv(t):
        return t.float().view(batch, L, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(hv(qkv[:, :C]) / fac, hv(qkv[:, C:2 * C]), hv(qkv[:, 2 * C:]))
    ref = ref.transpose(1, 2).reshape(batch * L, C)
    dq = qkv.to("cuda")
    out = ops.attention(dq[:, :C], dq[:, C:2 * C], dq[:, 2 * C:], batch, heads, L, q_scaled=True)
    assert out.dtype == F16
    if parts:
        ls = L // parts
        kv = dq[:, C:].contiguous()
        full = out.view(batch, L, C)
        for r in range(parts):
            q_loc = dq[:, :C].view(batch, L, C)[:, r * ls:(r + 1) * ls].reshape(batch * ls, C).contiguous()
            o = ops.attention(q_loc, kv[:, :C], kv[:, C:], batch, heads, ls, kv_seq=L, q_scaled=True).view(batch, ls, C)
            assert torch.equal(o, full[:, r * ls:(r + 1) * ls]), f"part {r} of {parts} differs from the unsharded result"
    return rel_l2(out, ref), float((out.float().cpu() - ref).abs().max())


def case_h16_pack(F_=8, HW=30, use_cfg=True, skel=True, seed=0):
    """dm4d_pack_model_input_f32_f16: the fp16 operand of conv_in against the parity precision's two-term operand of the same call
    (hi + lo carries 16 mantissa bits of the fp32 value: the fp16 plane must be its correctly rounded fp16, up to the double-rounding
    ties hi + lo can introduce, i.e. within half an fp16 ulp plus 2^-17 relative)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    N = F_ + 3
    t = lambda c: torch.randn(N, HW, c, generator=g)  # noqa: E731
    lat, pv, pl, sk, mask = t(4), t(4), t(6), (t(4) if skel else None), (torch.rand(N, HW, 1, generator=g) > 0.5).float()
    cond = (torch.arange(F_) % 3 == 0).to(torch.int32)
    widx = torch.randperm(N, generator=g)[:F_].to(torch.int32)
    d = "cuda"
    args = lambda: (lat.clone().to(d), pv.to(d), pl.to(d), sk.to(d) if skel else None, mask.to(d), cond.to(d))  # noqa: E731
    a = ops.pack_model_input(*args(), 32, use_cfg, frame_idx=widx.to(d), h16=True)
    b = ops.pack_model_input(*args(), 32, use_cfg, frame_idx=widx.to(d))  # [hi | lo]
    rec = b[..., :32].double().cpu() + b[..., 32:].double().cpu()
    assert a.dtype == F16 and a.shape == rec.shape
    dev = (a.double().cpu() - rec).abs()
    bound = (2.0 ** -11 + 2.0 ** -16) * rec.abs() + 2.0 ** -25  # half an fp16 ulp (normal range; subnormal spacing 2^-24 below it)
    assert bool((dev <= bound).all()), f"max excess {float((dev - bound).max()):.3e}"
    nz = rec != 0
    assert bool((a.cpu()[~nz] == 0).all()), "padding / zero channels must be exact zeros"
    return rel_l2(a, rec), float(dev.max())


def case_multistep_step(f32=False, slots=3, seed=0):
    """dm4d_cfg_multistep_step_*: the general linear multistep update (UniPC with its corrector, DEIS) against its formula in fp64, two
    consecutive steps so that the stored tensors written by the first are read by the second."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    dt = torch.float32 if f32 else BF
    N, F_, HW = 6, 4, 48
    lat = torch.randn(N, HW, 4, generator=g).to(dt)
    cond = torch.tensor([1, 0, 0, 0], dtype=torch.int32)
    widx = torch.tensor([4, 0, 2, 5], dtype=torch.int32)
    rows = widx.long()
    keep = ~cond.bool()
    states = [torch.zeros(N, HW, 4, dtype=dt) for _ in range(slots)]
    d_lat, d_st = lat.clone().cuda(), [t.clone().cuda() for t in states]
    x = lat.double()
    st = [t.double() for t in states] + [torch.zeros(N, HW, 4, dtype=torch.float64)] * (3 - slots)
    worst = 0.0
    for step in range(2):
        eps = torch.randn(2 * F_, HW, 4, generator=g).to(dt)
        k = (torch.rand(F_, 16, generator=g) - 0.3).float()
        # k12 = what the stored tensors become (0: conv / s1 / xc; 1: kept; 2: shifted history -- PLMS rows): every mode over the two steps
        k[:, 12] = torch.tensor([0.0, 1.0, 2.0, 0.0] if step == 0 else [2.0, 0.0, 1.0, 2.0])
        ops.cfg_multistep_step(d_lat, d_st, eps.cuda(), k.cuda(), cond.cuda(), True, 2.0, frame_idx=widx.cuda())
        m = eps[:F_].double() + 2.0 * (eps[F_:].double() - eps[:F_].double())
        kk = [k[:, j].double()[:, None, None] for j in range(12)]
        xr, s1, s2, s3 = x[rows], st[0][rows], st[1][rows], st[2][rows]
        conv = kk[0] * xr + kk[1] * m
        xc = kk[2] * xr + kk[3] * s3 + kk[4] * s1 + kk[5] * s2 + kk[6] * conv
        xn = kk[7] * xc + kk[8] * conv + kk[9] * s1 + kk[10] * s2 + kk[11] * s3
        rnd = (lambda t: t.to(dt).double())
        x[rows[keep]] = rnd(xn)[keep]
        mode = k[:, 12].long()
        upd = keep & (mode != 1)
        if slots >= 3:
            st[2][rows[upd]] = torch.where((mode == 2)[:, None, None], s2, rnd(xc))[upd]
        if slots >= 2:
            st[1][rows[upd]] = s1[upd]
        st[0][rows[upd]] = rnd(conv)[upd]
        worst = max(worst, rel_l2(d_lat, x), *(rel_l2(d_st[j], st[j]) for j in range(slots)))
    return worst, 0.0


def case_resize_aa(N=3, C=3, H=576, W=320, h=306, w=170, seed=0):
    """dm4d_resize_aa_nchw_f32 against F.interpolate(mode="bilinear", antialias=True) on the CPU (fp32 both; summation order differs)."""
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, C, H, W, generator=g)
    ref = F.interpolate(x, size=(h, w), mode="bilinear", antialias=True, align_corners=False)
    out = ops.resize_aa(x.cuda(), (h, w)).cpu()
    return rel_l2(out, ref), float((out - ref).abs().max())


def case_par_small_kernels(seed=0):
    """fp32 forms of the kernels around the UNet / VAE calls: pack (operand of conv_in), CFG + DDIM / linear multistep steps, posterior
    sample, resize, postprocess, layout, timestep embedding -- each against its torch formula in fp64 (worst rel-L2 returned)."""
    import numpy as np
    from diffuman4d_amd.host import ops
    g = torch.Generator().manual_seed(seed)
    worst = 0.0
    N, F_, HW = 6, 4, 48
    lat, pv, sk = (torch.randn(N, HW, 4, generator=g) for _ in range(3))
    pl, mask = torch.randn(N, HW, 6, generator=g), torch.ones(N, HW, 1)
    cond = torch.tensor([1, 0, 0, 1], dtype=torch.int32)
    widx = torch.tensor([4, 0, 2, 5], dtype=torch.int32)
    mask[widx.long()[cond.bool()]] = 0.0
    dlat = lat.clone().cuda()
    x = ops.pack_model_input(dlat, pv.cuda(), pl.cuda(), sk.cuda(), mask.cuda(), cond.cuda(), 32, True, frame_idx=widx.cuda()).cpu()
    assert x.shape == (2 * F_, HW, 64)
    val = x[..., :32].double() + x[..., 32:].double()
    rows = widx.long()
    xin = torch.where(cond.bool()[:, None, None], pv[rows], lat[rows])
    pos = torch.cat([xin, pl[rows], sk[rows], mask[rows]], dim=-1)
    neg = torch.cat([torch.where(cond.bool()[:, None, None], torch.ones_like(xin), xin), torch.zeros_like(pl[rows]),
                     -torch.ones_like(sk[rows]), mask[rows]], dim=-1)
    want = F.pad(torch.cat([neg, pos]), (0, 32 - 15)).double()
    worst = max(worst, rel_l2(val, want))
    lat_after = lat.clone()
    lat_after[rows[cond.bool()]] = pv[rows[cond.bool()]]
    assert torch.equal(dlat.cpu(), lat_after), "aliasing side effect: cond rows of the latents take the image latents"
    # CFG + DDIM (epsilon and v prediction) and the linear multistep row
    eps = torch.randn(2 * F_, HW, 4, generator=g)
    coef = torch.rand(F_, 4, generator=g) * 0.8 + 0.1
    for vpred in (False, True):
        d2 = lat_after.clone().cuda()
        ops.cfg_ddim_step(d2, eps.cuda(), coef.cuda(), cond.cuda(), True, 2.0, vpred, frame_idx=widx.cuda())
        e = eps[:F_].double() + 2.0 * (eps[F_:].double() - eps[:F_].double())
        xx = lat_after[rows].double()
        sa, sb, sap, sbp = (coef[:, i].double()[:, None, None] for i in range(4))
        x0, ee = ((sa * xx - sb * e, sa * e + sb * xx) if vpred else ((xx - sb * e) / sa, e))
        new = sap * x0 + sbp * ee
        ref = lat_after.clone().double()
        ref[rows[~cond.bool()]] = new[~cond.bool()]
        worst = max(worst, rel_l2(d2.cpu().double(), ref))
    c8 = torch.rand(F_, 8, generator=g)
    d3, p3 = lat_after.clone().cuda(), torch.randn(N, HW, 4, generator=g)
    dp = p3.clone().cuda()
    ops.cfg_linear_step(d3, dp, eps.cuda(), c8.cuda(), cond.cuda(), True, 2.0, frame_idx=widx.cuda())
    m = eps[:F_].double() + 2.0 * (eps[F_:].double() - eps[:F_].double())
    a, b, c, dd, ee = (c8[:, i].double()[:, None, None] for i in range(5))
    xx, pp = lat_after[rows].double(), p3[rows].double()
    ref_x, ref_p = lat_after.clone().double(), p3.clone().double()
    ref_x[rows[~cond.bool()]] = (a * xx + b * m + c * pp)[~cond.bool()]
    ref_p[rows[~cond.bool()]] = (dd * xx + ee * m)[~cond.bool()]
    worst = max(worst, rel_l2(d3.cpu().double(), ref_x), rel_l2(dp.cpu().double(), ref_p))
    # posterior sample
    mom, nz = torch.randn(5, 7, 8, generator=g), torch.randn(5, 7, 4, generator=g)
    z = ops.vae_sample(mom.cuda(), nz.cuda(), 4, 0.18215).cpu().double()
    ref = (mom[..., :4].double() + torch.exp(0.5 * mom[..., 4:].double().clamp(-30, 20)) * nz.double()) * 0.18215
    worst = max(worst, rel_l2(z, ref))
    # resize (bilinear / nearest), postprocess, layout
    img = torch.randn(2, 6, 64, 48, generator=g)
    for mode in ("bilinear", "nearest"):
        o = ops.resize_to_nhwc(img.cuda(), (8, 6), mode, out_f32=True).cpu().permute(0, 3, 1, 2).double()
        worst = max(worst, rel_l2(o, F.interpolate(img.double(), size=(8, 6), mode=mode)))
    y = torch.randn(2, 5, 7, 8, generator=g)
    worst = max(worst, rel_l2(ops.postprocess_images(y.cuda(), 3).cpu().double(), (y[..., :3].double() / 2 + 0.5).clamp(0, 1).permute(0, 3, 1, 2)))
    assert torch.equal(ops.nhwc_to_nchw(y.cuda(), 6).cpu(), y[..., :6].permute(0, 3, 1, 2).contiguous())
    # timestep embedding: fp32 trigonometry of arguments up to 1000 -- the comparison is against the fp64 formula, so the bound is
    # the argument's fp32 resolution (6e-5 at t = 999), not the output format
    t = torch.tensor([0.0, 1.0, 37.0, 999.0])
    emb = ops.timestep_embedding(t.cuda(), 320, True, 0.0, out_f32=True).cpu().double()
    k = torch.arange(160, dtype=torch.float64)
    arg = t.double()[:, None] * torch.exp(-math.log(10000.0) * k / 160.0)[None]
    e_t = rel_l2(emb, torch.cat([torch.cos(arg), torch.sin(arg)], dim=1))
    assert e_t < 2e-4, e_t
    del np
    return worst, e_t


CASES = {
    # --- GEMM: every tile config, tails, epilogues -------------------------------------------
    "gemm_256x128_plain": (case_gemm, dict(M=1024, N=256, K=320)),
    "gemm_big_tiles": (case_gemm, dict(M=4096 * 3, N=1280, K=640, residual=True)),
    "gemm_n64_tiles": (case_gemm, dict(M=4096 * 6 + 40, N=320, K=320, residual=True)),
    # N = 320 on a tall problem: the 128x160 tiles (two workgroups per CU), ragged last row tile; split A on the same tile
    "gemm_n320_tall": (case_gemm, dict(M=256 * 257 + 40, N=320, K=320, residual=True)),
    "gemm_n320_tall_rowbias_k1280": (case_gemm, dict(M=256 * 256 + 8, N=320, K=1280, rowbias=True)),
    "gemm_small": (case_gemm, dict(M=77, N=320, K=64)),
    "gemm_mtail_ntail": (case_gemm, dict(M=333, N=200, K=96, residual=True, rowbias=True)),
    "gemm_n4": (case_gemm, dict(M=500, N=4, K=288, bias=True)),
    "gemm_nobias": (case_gemm, dict(M=512, N=960, K=320, bias=False)),
    "gemm_geglu": (case_gemm, dict(M=700, N=1280, K=320, geglu=True)),
    "gemm_geglu_small": (case_gemm, dict(M=130, N=256, K=64, geglu=True)),
    "gemm_silu": (case_gemm, dict(M=32, N=1280, K=320, silu=True)),
    "gemm_split_a": (case_gemm, dict(M=300, N=640, K=1280 + 640, split=1280, residual=True)),
    "gemm_rowbias": (case_gemm, dict(M=2880 * 2, N=320, K=320, rowbias=True)),
    # --- conv3x3 -------------------------------------------------------------------------------
    "conv_s1": (case_conv, dict(B=2, H=18, W=10, Cin=64, Cout=128, rowbias=True)),
    "conv_s1_res": (case_conv, dict(B=3, H=9, W=5, Cin=128, Cout=64, residual=True)),
    "conv_s2": (case_conv, dict(B=2, H=18, W=10, Cin=64, Cout=64, stride=2)),
    "conv_s2_odd": (case_conv, dict(B=2, H=9, W=5, Cin=32, Cout=64, stride=2)),
    "conv_up": (case_conv, dict(B=2, H=9, W=5, Cin=64, Cout=64, upsample=True)),
    "conv_up2x_128x64": (case_conv_up2x, dict(B=2, H=9, W=5, Cin=64, Cout=64)),
    "conv_up2x_128x128": (case_conv_up2x, dict(B=3, H=18, W=10, Cin=128, Cout=128)),
    "conv_up2x_256x128_l1": (case_conv_up2x, dict(B=32, H=36, W=20, Cin=640, Cout=640)),
    "conv_up2x_l3": (case_conv_up2x, dict(B=5, H=9, W=5, Cin=1280, Cout=1280, seed=1)),
    "conv_up2x_w1": (case_conv_up2x, dict(B=2, H=7, W=1, Cin=64, Cout=128, seed=2)),
    "conv_up2x_h1_nobias": (case_conv_up2x, dict(B=3, H=1, W=6, Cin=64, Cout=72, bias=False, seed=3)),
    "conv_up2x_vae_512": (case_conv_up2x, dict(B=1, H=72, W=40, Cin=512, Cout=512, seed=4)),
    "conv_vae_down": (case_conv, dict(B=2, H=16, W=12, Cin=32, Cout=32, stride=2, pad=0, pad_hi=1)),
    "conv_cin32_cout4": (case_conv, dict(B=4, H=36, W=20, Cin=32, Cout=4)),
    "conv_big": (case_conv, dict(B=8, H=36, W=20, Cin=320, Cout=320, rowbias=True, residual=True)),
    # stride-1 strip kernels (horizontal tap reuse): one case per tile configuration the heuristic picks
    "conv_strip_128x64": (case_conv, dict(B=10, H=36, W=20, Cin=64, Cout=320, rowbias=True)),
    "conv_strip_256x128": (case_conv, dict(B=256, H=9, W=5, Cin=64, Cout=640, residual=True)),
    "conv_strip_256x256": (case_conv, dict(B=16, H=36, W=20, Cin=64, Cout=1024)),
    "conv_strip_128x128": (case_conv, dict(B=11, H=36, W=20, Cin=128, Cout=640, rowbias=True, residual=True)),
    "conv_strip_w1": (case_conv, dict(B=2, H=5800, W=1, Cin=64, Cout=640)),
    # split over the three kernel rows (small image, deep K: the 9x5 level) + reduce/epilogue launch
    # the 160-wide strip tiles of level 0 (N = 320 on a tall problem): 128x160 with a ragged last row tile, 256x160 likewise
    "conv_strip_128x160": (case_conv, dict(B=23, H=72, W=40, Cin=64, Cout=320, rowbias=True, residual=True)),
    "conv_strip_256x160": (case_conv, dict(B=31, H=72, W=40, Cin=128, Cout=320, residual=True)),
    "conv_splitk": (case_conv, dict(B=5, H=9, W=5, Cin=512, Cout=320, rowbias=True, residual=True)),
    "conv_splitk_ragged": (case_conv, dict(B=3, H=9, W=5, Cin=640, Cout=200)),
    "conv_splitk_8x8": (case_conv, dict(B=9, H=8, W=8, Cin=512, Cout=128, residual=True)),
    "conv_batch_invariance_l3": (case_conv_batch_invariance, dict(B=8, H=9, W=5, Cin=512, Cout=256)),
    "conv_batch_invariance_l2": (case_conv_batch_invariance, dict(B=8, H=18, W=10, Cin=128, Cout=256)),
    # --- thin direct convs (PoseEncoder) ---------------------------------------------------------
    "convd_3to3_k3": (case_conv_direct, dict(B=2, H=40, W=24, Cin=3, Cout=3, k=3, stride=1)),
    "convd_3to16_k4s2": (case_conv_direct, dict(B=2, H=40, W=24, Cin=3, Cout=16, k=4, stride=2)),
    "convd_16to32_k4s2": (case_conv_direct, dict(B=3, H=20, W=12, Cin=16, Cout=32, k=4, stride=2)),
    "convd_32to64_k4s2_odd": (case_conv_direct, dict(B=2, H=11, W=7, Cin=32, Cout=64, k=4, stride=2)),
    "convd_64to128_k3": (case_conv_direct, dict(B=2, H=9, W=5, Cin=64, Cout=128, k=3, stride=1, silu=False)),
    "par_convd_3to16_k4s2": (case_conv_direct, dict(B=2, H=40, W=24, Cin=3, Cout=16, k=4, stride=2, f32=True)),
    "par_convd_32to64_k4s2_odd": (case_conv_direct, dict(B=2, H=11, W=7, Cin=32, Cout=64, k=4, stride=2, f32=True)),
    "par_convd_64to128_k3": (case_conv_direct, dict(B=2, H=9, W=5, Cin=64, Cout=128, k=3, stride=1, silu=False, f32=True)),
    # --- attention -------------------------------------------------------------------------------
    # fused level-0 feed-forward: one 128-row tile, a ragged last tile, several tiles, no biases, row-strided operands, and the
    # judged shape (CFG batch 32 at 72x40: M = 92 160)
    "ff_fused_128": (case_ff_fused, dict(M=128)),
    "ff_fused_tail": (case_ff_fused, dict(M=300, seed=1)),
    "ff_fused_small_hidden": (case_ff_fused, dict(M=257, hidden=96, seed=2)),
    "ff_fused_nobias": (case_ff_fused, dict(M=640, bias=False, seed=3)),
    "ff_fused_strided": (case_ff_fused, dict(M=384, strided=True, seed=4)),
    "ff_fused_judged": (case_ff_fused, dict(M=32 * 2880, seed=5)),
    "ff_fused_ln_tail": (case_ff_fused, dict(M=300, ln=True, seed=6)),
    "ff_fused_ln_judged": (case_ff_fused, dict(M=32 * 2880, ln=True, seed=7)),
    # the same launch with the attention output projection + residual in front (the whole tail of a transformer block)
    "ff_proj_fused_128": (case_ff_proj_fused, dict(M=128)),
    "ff_proj_fused_tail": (case_ff_proj_fused, dict(M=300, seed=1)),
    "ff_proj_fused_nobias_small_hidden": (case_ff_proj_fused, dict(M=257, hidden=96, bias=False, seed=2)),
    "ff_proj_fused_strided": (case_ff_proj_fused, dict(M=384, strided=True, seed=4)),
    "ff_proj_fused_judged": (case_ff_proj_fused, dict(M=32 * 2880, seed=5)),
    "attn_small": (case_attention, dict(batch=2, heads=2, L=128)),
    "attn_tail45": (case_attention, dict(batch=3, heads=1, L=45)),
    # tile-count edge cases of the software-pipelined loop (64-key tiles, look-ahead, tail mask on the last one)
    "attn_L64": (case_attention, dict(batch=2, heads=1, L=64)),
    "attn_L65": (case_attention, dict(batch=2, heads=1, L=65)),
    "attn_L129": (case_attention, dict(batch=1, heads=2, L=129)),
    "attn_L191": (case_attention, dict(batch=1, heads=1, L=191)),
    "attn_L256": (case_attention, dict(batch=1, heads=1, L=256)),
    # 5 / 6 / 7 / 8 tiles: first trips of the unclamped two-step main loop, with and without a ragged last tile
    "attn_L320": (case_attention, dict(batch=1, heads=2, L=320)),
    "attn_L321": (case_attention, dict(batch=1, heads=1, L=321)),
    "attn_L383": (case_attention, dict(batch=2, heads=1, L=383)),
    "attn_L448": (case_attention, dict(batch=1, heads=1, L=448)),
    "attn_L512": (case_attention, dict(batch=1, heads=2, L=512)),
    "attn_tail720": (case_attention, dict(batch=2, heads=3, L=720)),
    "attn_2d": (case_attention, dict(batch=8, heads=5, L=2880)),
    "attn_3d": (case_attention, dict(batch=2, heads=10, L=4320)),
    "attn_spike": (case_attention, dict(batch=1, heads=2, L=1000, spike=True)),
    "attn_ramp_fallback": (case_attention, dict(batch=2, heads=2, L=1500, ramp=True)),
    "attn_kv_split": (case_attention_kv_split, dict(batch=2, heads=2, L=16 * 180, parts=8)),
    "attn_kv_split3": (case_attention_kv_split, dict(batch=2, heads=1, L=24 * 45, parts=3)),
    # the entry the model uses: Q pre-scaled by scale * log2(e) (QK^T accumulator starts from -rowmax)
    "attn_qs_small": (case_attention, dict(batch=2, heads=2, L=128, q_scaled=True)),
    "attn_qs_tail45": (case_attention, dict(batch=3, heads=1, L=45, q_scaled=True)),
    "attn_qs_L65": (case_attention, dict(batch=2, heads=1, L=65, q_scaled=True)),
    "attn_qs_L321": (case_attention, dict(batch=1, heads=1, L=321, q_scaled=True)),
    "attn_qs_L448": (case_attention, dict(batch=1, heads=1, L=448, q_scaled=True)),
    "attn_qs_tail720": (case_attention, dict(batch=2, heads=3, L=720, q_scaled=True)),
    "attn_qs_2d": (case_attention, dict(batch=8, heads=5, L=2880, q_scaled=True)),
    "attn_qs_3d": (case_attention, dict(batch=2, heads=10, L=4320, q_scaled=True)),
    "attn_qs_spike": (case_attention, dict(batch=1, heads=2, L=1000, spike=True, q_scaled=True)),
    "attn_qs_ramp_fallback": (case_attention, dict(batch=2, heads=2, L=1500, ramp=True, q_scaled=True)),
    "attn_qs_kv_split": (case_attention_kv_split, dict(batch=2, heads=2, L=16 * 180, parts=8, q_scaled=True)),
    # whole-tile key counts = the hand-placed 4 x 64 kernel (attn64_kernel): its minimum of three tiles, a query tail inside the last
    # workgroup, the optimistic soft-max under a late spike, and the in-kernel exact-loop fallback
    "attn_qs_L192": (case_attention, dict(batch=2, heads=1, L=192, q_scaled=True)),
    "attn_qs_L256": (case_attention, dict(batch=1, heads=2, L=256, q_scaled=True)),
    "attn_qs_L320": (case_attention, dict(batch=3, heads=1, L=320, q_scaled=True, seed=3)),
    "attn_qs_spike1024": (case_attention, dict(batch=1, heads=2, L=1024, spike=True, q_scaled=True)),
    "attn_qs_ramp_fallback1536": (case_attention, dict(batch=2, heads=2, L=1536, ramp=True, q_scaled=True)),
    "attn_qs_kv_split_f24": (case_attention_kv_split, dict(batch=2, heads=1, L=24 * 720, parts=8, q_scaled=True)),
    # the shapes the bench TIMES (72x40 latents, SD-2.1 heads; 3-D = CFG batch 2 over F*HW tokens, 2-D = CFG*F frames) ...
    "attn_qs_judged_3d_l1_f16": (case_attention, dict(batch=2, heads=10, L=16 * 720, q_scaled=True, threads=32)),
    "attn_qs_judged_3d_l1_f24": (case_attention, dict(batch=2, heads=10, L=24 * 720, q_scaled=True, threads=32)),
    "attn_judged_3d_l1_f24": (case_attention, dict(batch=2, heads=10, L=24 * 720, threads=32)),
    "attn_qs_judged_2d_l0": (case_attention, dict(batch=32, heads=5, L=2880, q_scaled=True, threads=32)),
    "attn_judged_2d_l0": (case_attention, dict(batch=32, heads=5, L=2880, threads=32)),
    "attn_qs_judged_3d_l2_f24": (case_attention, dict(batch=2, heads=20, L=24 * 180, q_scaled=True, threads=32)),
    # ... and one sequence of the 128x128 grid (1024^2 images, level-1 3-D attention at F = 16: 16 * 64 * 64 tokens)
    "attn_qs_128sq_L65536": (case_attention, dict(batch=1, heads=1, L=65536, q_scaled=True, threads=32)),
    "attn_128sq_L65536": (case_attention, dict(batch=1, heads=1, L=65536, threads=32)),
    # --- norms -----------------------------------------------------------------------------------
    "gn_320": (case_groupnorm, dict(B=4, HW=720, C1=320, C2=0, groups=32, silu=True)),
    "gn_concat_1920": (case_groupnorm, dict(B=3, HW=180, C1=1280, C2=640, groups=32, silu=True)),
    "gn_concat_2560": (case_groupnorm, dict(B=2, HW=45, C1=1280, C2=1280, groups=32, silu=False, eps=1e-6)),
    "gn_320_two_launch": (case_groupnorm, dict(B=4, HW=720, C1=320, C2=0, groups=32, silu=True, two_launch=True)),
    "gn_concat_1920_two_launch": (case_groupnorm, dict(B=3, HW=180, C1=1280, C2=640, groups=32, silu=True, two_launch=True)),
    "gn_l0_2880": (case_groupnorm, dict(B=2, HW=2880, C1=320, C2=0, groups=32, silu=True)),  # too big for registers
    "gn_l0_concat_640_b8": (case_groupnorm, dict(B=8, HW=2880, C1=320, C2=320, groups=32, silu=True)),
    "gn_l0_concat_960": (case_groupnorm, dict(B=2, HW=2880, C1=640, C2=320, groups=32, silu=True)),
    "gn_batch_invariant_l0": (case_groupnorm_batch_invariant, dict(B=8, HW=2880, C1=320, C2=0, groups=32)),
    "gn_l1_640_b16": (case_groupnorm, dict(B=16, HW=720, C1=640, C2=0, groups=32, silu=True)),  # XCD-grouped grid
    "gn_l1_concat_1280_b8": (case_groupnorm, dict(B=8, HW=720, C1=640, C2=640, groups=32, silu=False)),
    "gn_l1_concat_960": (case_groupnorm, dict(B=2, HW=720, C1=640, C2=320, groups=32, silu=True)),  # 60-B groups: two-launch
    "gn_l3_1280_b32": (case_groupnorm, dict(B=32, HW=45, C1=1280, C2=0, groups=32, silu=True)),
    "gn_batch_invariant_l1": (case_groupnorm_batch_invariant, dict(B=9, HW=720, C1=640, C2=0, groups=32)),
    "gn_batch_invariant_l2_concat": (case_groupnorm_batch_invariant, dict(B=16, HW=180, C1=1280, C2=640, groups=32)),
    # |mean| >> std: the statistics are taken of (x - shift), see norm.hip; raw fp32 sums lose 11 / 15 bits at 50 / 200 sigma
    "gn_mean_50sigma": (case_groupnorm, dict(B=2, HW=720, C1=640, C2=0, groups=32, silu=False, offset=50.0)),
    "gn_mean_50sigma_l0": (case_groupnorm, dict(B=2, HW=2880, C1=320, C2=0, groups=32, silu=False, offset=50.0)),
    "gn_mean_200sigma": (case_groupnorm, dict(B=2, HW=720, C1=640, C2=0, groups=32, silu=False, offset=-200.0)),
    "gn_mean_200sigma_l0_concat": (case_groupnorm, dict(B=2, HW=2880, C1=320, C2=320, groups=32, silu=True, offset=200.0)),
    "gn_tiny64": (case_groupnorm, dict(B=5, HW=100, C1=64, C2=0, groups=32, silu=True)),
    "gn_tiny_concat": (case_groupnorm, dict(B=2, HW=50, C1=128, C2=64, groups=32, silu=True)),
    "ln_320": (case_layernorm, dict(M=1000, C=320)),
    "ln_1280": (case_layernorm, dict(M=77, C=1280)),
    "ln_64": (case_layernorm, dict(M=130, C=64)),
    "ln_320_tall_odd_rows": (case_layernorm, dict(M=9001, C=320)),   # tall problems, odd row count (round 6 measured two / four rows per wave on these: null, profiles/r06_ln.log)
    "ln_640_tall": (case_layernorm, dict(M=8200, C=640, seed=1)),
    "ln_1280_tall": (case_layernorm, dict(M=8193, C=1280, seed=2)),
    "softmax": (case_softmax, dict(M=50, N=2880)),
    "logits_softmax_f32": (case_logits_softmax_f32, dict(M=320, N=2880, K=512)),
    "logits_softmax_f32_ragged": (case_logits_softmax_f32, dict(M=77, N=200, K=64)),
    "logits_softmax_f32_padded_odd": (case_logits_softmax_f32_padded, dict(M=70, L=1353, K=128)),
    "logits_softmax_f32_padded_mod4_2": (case_logits_softmax_f32_padded, dict(M=33, L=1030, K=64)),
    # --- conditioning prep on the device ---------------------------------------------
    "plucker_576x320": (case_plucker, dict(n=6, H=576, W=320, h=72, w=40)),
    "plucker_odd_ratio": (case_plucker, dict(n=3, H=100, W=60, h=7, w=5, seed=2)),
    "plucker_identity_size": (case_plucker, dict(n=2, H=16, W=24, h=16, w=24, seed=3)),
    # --- small kernels ---------------------------------------------------------------------------
    "temb": (case_temb, dict(B=8, dim=320)),
    "silu": (case_silu, dict(n=32 * 1280 + 3)),
    "layout": (case_layout, dict(B=3, C=4, H=9, W=5, cpad=8)),
    "pack_ddim_cfg_eps": (case_pack_ddim, dict(F_=16, HW=45, use_cfg=True, vpred=False)),
    "pack_ddim_nocfg_v": (case_pack_ddim, dict(F_=8, HW=30, use_cfg=False, vpred=True)),
    "pack_ddim_noskel": (case_pack_ddim, dict(F_=8, HW=30, use_cfg=True, vpred=False, skel=False)),
    # --- parity precision: fp32 tensors, two-term operands (TOL_PAR) ---------------------------------------------------------
    "par_split": (case_par_split, dict()),
    "par_split_concat_pad_scale": (case_par_split, dict(M=77, C1=4, C2=0, cpad=32, scale=1.0 / 0.18215)),
    "par_split_two_sources": (case_par_split, dict(M=333, C1=64, C2=32)),
    "par_split_silu": (case_par_split, dict(M=64, C1=320, silu=True)),
    "par_split_pattern1": (case_par_split, dict(M=100, C1=64, pattern=1)),
    "par_split_pattern2_transposed_pad": (case_par_split, dict(M=64, C1=45, cpad=64, pattern=2, transposed=True)),
    "par_gemm_resid": (case_par_gemm, dict(M=1000, N=320, K=320, residual=True)),
    "par_gemm_rowbias_tails": (case_par_gemm, dict(M=333, N=200, K=96, residual=True, rowbias=True)),
    "par_gemm_n4": (case_par_gemm, dict(M=500, N=4, K=288)),
    "par_gemm_geglu_split": (case_par_gemm, dict(M=700, N=1280, K=320, geglu=True, split_out=True)),
    "par_gemm_silu_split": (case_par_gemm, dict(M=32, N=1280, K=320, silu=True, split_out=True)),
    "par_gemm_qkv_split": (case_par_gemm, dict(M=2880, N=960, K=320, bias=False, split_out=True)),
    "par_gemm_tall_n320": (case_par_gemm, dict(M=256 * 257 + 40, N=320, K=320, residual=True)),
    "par_gemm_deep": (case_par_gemm, dict(M=1440, N=1280, K=5120, residual=True)),
    "par_conv_l0": (case_par_conv, dict(B=2, H=72, W=40, Cin=320, Cout=320, rowbias=True)),
    "par_conv_resid_scale": (case_par_conv, dict(B=2, H=36, W=20, Cin=640, Cout=640, residual=True, scale=0.5)),
    "par_conv_in": (case_par_conv, dict(B=3, H=24, W=16, Cin=32, Cout=320)),
    "par_conv_out4": (case_par_conv, dict(B=2, H=24, W=16, Cin=320, Cout=4)),
    "par_conv_9x5_deep": (case_par_conv, dict(B=4, H=9, W=5, Cin=1280, Cout=1280, rowbias=True, residual=True)),
    "par_conv_stride2": (case_par_conv, dict(B=2, H=36, W=20, Cin=320, Cout=320, stride=2)),
    "par_conv_stride2_vae_pad": (case_par_conv, dict(B=2, H=32, W=24, Cin=128, Cout=128, stride=2, pad=0, pad_hi=1)),
    "par_conv_upsample": (case_par_conv, dict(B=2, H=18, W=10, Cin=640, Cout=640, upsample=True)),
    "par_gn_silu": (case_par_groupnorm, dict(B=3, HW=720, C1=320)),
    "par_gn_two_sources": (case_par_groupnorm, dict(B=2, HW=180, C1=1280, C2=640)),
    "par_gn_vae_128ch_eps6": (case_par_groupnorm, dict(B=2, HW=4096, C1=128, eps=1e-6)),
    "par_gn_mean_200sigma": (case_par_groupnorm, dict(B=2, HW=512, C1=64, groups=8, silu=False, mean_shift=400.0)),
    "par_gn_nosilu_small": (case_par_groupnorm, dict(B=1, HW=45, C1=1280, silu=False, eps=1e-6)),
    "par_ln": (case_par_layernorm, dict(M=1000, C=320)),
    "par_ln_1280": (case_par_layernorm, dict(M=333, C=1280)),
    "par_softmax": (case_par_softmax, dict(M=96, N=2880, Np=2880)),
    "par_softmax_padded": (case_par_softmax, dict(M=33, N=1353, Np=1376)),
    "par_attn_small": (case_par_attention, dict(batch=2, heads=3, L=200)),
    "par_attn_tail": (case_par_attention, dict(batch=1, heads=2, L=333, seed=1)),
    "par_attn_spike": (case_par_attention, dict(batch=1, heads=2, L=1000, spike=True, seed=2)),
    "par_attn_large_logits": (case_par_attention, dict(batch=1, heads=1, L=512, qk_scale=3.0, seed=3)),
    "par_attn_l0_2d": (case_par_attention, dict(batch=4, heads=5, L=2880, seed=4)),
    "par_small_kernels": (case_par_small_kernels, dict()),
    # the parity attention at the judged 3-D level-1 shapes (the fast kernel's attn_qs_judged_3d_l1_f16 / _f24)
    "par_attn_judged_3d_l1_f16": (case_par_attention, dict(batch=2, heads=10, L=16 * 720, seed=5)),
    "par_attn_judged_3d_l1_f24": (case_par_attention, dict(batch=1, heads=6, L=24 * 720, seed=6)),  # 6 of the 20 (batch, head) pairs: the CPU reference is the cost
    # --- fp16 precision: fp32 tensors, single-term fp16 operands (TOL_H16*) -------------------------------------------------------
    "h16_split": (case_h16_split, dict()),
    "h16_split_concat_pad_scale": (case_h16_split, dict(M=77, C1=4, C2=0, cpad=32, scale=1.0 / 0.18215)),
    "h16_split_two_sources": (case_h16_split, dict(M=333, C1=64, C2=32)),
    "h16_split_two_sources_pad_vec8": (case_h16_split, dict(M=333, C1=64, C2=32, cpad=128, scale=0.5)),
    "h16_split_l0_down": (case_h16_split, dict(M=2 * 2880, C1=320)),
    "h16_split_silu": (case_h16_split, dict(M=64, C1=320, silu=True)),
    "h16_split_transposed_pad": (case_h16_split, dict(M=64, C1=45, cpad=64, transposed=True)),
    "h16_gemm_resid": (case_h16_gemm, dict(M=1000, N=320, K=320, residual=True)),
    "h16_gemm_rowbias_tails": (case_h16_gemm, dict(M=333, N=200, K=96, residual=True, rowbias=True)),
    "h16_gemm_n4": (case_h16_gemm, dict(M=500, N=4, K=288)),
    "h16_gemm_geglu_h16out": (case_h16_gemm, dict(M=700, N=1280, K=320, geglu=True, out_f32=False)),
    "h16_gemm_silu_h16out": (case_h16_gemm, dict(M=32, N=1280, K=320, silu=True, out_f32=False)),
    "h16_gemm_qkv_qscale": (case_h16_gemm, dict(M=2880, N=960, K=320, bias=False, out_f32=False, scale_cols=320)),
    "h16_gemm_qkv_qscale_l1": (case_h16_gemm, dict(M=1440, N=1920, K=640, bias=False, out_f32=False, scale_cols=640)),
    "h16_gemm_tall_n320": (case_h16_gemm, dict(M=256 * 257 + 40, N=320, K=320, residual=True)),
    "h16_gemm_deep": (case_h16_gemm, dict(M=1440, N=1280, K=5120, residual=True)),
    "h16_gemm_wide_k640": (case_h16_gemm, dict(M=23040, N=1280, K=640, residual=False, out_f32=False)),
    "h16_gemm_two_sources": (case_h16_gemm, dict(M=2880, N=640, K=1920, a2=640)),
    "h16_ff_proj_fused_128": (case_h16_ff_proj_fused, dict(M=128)),
    "h16_ff_proj_fused_tail_f32out": (case_h16_ff_proj_fused, dict(M=300, out_f32=True, seed=1)),
    "h16_ff_proj_fused_nobias_small_hidden": (case_h16_ff_proj_fused, dict(M=257, hidden=96, bias=False, seed=2)),
    "h16_ff_proj_fused_strided": (case_h16_ff_proj_fused, dict(M=384, strided=True, seed=4)),
    "h16_ff_proj_fused_judged": (case_h16_ff_proj_fused, dict(M=32 * 2880, seed=5)),
    "h16_conv_l0": (case_h16_conv, dict(B=2, H=72, W=40, Cin=320, Cout=320, rowbias=True)),
    "h16_conv_l0_f32rowbias_h16out": (case_h16_conv, dict(B=2, H=72, W=40, Cin=320, Cout=320, rowbias=True, out_f32=False, f32side=True)),
    "h16_conv_9x5_splitk_f32rowbias_h16out": (case_h16_conv, dict(B=4, H=9, W=5, Cin=1280, Cout=1280, rowbias=True, out_f32=False, f32side=True)),
    "h16_conv_l0_tall": (case_h16_conv, dict(B=24, H=72, W=40, Cin=320, Cout=320, rowbias=True, residual=True)),
    "h16_conv_resid_scale": (case_h16_conv, dict(B=2, H=36, W=20, Cin=640, Cout=640, residual=True, scale=0.5)),
    "h16_conv_l1_wide": (case_h16_conv, dict(B=32, H=36, W=20, Cin=640, Cout=640, rowbias=True)),
    "h16_conv_l2": (case_h16_conv, dict(B=32, H=18, W=10, Cin=1280, Cout=1280, residual=True)),
    "h16_conv_in": (case_h16_conv, dict(B=3, H=24, W=16, Cin=32, Cout=320)),
    "h16_conv_out4": (case_h16_conv, dict(B=2, H=24, W=16, Cin=320, Cout=4)),
    "h16_conv_9x5_deep_splitk": (case_h16_conv, dict(B=4, H=9, W=5, Cin=1280, Cout=1280, rowbias=True, residual=True)),
    "h16_conv_stride2": (case_h16_conv, dict(B=2, H=36, W=20, Cin=320, Cout=320, stride=2)),
    "h16_conv_stride2_vae_pad": (case_h16_conv, dict(B=2, H=32, W=24, Cin=128, Cout=128, stride=2, pad=0, pad_hi=1)),
    "h16_conv_upsample": (case_h16_conv, dict(B=2, H=18, W=10, Cin=640, Cout=640, upsample=True)),
    "h16_conv_upsample_1280": (case_h16_conv, dict(B=8, H=18, W=10, Cin=1280, Cout=1280, upsample=True)),
    "h16_conv_up2x_l1": (case_h16_conv_up2x, dict(B=32, H=36, W=20, Cin=640, Cout=640)),
    "h16_conv_up2x_l3": (case_h16_conv_up2x, dict(B=5, H=9, W=5, Cin=1280, Cout=1280, seed=1)),
    "h16_conv_up2x_h1_nobias": (case_h16_conv_up2x, dict(B=3, H=1, W=6, Cin=64, Cout=72, bias=False, seed=3)),
    "h16_gn_silu": (case_h16_groupnorm, dict(B=3, HW=720, C1=320)),
    "h16_gn_l0_two_launch": (case_h16_groupnorm, dict(B=4, HW=2880, C1=320)),
    "h16_gn_l0_concat": (case_h16_groupnorm, dict(B=2, HW=2880, C1=320, C2=320)),
    "h16_gn_l0_concat_raw": (case_h16_groupnorm, dict(B=2, HW=2880, C1=640, C2=320, raw=True)),
    "h16_gn_l2_concat_raw_resident": (case_h16_groupnorm, dict(B=3, HW=180, C1=1280, C2=640, raw=True)),
    "h16_gn_odd_raw_fallback": (case_h16_groupnorm, dict(B=2, HW=77, C1=66, groups=6, raw=True)),
    "h16_gn_two_sources": (case_h16_groupnorm, dict(B=2, HW=180, C1=1280, C2=640)),
    "h16_gn_vae_128ch_eps6": (case_h16_groupnorm, dict(B=2, HW=4096, C1=128, eps=1e-6)),
    "h16_gn_mean_200sigma": (case_h16_groupnorm, dict(B=2, HW=512, C1=64, groups=8, silu=False, mean_shift=400.0)),
    "h16_gn_f16in_l0_two_launch": (case_h16_groupnorm, dict(B=4, HW=2880, C1=320, f16in=True)),
    "h16_gn_f16in_l2_resident": (case_h16_groupnorm, dict(B=3, HW=180, C1=1280, f16in=True, seed=1)),
    "h16_gn_f16in_l3": (case_h16_groupnorm, dict(B=5, HW=45, C1=1280, f16in=True, seed=2)),
    "h16_gn_f16in_mean_50sigma": (case_h16_groupnorm, dict(B=2, HW=512, C1=64, groups=8, silu=False, mean_shift=100.0, f16in=True)),
    "h16_gn_odd_channels": (case_h16_groupnorm, dict(B=2, HW=77, C1=66, groups=6, silu=True)),
    "h16_ln": (case_h16_layernorm, dict(M=1000, C=320)),
    "h16_ln_1280": (case_h16_layernorm, dict(M=333, C=1280)),
    "h16_ln_320_tall_odd_rows": (case_h16_layernorm, dict(M=9001, C=320, seed=3)),
    "h16_ln_640_tall": (case_h16_layernorm, dict(M=8200, C=640, seed=4)),
    "h16_softmax": (case_h16_softmax, dict(M=96, N=2880, Np=2880)),
    "h16_softmax_padded": (case_h16_softmax, dict(M=33, N=1353, Np=1376)),
    "h16_attn_small": (case_h16_attention, dict(batch=2, heads=2, L=128)),
    "h16_attn_tail45": (case_h16_attention, dict(batch=3, heads=1, L=45)),
    "h16_attn_L65": (case_h16_attention, dict(batch=2, heads=1, L=65)),
    "h16_attn_L321": (case_h16_attention, dict(batch=1, heads=1, L=321)),
    "h16_attn_tail720": (case_h16_attention, dict(batch=2, heads=3, L=720)),
    "h16_attn_2d": (case_h16_attention, dict(batch=8, heads=5, L=2880)),
    "h16_attn_spike": (case_h16_attention, dict(batch=1, heads=2, L=1000, spike=True)),
    "h16_attn_ramp_fallback": (case_h16_attention, dict(batch=2, heads=2, L=1500, ramp=True)),
    "h16_attn_flat_rows": (case_h16_attention, dict(batch=1, heads=1, L=24 * 720, flat=True, threads=32)),
    "h16_attn_kv_split": (case_h16_attention, dict(batch=2, heads=2, L=16 * 180, parts=8)),
    "h16_attn_L192": (case_h16_attention, dict(batch=2, heads=1, L=192)),
    "h16_attn_L320": (case_h16_attention, dict(batch=3, heads=1, L=320, seed=3)),
    "h16_attn_spike1024": (case_h16_attention, dict(batch=1, heads=2, L=1024, spike=True)),
    "h16_attn_ramp_fallback1536": (case_h16_attention, dict(batch=2, heads=2, L=1536, ramp=True)),
    "h16_attn_judged_3d_l1_f16": (case_h16_attention, dict(batch=2, heads=10, L=16 * 720, threads=32)),
    "h16_attn_judged_3d_l1_f24": (case_h16_attention, dict(batch=2, heads=10, L=24 * 720, threads=32)),
    "h16_attn_judged_2d_l0": (case_h16_attention, dict(batch=32, heads=5, L=2880, threads=32)),
    "h16_pack_cfg": (case_h16_pack, dict()),
    "h16_pack_nocfg_noskel": (case_h16_pack, dict(F_=5, HW=17, use_cfg=False, skel=False, seed=1)),
    "par_multistep_step_3slots": (case_multistep_step, dict(f32=True, slots=3)),
    "multistep_step_3slots": (case_multistep_step, dict(slots=3)),
    "multistep_step_2slots": (case_multistep_step, dict(slots=2, seed=1)),
    # the result writer's antialiased down-scale (mosaic of a 48-view spatial task; of a 300-frame temporal task; an up-scale axis)
    "resize_aa_spatial_mosaic": (case_resize_aa, dict()),
    "resize_aa_temporal_mosaic": (case_resize_aa, dict(N=4, H=576, W=320, h=48, w=27)),
    "resize_aa_mixed": (case_resize_aa, dict(N=2, C=1, H=50, W=30, h=17, w=45)),
}

TOLS = {"multistep_step_3slots": 1e-6, "multistep_step_2slots": 1e-6, "resize_aa_spatial_mosaic": 5e-5, "resize_aa_temporal_mosaic": 5e-5, "resize_aa_mixed": 5e-5, "plucker_576x320": 2e-3, "plucker_odd_ratio": 2e-3, "plucker_identity_size": 2e-3, "layout": 0.0, "temb": 6e-3, "attn_kv_split": 0.0, "attn_kv_split3": 0.0, "attn_qs_kv_split": 0.0, "attn_qs_kv_split_f24": 0.0,
        "conv_batch_invariance_l3": 0.0, "conv_batch_invariance_l2": 0.0}


def run_case(name):
    fn, kw = CASES[name]
    err, mx = fn(**kw)
    default = (TOL_PAR_ATTN if name.startswith("par_attn") else TOL_PAR) if name.startswith("par_") else TOL
    if name.startswith("h16_"):  # fp16 precision: an fp16 result carries its one rounding, an fp32 result only the accumulation
        kw = CASES[name][1]
        f32_out = name.startswith(("h16_gemm", "h16_conv")) and kw.get("out_f32", True) and not name.startswith("h16_conv_up2x")
        default = TOL_H16_ATTN if name.startswith("h16_attn") else (TOL_H16_F32 if f32_out else TOL_H16)
    return err, mx, TOLS.get(name, default)


def main():
    torch.manual_seed(0)
    bad = 0
    prefixes = tuple(sys.argv[1:]) or ("",)  # e.g. `opcheck.py attn par_attn`
    names = [n for n in CASES if n.startswith(prefixes)]
    for name in names:
        try:
            err, mx, tol = run_case(name)
            ok = err <= tol and math.isfinite(err)
            print(f"{'PASS' if ok else 'FAIL'} {name:24s} rel_l2={err:.3e} max_abs={mx:.3e} tol={tol:.1e}", flush=True)
            bad += 0 if ok else 1
        except Exception as e:  # keep going: one GPU trip should report everything
            bad += 1
            print(f"ERROR {name}: {type(e).__name__}: {e}", flush=True)
            traceback.print_exc()
    print(f"opcheck: {len(names) - bad}/{len(names)} passed", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
