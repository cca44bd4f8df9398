"""Tensor-level wrappers over the C ABI (torch is only the allocator / stream provider here).

Every function requires tensors resident on a HIP device and launches on torch's current
stream.  Nothing here computes with torch: a tensor on the CPU is an error, not a fallback.

Three arithmetics share these wrappers (include/dm4d.h, "Parity precision" / "fp16 precision"):
  * fast (default): bf16 tensors between kernels, bf16 MFMA operands, fp32 accumulation;
  * parity (``precision="parity"`` on the model objects): fp32 tensors between kernels; every activation that feeds the
    matrix unit is a two-term OPERAND ``[hi(C) | lo(C)]`` (bf16, 2 C columns) against weights duplicated along K
    (``dup_k``);
  * fp16 (``precision="fp16"``): fp32 tensors between kernels; every activation that feeds the matrix unit is ONE fp16 plane
    against fp16 weights (one MFMA per product, as in the fast precision).
Functions that exist in several forms dispatch on the dtypes they are given: an fp32 activation means a wide (parity / fp16)
precision, and fp16 weights / norm parameters / operands mean the fp16 one.
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch

from . import lib as _l

BF16 = torch.bfloat16
F16 = torch.float16
F32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# tests/modelcheck.py (the *_opreplay cases) sets this to a list: every fast-precision launch then appends (operator name, its
# inputs as a dict of tensors / scalars, its output tensor), so that each launch of a real model call can be recomputed on the CPU
# from the very tensors the device was given (oracle/replay.py).  None = no tracing.  The tensors are kept alive by the list.
TRACE = None


def _trace(name: str, out, **inputs) -> None:
    if TRACE is not None:
        TRACE.append((name, inputs, out))


def _req(t: torch.Tensor, name: str, dtype=BF16):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _l.Dm4dError(f"{name}: expected a tensor on a HIP device (no CPU fallback in diffuman4d_amd)")
    if t.dtype != dtype:
        raise _l.Dm4dError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise _l.Dm4dError(f"{name}: last dim must be contiguous")
    return t


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def gemm(a: torch.Tensor, w: torch.Tensor, *, a2: Optional[torch.Tensor] = None, bias=None, rowbias=None,
         rows_per_rowbias: int = 1, residual=None, geglu: bool = False, silu: bool = False, out_scale: float = 1.0,
         out: Optional[torch.Tensor] = None, out_f32: bool = False, split_out: bool = False, scale_cols: int = 0,
         col_scale: float = 1.0) -> torch.Tensor:
    """out[M,N] = epi([a | a2] @ w^T); a [M,K1], a2 [M,K-K1], w [N,K] (GEGLU: [2N,K]).
    out_f32: the result is stored unrounded in an fp32 tensor (attention logits of the VAE mid block; wide precisions).
    split_out (wide precisions): the result leaves as the OPERAND of the next contraction -- parity: bf16 [M, 2N] = [hi | lo];
    fp16 (w is fp16): one fp16 plane [M, N].
    rowbias / residual may be fp32 tensors (wide precisions: both must then be fp32).
    scale_cols / col_scale (fp16 precision): output columns [0, scale_cols) are multiplied by col_scale before the one rounding."""
    if isinstance(w, torch.Tensor) and w.dtype == F16:
        return _gemm_f16(a, w, a2, bias, rowbias, rows_per_rowbias, residual, geglu, silu, out_scale, out, out_f32, scale_cols, col_scale)
    assert scale_cols == 0
    lib = _l.load()
    _req(a, "a"), _req(w, "w")
    M, K1 = a.shape
    K = w.shape[1]
    N = w.shape[0] // 2 if geglu else w.shape[0]
    if a2 is not None:
        _req(a2, "a2")
        assert a2.shape[0] == M and K1 + a2.shape[1] == K
    else:
        assert K1 == K, (K1, K)
    if bias is not None:
        _req(bias, "bias")
    side = [t for t in (rowbias, residual) if t is not None]
    f32_side = bool(side) and side[0].dtype == F32
    for t, n in ((rowbias, "rowbias"), (residual, "residual")):
        if t is not None:
            _req(t, n, F32 if f32_side else BF16)
    assert not (out_f32 and split_out)
    odt, ocols = (F32 if out_f32 else BF16), (2 * N if split_out else N)
    if out is None:
        out = torch.empty((M, ocols), dtype=odt, device=a.device)
    _req(out, "out", odt)
    assert out.shape[1] >= ocols or out.stride(0) >= ocols
    flags = ((_l.EPI_GEGLU if geglu else 0) | (_l.EPI_SILU if silu else 0) | (_l.EPI_F32OUT if out_f32 else 0)
             | (_l.EPI_F32SIDE if f32_side else 0) | (_l.EPI_SPLITOUT if split_out else 0))
    with _Prof("linear", 2.0 * M * w.shape[0] * K, "flop", M):
        rc = lib.dm4d_gemm_bf16(_stream(), _p(a), a.stride(0), _p(a2), a2.stride(0) if a2 is not None else 0,
                                K1 if a2 is not None else 0, _p(w), w.stride(0), _p(out), out.stride(0), M, N, K,
                                _p(bias), _p(rowbias), rowbias.stride(0) if rowbias is not None else 0,
                                rows_per_rowbias, _p(residual), residual.stride(0) if residual is not None else 0,
                                flags, out_scale)
    _l.check(rc, "dm4d_gemm_bf16")
    _trace("gemm", out, a=a, w=w, a2=a2, bias=bias, rowbias=rowbias, rows_per_rowbias=rows_per_rowbias, residual=residual, geglu=geglu,
           silu=silu, out_scale=out_scale, out_f32=out_f32, split_out=split_out)
    return out


def _gemm_f16(a, w, a2, bias, rowbias, rows_per_rowbias, residual, geglu, silu, out_scale, out, out_f32, scale_cols, col_scale):
    """precision "fp16": fp16 a / a2 / w / bias, fp32 (or fp16) rowbias / residual, fp32 or fp16 result (dm4d_gemm_f16)."""
    lib = _l.load()
    _req(a, "a", F16), _req(w, "w", F16)
    M, K1 = a.shape
    K = w.shape[1]
    N = w.shape[0] // 2 if geglu else w.shape[0]
    if a2 is not None:
        _req(a2, "a2", F16)
        assert a2.shape[0] == M and K1 + a2.shape[1] == K
    else:
        assert K1 == K, (K1, K)
    if bias is not None:
        _req(bias, "bias", F16)
    side = [t for t in (rowbias, residual) if t is not None]
    f32_side = bool(side) and side[0].dtype == F32
    for t, n in ((rowbias, "rowbias"), (residual, "residual")):
        if t is not None:
            _req(t, n, F32 if f32_side else F16)
    odt = F32 if out_f32 else F16
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=a.device)
    _req(out, "out", odt)
    assert out.shape[1] >= N or out.stride(0) >= N
    flags = ((_l.EPI_GEGLU if geglu else 0) | (_l.EPI_SILU if silu else 0) | (_l.EPI_F32OUT if out_f32 else 0)
             | (_l.EPI_F32SIDE if f32_side else 0))
    with _Prof("linear", 2.0 * M * w.shape[0] * K, "flop", M):
        rc = lib.dm4d_gemm_f16(_stream(), _p(a), a.stride(0), _p(a2), a2.stride(0) if a2 is not None else 0,
                               K1 if a2 is not None else 0, _p(w), w.stride(0), _p(out), out.stride(0), M, N, K,
                               _p(bias), _p(rowbias), rowbias.stride(0) if rowbias is not None else 0,
                               rows_per_rowbias, _p(residual), residual.stride(0) if residual is not None else 0,
                               flags, out_scale, int(scale_cols), float(col_scale))
    _l.check(rc, "dm4d_gemm_f16")
    return out


def conv_out_hw(h: int, w: int, stride: int, pad: int, upsample: bool, pad_hi: Optional[int] = None) -> Tuple[int, int]:
    if upsample:
        return 2 * h, 2 * w
    ph = pad if pad_hi is None else pad_hi
    return (h + pad + ph - 3) // stride + 1, (w + pad + ph - 3) // stride + 1


def conv3x3(x: torch.Tensor, wt: torch.Tensor, *, bias=None, rowbias=None, residual=None, stride: int = 1,
            pad: int = 1, pad_hi: Optional[int] = None, upsample: bool = False, out_scale: float = 1.0,
            out_f32: bool = False) -> torch.Tensor:
    """x [B,H,W,Cin] NHWC, wt [Cout, 9*Cin] ((ky,kx,ci) order) -> [B,Ho,Wo,Cout].
    out_f32 (parity precision): fp32 output; rowbias / residual must then be fp32 as well (x is usually a two-term operand
    and wt duplicated along Cin, which this function does not need to know)."""
    lib = _l.load()
    if isinstance(wt, torch.Tensor) and wt.dtype == F16:
        return _conv3x3_f16(x, wt, bias, rowbias, residual, stride, pad, pad_hi, upsample, out_scale, out_f32)
    _req(x, "x"), _req(wt, "wt")
    assert x.is_contiguous() and wt.is_contiguous()
    B, H, W, Cin = x.shape
    Cout = wt.shape[0]
    assert wt.shape[1] == 9 * Cin, (wt.shape, Cin)
    Ho, Wo = conv_out_hw(H, W, stride, pad, upsample, pad_hi)
    y = torch.empty((B, Ho, Wo, Cout), dtype=F32 if out_f32 else BF16, device=x.device)
    side = F32 if out_f32 else BF16
    if residual is not None:
        _req(residual, "residual", side)
        assert residual.numel() == y.numel() and residual.is_contiguous()
    if rowbias is not None:
        _req(rowbias, "rowbias", side)
        assert rowbias.shape[0] == B
    if out_f32:
        with _Prof("conv3x3", 2.0 * B * Ho * Wo * 9 * Cin * Cout, "flop", B * Ho * Wo):
            rc = lib.dm4d_conv3x3_nhwc_bf16_flags(_stream(), _p(x), B, H, W, Cin, _p(wt), _p(y), Ho, Wo, Cout, stride, pad,
                                                  1 if upsample else 0, _p(bias), _p(rowbias),
                                                  rowbias.stride(0) if rowbias is not None else 0, _p(residual),
                                                  Cout if residual is not None else 0, out_scale, _l.EPI_F32OUT | _l.EPI_F32SIDE)
        _l.check(rc, "dm4d_conv3x3_nhwc_bf16_flags")
        _trace("conv3x3", y, x=x, wt=wt, bias=bias, rowbias=rowbias, residual=residual, stride=stride, pad=pad, pad_hi=pad_hi,
               upsample=upsample, out_scale=out_scale, out_f32=True)
        return y
    # small images with a deep K (the 9x5 level) run split over the three kernel rows and need an fp32 workspace
    ws_bytes = lib.dm4d_conv3x3_ws_bytes(B, H, W, Cin, Ho, Wo, Cout, stride, pad, 1 if upsample else 0)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device) if ws_bytes else None
    with _Prof("conv3x3", 2.0 * B * Ho * Wo * 9 * Cin * Cout, "flop", B * Ho * Wo):
        rc = lib.dm4d_conv3x3_nhwc_bf16_ws(_stream(), _p(x), B, H, W, Cin, _p(wt), _p(y), Ho, Wo, Cout, stride, pad,
                                           1 if upsample else 0, _p(bias), _p(rowbias),
                                           rowbias.stride(0) if rowbias is not None else 0, _p(residual),
                                           Cout if residual is not None else 0, out_scale, _p(ws), ws_bytes)
    _l.check(rc, "dm4d_conv3x3_nhwc_bf16_ws")
    _trace("conv3x3", y, x=x, wt=wt, bias=bias, rowbias=rowbias, residual=residual, stride=stride, pad=pad, pad_hi=pad_hi,
           upsample=upsample, out_scale=out_scale, out_f32=False)
    return y


def _conv3x3_f16(x, wt, bias, rowbias, residual, stride, pad, pad_hi, upsample, out_scale, out_f32) -> torch.Tensor:
    """precision "fp16": x / wt / bias fp16; rowbias / residual fp32 (always with an fp32 result) or fp16; fp32 (out_f32) or fp16 result
    (an fp16 result with an fp32 row bias: conv1 of a resnet, whose fp32 sum is rounded once for norm2)."""
    lib = _l.load()
    _req(x, "x", F16), _req(wt, "wt", F16)
    assert x.is_contiguous() and wt.is_contiguous()
    B, H, W, Cin = x.shape
    Cout = wt.shape[0]
    assert wt.shape[1] == 9 * Cin, (wt.shape, Cin)
    Ho, Wo = conv_out_hw(H, W, stride, pad, upsample, pad_hi)
    y = torch.empty((B, Ho, Wo, Cout), dtype=F32 if out_f32 else F16, device=x.device)
    sides = [t for t in (rowbias, residual) if t is not None]
    side = F32 if (out_f32 or (sides and sides[0].dtype == F32)) else F16
    if bias is not None:
        _req(bias, "bias", F16)
    if residual is not None:
        _req(residual, "residual", side)
        assert residual.numel() == y.numel() and residual.is_contiguous()
    if rowbias is not None:
        _req(rowbias, "rowbias", side)
        assert rowbias.shape[0] == B
    ws_bytes = lib.dm4d_conv3x3_ws_bytes(B, H, W, Cin, Ho, Wo, Cout, stride, pad, 1 if upsample else 0)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device) if ws_bytes else None
    with _Prof("conv3x3", 2.0 * B * Ho * Wo * 9 * Cin * Cout, "flop", B * Ho * Wo):
        rc = lib.dm4d_conv3x3_nhwc_f16(_stream(), _p(x), B, H, W, Cin, _p(wt), _p(y), Ho, Wo, Cout, stride, pad,
                                       1 if upsample else 0, _p(bias), _p(rowbias),
                                       rowbias.stride(0) if rowbias is not None else 0, _p(residual),
                                       Cout if residual is not None else 0, out_scale,
                                       (_l.EPI_F32OUT if out_f32 else 0) | (_l.EPI_F32SIDE if side == F32 else 0), _p(ws), ws_bytes)
    _l.check(rc, "dm4d_conv3x3_nhwc_f16")
    return y


# ---- wide precisions: operands -------------------------------------------------------------------------------------------
def dup_k(w: torch.Tensor, taps: int = 1, times: int = 2) -> torch.Tensor:
    """Weights for a two-term operand: [N, taps * C] -> [N, taps * times * C] with every tap's C columns repeated `times` times
    ([W | W] per tap), so that  [hi | lo] [W | W]^T = (hi + lo) W^T.  Host / load-time helper (any device)."""
    n, k = w.shape
    c = k // taps
    return w.reshape(n, taps, 1, c).expand(n, taps, times, c).reshape(n, taps * times * c).contiguous()


def split(x: torch.Tensor, x2: Optional[torch.Tensor] = None, *, cpad: Optional[int] = None, silu: bool = False,
          scale: float = 1.0, pattern: int = 0, transposed: bool = False, h16: bool = False) -> torch.Tensor:
    """fp32 [..., C] (channel concat with x2) -> the OPERAND of a contraction.  Parity precision: two-term bf16 [..., 2 cpad] =
    [hi | lo] (pattern 1: [hi | lo | hi], 2: [hi | hi | lo], three planes); h16 (precision "fp16"): one fp16 plane [..., cpad].
    transposed: x is a 2-D view [K, M] read as its transpose (result [M, planes * K])."""
    lib = _l.load()
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != F32:
        raise _l.Dm4dError("split: expected an fp32 tensor on a HIP device")
    planes = 1 if h16 else (2 if pattern == 0 else 3)
    if transposed:
        assert x.ndim == 2 and x2 is None and x.stride(1) == 1
        C1, M = x.shape
        rs1, cs1, lead = 1, x.stride(0), (M,)
    else:
        assert x.stride(-1) == 1
        C1 = x.shape[-1]
        x2d = x.reshape(-1, C1) if x.is_contiguous() else x
        assert x2d.ndim == 2
        M, rs1, cs1, lead = x2d.shape[0], x2d.stride(0), 1, tuple(x.shape[:-1])
        x = x2d
    C2, rs2 = 0, 0
    if x2 is not None:
        _req(x2, "x2", F32)
        x2 = x2.reshape(-1, x2.shape[-1])
        assert x2.shape[0] == M
        C2, rs2 = x2.shape[1], x2.stride(0)
    Cp = cpad or (C1 + C2)
    y = torch.empty(lead + (planes * Cp,), dtype=F16 if h16 else BF16, device=x.device)
    with _Prof("split", 4.0 * M * (C1 + C2) + 2.0 * M * planes * Cp, "byte", M):  # fp32 in, 16-bit planes out
        if h16:
            rc = lib.dm4d_to_f16_f32(_stream(), _p(x), rs1, cs1, C1, _p(x2), rs2, C2, _p(y), Cp, M, Cp, 1 if silu else 0, scale)
        else:
            rc = lib.dm4d_split_f32(_stream(), _p(x), rs1, cs1, C1, _p(x2), rs2, C2, _p(y), planes * Cp, M, Cp, 1 if silu else 0,
                                    scale, pattern)
    _l.check(rc, "dm4d_split_f32")
    return y


UP2X_PHASE = os.environ.get("DM4D_UP2X_PHASE", "1") != "0"  # Upsample2D as four 2x2 phase convolutions; off = the gather kernel (A/B, tests)
FF_FUSED = os.environ.get("DM4D_FF_FUSED", "1") != "0"  # level-0 feed-forward (C = 320) as one launch; off = the two-GEMM form (A/B, tests)
# the attention output projection + residual as a prologue of that launch (FeedForward.after_attention); off = its own GEMM (A/B, tests)
FF_PROJ_FUSED = os.environ.get("DM4D_FF_PROJ_FUSED", "1") != "0"


class FeedForward:
    """ff(n) + residual of a transformer block (attention.py:129-149): GEGLU projection then output projection.  Where the fused
    kernel is built for the shape (dm4d_ff_geglu_supported: C = 320) the per-step packed weights are made here, at load time
    (the stream is drained before the object is handed out, as in Upsampler), and the call is ONE launch whose hidden tensor
    never leaves the chip; other shapes -- and FF_FUSED = False -- run gemm(GEGLU) + gemm(residual).  Both forms are
    bit-identical (tests/opcheck.py ff_fused_*)."""

    def __init__(self, w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, b2: Optional[torch.Tensor]):
        self.w1, self.b1, self.w2, self.b2 = w1, b1, w2, b2
        self.C, self.hidden = w2.shape[0], w2.shape[1]
        assert w1.shape == (2 * self.hidden, self.C)
        self.packed = None
        if w1.is_cuda and _l.load().dm4d_ff_geglu_supported(self.C, self.hidden):
            lib = _l.load()
            _req(w1, "w1", w1.dtype), _req(w2, "w2", w1.dtype)  # bf16, or fp16 (precision "fp16": the packing permutes 16-bit words)
            with torch.cuda.device(w1.device):
                w1p, b1p, w2p = torch.empty_like(w1), torch.empty(2 * self.hidden, dtype=w1.dtype, device=w1.device), torch.empty_like(w2)
                _l.check(lib.dm4d_ff_geglu_prepare_bf16(_stream(), _p(w1.contiguous()), _p(b1), _p(w2.contiguous()), _p(w1p), _p(b1p),
                                                        _p(w2p), self.C, self.hidden), "dm4d_ff_geglu_prepare_bf16")
                torch.cuda.current_stream(w1.device).synchronize()
            self.packed = (w1p, b1p, w2p)

    def __call__(self, n: torch.Tensor, residual: torch.Tensor, ln=None) -> torch.Tensor:
        """ln = (gamma, beta, eps): `n` is the un-normalised input and the LayerNorm in front of the feed-forward (norm3) is applied
        here -- inside the fused launch, or as its own launch in the two-GEMM form."""
        # the one-launch form wants 16-byte aligned rows; any other view takes the two GEMMs, which accept 8-element-aligned strides
        aligned = all(t is None or (t.data_ptr() % 16 == 0 and (t.ndim < 2 or t.stride(0) % 8 == 0))
                      for t in (n, residual) + (tuple(ln[:2]) if ln is not None else ()))
        if self.packed is None or not FF_FUSED or not aligned or n.shape[0] * n.stride(0) * 2 >= (1 << 32):
            if ln is not None:
                n = layernorm(n, ln[0], ln[1], ln[2])
            f = gemm(n, self.w1, bias=self.b1, geglu=True)
            return gemm(f, self.w2, bias=self.b2, residual=residual)
        lib = _l.load()
        _req(n, "n"), _req(residual, "residual")
        if ln is not None:
            _req(ln[0], "gamma"), _req(ln[1], "beta")
        M = n.shape[0]
        out = torch.empty((M, self.C), dtype=BF16, device=n.device)
        w1p, b1p, w2p = self.packed
        with _Prof("linear", 2.0 * M * 3 * self.hidden * self.C, "flop", M):
            rc = lib.dm4d_ff_geglu_fused_bf16(_stream(), _p(n), n.stride(0), _p(ln[0]) if ln is not None else None,
                                              _p(ln[1]) if ln is not None else None, float(ln[2]) if ln is not None else 0.0, _p(w1p),
                                              _p(b1p), _p(w2p), _p(self.b2), _p(residual), residual.stride(0), _p(out), out.stride(0), M,
                                              self.C, self.hidden)
        _l.check(rc, "dm4d_ff_geglu_fused_bf16")
        _trace("ff_fused", out, n=n, residual=residual, ln=ln, w1=self.w1, b1=self.b1, w2=self.w2, b2=self.b2)
        return out


    def after_attention(self, a: torch.Tensor, wo: torch.Tensor, bo: Optional[torch.Tensor], x: torch.Tensor, ln) -> torch.Tensor:
        """The tail of a transformer block: h = a wo^T + bo + x (attention output projection + residual), then ff(LayerNorm(h)) + h.
        One launch where the fused kernel is built for the shape (FF_PROJ_FUSED), gemm + __call__ otherwise: the same products and bf16
        rounding points either way (since round 6 the one-launch form adds its fp32 terms in another order: one-ulp differences on isolated
        elements, tests/opcheck.py ff_proj_fused_*)."""
        # the one-launch form wants 16-byte aligned rows (row strides in multiples of 8 elements) and a contiguous square weight; a
        # column view of a wider tensor as `a` or `x` takes the separate launches, which accept any 8-element-aligned view
        aligned = all(t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0 for t in (a, x)) and wo.data_ptr() % 16 == 0
        if (self.packed is None or not FF_FUSED or not FF_PROJ_FUSED or ln is None or not aligned
                or a.shape[0] * a.stride(0) * 2 >= (1 << 32) or wo.shape != (self.C, self.C) or not wo.is_contiguous()):
            h = gemm(a, wo, bias=bo, residual=x)
            return self(h, h, ln=ln)
        lib = _l.load()
        _req(a, "a"), _req(wo, "wo"), _req(x, "x"), _req(ln[0], "gamma"), _req(ln[1], "beta")
        M = a.shape[0]
        out = torch.empty((M, self.C), dtype=BF16, device=a.device)
        w1p, b1p, w2p = self.packed
        with _Prof("linear", 2.0 * M * (3 * self.hidden + self.C) * self.C, "flop", M):
            rc = lib.dm4d_attn_out_ff_geglu_fused_bf16(_stream(), _p(a), a.stride(0), _p(wo), _p(bo), _p(x), x.stride(0), _p(ln[0]),
                                                       _p(ln[1]), float(ln[2]), _p(w1p), _p(b1p), _p(w2p), _p(self.b2), _p(out),
                                                       out.stride(0), M, self.C, self.hidden)
        _l.check(rc, "dm4d_attn_out_ff_geglu_fused_bf16")
        _trace("attn_out_ff_fused", out, a=a, wo=wo, bo=bo, x=x, ln=ln, w1=self.w1, b1=self.b1, w2=self.w2, b2=self.b2)
        return out

    def after_attention_f16(self, a: torch.Tensor, wo: torch.Tensor, bo: Optional[torch.Tensor], x: torch.Tensor, ln, out_f32: bool) -> torch.Tensor:
        """The same tail in precision "fp16": a fp16 [M, C], x the fp32 residual stream; the result is fp32 (out_f32) or the fp16 operand
        of the next contraction.  One launch (dm4d_attn_out_ff_geglu_fused_f16: h stays in the accumulators, never stored) where the kernel
        is built for the shape, otherwise gemm -> fp32 h, layernorm, gemm(GEGLU), gemm(+ h); the two forms agree to fp32 rounding."""
        aligned = (all(t.data_ptr() % 16 == 0 for t in (a, x, wo, ln[0], ln[1])) and a.stride(0) % 8 == 0 and x.stride(0) % 4 == 0)
        if (self.packed is None or not FF_FUSED or not FF_PROJ_FUSED or not aligned or a.shape[0] * a.stride(0) * 2 >= (1 << 32)
                or wo.shape != (self.C, self.C) or not wo.is_contiguous()):
            h = gemm(a, wo, bias=bo, residual=x, out_f32=True)
            f = gemm(layernorm(h, ln[0], ln[1], ln[2]), self.w1, bias=self.b1, geglu=True)
            return gemm(f, self.w2, bias=self.b2, residual=h, out_f32=out_f32)
        lib = _l.load()
        _req(a, "a", F16), _req(wo, "wo", F16), _req(x, "x", F32), _req(ln[0], "gamma", F16), _req(ln[1], "beta", F16)
        M = a.shape[0]
        out = torch.empty((M, self.C), dtype=F32 if out_f32 else F16, device=a.device)
        w1p, b1p, w2p = self.packed
        with _Prof("linear", 2.0 * M * (3 * self.hidden + self.C) * self.C, "flop", M):
            rc = lib.dm4d_attn_out_ff_geglu_fused_f16(_stream(), _p(a), a.stride(0), _p(wo), _p(bo), _p(x), x.stride(0), _p(ln[0]),
                                                      _p(ln[1]), float(ln[2]), _p(w1p), _p(b1p), _p(w2p), _p(self.b2), _p(out),
                                                      out.stride(0), 1 if out_f32 else 0, M, self.C, self.hidden)
        _l.check(rc, "dm4d_attn_out_ff_geglu_fused_f16")
        return out


def conv_up2x_prepare(wt: torch.Tensor) -> torch.Tensor:
    """3x3 weights [Cout, 9*Cin] ((ky,kx,ci) order) -> the four 2x2 phase kernels [4, Cout, 4*Cin] of conv_up2x (once per
    layer: sums of the taps that read the same low-resolution pixel, fp32, rounded to bf16 once)."""
    lib = _l.load()
    h16 = wt.dtype == F16  # precision "fp16": fp16 taps in, fp16 phase kernels out
    _req(wt, "wt", F16 if h16 else BF16)
    assert wt.is_contiguous() and wt.shape[1] % 9 == 0
    Cout, Cin = wt.shape[0], wt.shape[1] // 9
    wp = torch.empty((4, Cout, 4 * Cin), dtype=wt.dtype, device=wt.device)
    fn = lib.dm4d_conv_up2x_prepare_f16 if h16 else lib.dm4d_conv_up2x_prepare_bf16
    _l.check(fn(_stream(), _p(wt), _p(wp), Cout, Cin), "dm4d_conv_up2x_prepare")
    return wp


def conv_up2x_supported(Cin: int, Cout: int) -> bool:
    return Cin % 64 == 0 and Cout % 8 == 0


class Upsampler:
    """Upsample2D of the UNet / VAE decoder: nearest x2 + 3x3 convolution.  Runs as four 2x2 phase convolutions of the
    low-resolution input (conv_up2x) when the channel counts allow, else as the gather kernel that reads the upsampled
    image through index arithmetic.  The phase kernels are made from the checkpoint's 3x3 weights HERE, at load time, and the
    stream is drained before the object is handed out: the runner's task streams (one worker thread and HIP stream each,
    sharing one pipeline) must never see a published `wp` whose prepare kernel is still queued on another stream."""

    def __init__(self, wt: torch.Tensor, bias: Optional[torch.Tensor], parity: bool = False, h16: bool = False):
        """parity (precision "parity"): wt is duplicated along Cin, the input fp32; the phase kernels -- whose weights are SUMS of taps
        rounded to bf16 once more, a deviation from the checkpoint's arithmetic -- are not used: the gather kernel reads the upsampled
        image of the two-term operand through index arithmetic and multiplies by the checkpoint's own weights."""
        # h16 (precision "fp16"; implies `parity` = fp32 tensors): fp16 weights, one fp16 operand plane; the phase kernels ARE used (their
        # extra weight rounding is 2^-12 relative in fp16, inside the precision's budget -- in bf16 it would be 2^-9)
        self.wt, self.bias, self.wp, self.parity, self.h16 = wt, bias, None, parity or h16, h16
        self.phase = (h16 or not self.parity) and conv_up2x_supported(wt.shape[1] // 9, wt.shape[0]) and UP2X_PHASE
        if self.phase and wt.is_cuda:
            with torch.cuda.device(wt.device):
                self.wp = conv_up2x_prepare(wt)
                torch.cuda.current_stream(wt.device).synchronize()

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.h16 and self.phase and x.numel() * 2 < (1 << 32):
            if self.wp is None or self.wp.device != x.device:
                raise _l.Dm4dError("Upsampler: phase kernels were not prepared on the input's device")
            return conv_up2x(split(x, h16=True), self.wp, bias=self.bias, out_f32=True)
        if self.parity:
            return conv3x3(split(x, h16=self.h16), self.wt, bias=self.bias, upsample=True, out_f32=True)
        # the phase kernel addresses its input with 32-bit byte offsets: inputs of 4 GiB or more take the gather kernel
        if not self.phase or x.numel() * 2 >= (1 << 32):
            return conv3x3(x, self.wt, bias=self.bias, upsample=True)
        if self.wp is None or self.wp.device != x.device:
            raise _l.Dm4dError(f"Upsampler: phase kernels live on {None if self.wp is None else self.wp.device}, input on {x.device} "
                            "(reload the pipeline on the new device: host/loader.py does)")
        return conv_up2x(x, self.wp, bias=self.bias)


def conv_up2x(x: torch.Tensor, wp: torch.Tensor, *, bias=None, out_f32: bool = False) -> torch.Tensor:
    """conv3x3(nearest_upsample_x2(x)) from the phase kernels of conv_up2x_prepare: x [B,H,W,Cin] -> [B,2H,2W,Cout].
    fp16 x / wp / bias (precision "fp16"): fp32 (out_f32) or fp16 result."""
    lib = _l.load()
    h16 = wp.dtype == F16
    dt = F16 if h16 else BF16
    _req(x, "x", dt), _req(wp, "wp", dt)
    assert x.is_contiguous() and wp.is_contiguous() and (h16 or not out_f32)
    B, H, W, Cin = x.shape
    Cout = wp.shape[1]
    assert wp.shape == (4, Cout, 4 * Cin), (wp.shape, Cin)
    y = torch.empty((B, 2 * H, 2 * W, Cout), dtype=F32 if out_f32 else dt, device=x.device)
    # credited with the multiply-adds it executes (4 taps per output pixel), not the 9 of the op it replaces
    with _Prof("conv3x3", 2.0 * B * 4 * H * W * 4 * Cin * Cout, "flop", B * 4 * H * W):
        if h16:
            rc = lib.dm4d_conv_up2x_nhwc_f16(_stream(), _p(x), B, H, W, Cin, _p(wp), _p(y), Cout, _p(bias), _l.EPI_F32OUT if out_f32 else 0)
        else:
            rc = lib.dm4d_conv_up2x_nhwc_bf16(_stream(), _p(x), B, H, W, Cin, _p(wp), _p(y), Cout, _p(bias))
    _l.check(rc, "dm4d_conv_up2x_nhwc")
    if not h16:
        _trace("conv_up2x", y, x=x, wp=wp, bias=bias)
    return y


def conv2d_direct(x: torch.Tensor, wt: torch.Tensor, *, ksize: int, bias=None, stride: int = 1, pad: int = 1,
                  silu: bool = False) -> torch.Tensor:
    """Thin-layer convolution (PoseEncoder): x [B,H,W,Cin] NHWC, wt [Cout, ksize*ksize*Cin] ((ky,kx,ci) order).  bf16 tensors, or
    (parity precision) x, wt, bias and the result all fp32."""
    lib = _l.load()
    dt = F32 if x.dtype == F32 else BF16
    _req(x, "x", dt), _req(wt, "wt", dt)
    if bias is not None:
        _req(bias, "bias", dt)
    assert x.is_contiguous() and wt.is_contiguous()
    B, H, W, Cin = x.shape
    Cout = wt.shape[0]
    assert wt.shape[1] == ksize * ksize * Cin, (wt.shape, ksize, Cin)
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    y = torch.empty((B, Ho, Wo, Cout), dtype=dt, device=x.device)
    fn = lib.dm4d_conv2d_direct_nhwc_f32 if dt == F32 else lib.dm4d_conv2d_direct_nhwc_bf16
    rc = fn(_stream(), _p(x), B, H, W, Cin, _p(wt), _p(bias), _p(y), Ho, Wo, Cout, ksize, stride, pad, 1 if silu else 0)
    _l.check(rc, "dm4d_conv2d_direct_nhwc")
    return y


def groupnorm(x1: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, *,
              x2: Optional[torch.Tensor] = None, silu: bool = False, raw_out: bool = False):
    """GroupNorm(+SiLU) over the channel concat [x1 | x2]; x [B, HW, C] (any leading spatial shape).
    fp32 input (wide precisions): the result leaves as an operand -- parity: fp64 statistics, two-term [..., 2 C]; fp16 (fp16 gamma / beta):
    one fp16 plane.  raw_out (fp16 precision): also returns fp16 of the un-normalised concat [x1 | x2] (the shortcut convolution's operand)
    -> (y, raw)."""
    lib = _l.load()
    if isinstance(x1, torch.Tensor) and x1.dtype == F32:
        return _groupnorm_f32(x1, gamma, beta, groups, eps, x2, silu, raw_out)
    assert not raw_out
    if isinstance(x1, torch.Tensor) and x1.dtype == F16:
        return _groupnorm_f16(x1, gamma, beta, groups, eps, x2, silu)
    _req(x1, "x1"), _req(gamma, "gamma"), _req(beta, "beta")
    assert x1.is_contiguous()
    B, C1 = x1.shape[0], x1.shape[-1]
    HW = x1.numel() // (B * C1)
    C2 = 0
    if x2 is not None:
        _req(x2, "x2")
        assert x2.is_contiguous() and x2.shape[0] == B
        C2 = x2.shape[-1]
    y = torch.empty(x1.shape[:-1] + (C1 + C2,), dtype=BF16, device=x1.device)
    ws = torch.empty(lib.dm4d_groupnorm_ws_bytes(B, HW, groups) // 4, dtype=torch.float32, device=x1.device)
    with _Prof("groupnorm", 2.0 * y.numel() * 2, "byte", B * HW):  # ALGORITHMIC bytes: read once + write once; the
        # statistics pass re-reads the input (mostly from L2 / Infinity Cache), so the device moves up to 1.5x this
        rc = lib.dm4d_groupnorm_nhwc_bf16(_stream(), _p(x1), C1, _p(x2), C2, B, HW, groups, eps, _p(gamma), _p(beta),
                                          _p(y), 1 if silu else 0, _p(ws))
    _l.check(rc, "dm4d_groupnorm_nhwc_bf16")
    _trace("groupnorm", y, x1=x1, x2=x2, gamma=gamma, beta=beta, groups=groups, eps=eps, silu=silu)
    return y


def _groupnorm_f32(x1, gamma, beta, groups, eps, x2, silu, raw_out=False):
    lib = _l.load()
    h16 = gamma.dtype == F16  # precision "fp16": fp16 parameters, one fp16 plane out
    pdt = F16 if h16 else BF16
    _req(x1, "x1", F32), _req(gamma, "gamma", pdt), _req(beta, "beta", pdt)
    assert x1.is_contiguous()
    B, C1 = x1.shape[0], x1.shape[-1]
    HW = x1.numel() // (B * C1)
    C2 = 0
    if x2 is not None:
        _req(x2, "x2", F32)
        assert x2.is_contiguous() and x2.shape[0] == B
        C2 = x2.shape[-1]
    y = torch.empty(x1.shape[:-1] + ((1 if h16 else 2) * (C1 + C2),), dtype=pdt, device=x1.device)
    ws = torch.empty(lib.dm4d_groupnorm_f32_ws_bytes(B, HW, groups) // 8, dtype=torch.float64, device=x1.device)
    bpe = 6.0 if h16 else 8.0  # algorithmic bytes per element: fp32 in + the operand out (one fp16 plane, or hi + lo)
    if raw_out:
        if not h16:
            raise _l.Dm4dError("groupnorm: raw_out is a feature of the fp16 precision")
        fused = C1 % 8 == 0 and C2 % 8 == 0 and all(t is None or t.data_ptr() % 16 == 0 for t in (x1, x2))
        if not fused:  # shapes the vectorised kernels do not take: the operand from its own pass
            M = B * HW
            return (_groupnorm_f32(x1, gamma, beta, groups, eps, x2, silu),
                    split(x1.reshape(M, C1), x2.reshape(M, C2) if x2 is not None else None, h16=True).view(y.shape))
        raw = torch.empty_like(y)
        with _Prof("groupnorm", (bpe + 2.0) * (x1.numel() + (x2.numel() if x2 is not None else 0)), "byte", B * HW):
            rc = lib.dm4d_groupnorm_nhwc_f32_f16_raw(_stream(), _p(x1), C1, _p(x2), C2, B, HW, groups, eps, _p(gamma), _p(beta), _p(y), _p(raw),
                                                     1 if silu else 0, _p(ws))
        _l.check(rc, "dm4d_groupnorm_nhwc_f32_f16_raw")
        return y, raw
    with _Prof("groupnorm", bpe * x1.numel() + (bpe * x2.numel() if x2 is not None else 0.0), "byte", B * HW):
        fn = lib.dm4d_groupnorm_nhwc_f32_f16 if h16 else lib.dm4d_groupnorm_nhwc_f32_split
        rc = fn(_stream(), _p(x1), C1, _p(x2), C2, B, HW, groups, eps, _p(gamma), _p(beta), _p(y), 1 if silu else 0, _p(ws))
    _l.check(rc, "dm4d_groupnorm_nhwc_f32_split")
    return y


def _groupnorm_f16(x1, gamma, beta, groups, eps, x2, silu):
    """precision "fp16", fp16 input (an activation its producer left in fp16: conv1's output in front of norm2) -> one fp16 plane."""
    lib = _l.load()
    _req(x1, "x1", F16), _req(gamma, "gamma", F16), _req(beta, "beta", F16)
    assert x1.is_contiguous()
    B, C1 = x1.shape[0], x1.shape[-1]
    HW = x1.numel() // (B * C1)
    C2 = 0
    if x2 is not None:
        _req(x2, "x2", F16)
        assert x2.is_contiguous() and x2.shape[0] == B
        C2 = x2.shape[-1]
    y = torch.empty(x1.shape[:-1] + (C1 + C2,), dtype=F16, device=x1.device)
    ws = torch.empty(lib.dm4d_groupnorm_ws_bytes(B, HW, groups) // 4, dtype=torch.float32, device=x1.device)
    with _Prof("groupnorm", 2.0 * y.numel() * 2, "byte", B * HW):
        rc = lib.dm4d_groupnorm_nhwc_f16_f16(_stream(), _p(x1), C1, _p(x2), C2, B, HW, groups, eps, _p(gamma), _p(beta), _p(y),
                                             1 if silu else 0, _p(ws))
    _l.check(rc, "dm4d_groupnorm_nhwc_f16_f16")
    return y


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """fp32 input (parity precision): the result leaves as a two-term operand [..., 2 C]."""
    lib = _l.load()
    if isinstance(x, torch.Tensor) and x.dtype == F32:
        h16 = gamma.dtype == F16  # precision "fp16": fp16 parameters, one fp16 plane out
        pdt, planes = (F16, 1) if h16 else (BF16, 2)
        _req(x, "x", F32), _req(gamma, "gamma", pdt), _req(beta, "beta", pdt)
        x2 = x.reshape(-1, x.shape[-1])
        C = x2.shape[1]
        y = torch.empty((x2.shape[0], planes * C), dtype=pdt, device=x.device)
        with _Prof("layernorm", (4.0 + 2.0 * planes) * x2.numel(), "byte", x2.shape[0]):
            fn = lib.dm4d_layernorm_f32_f16 if h16 else lib.dm4d_layernorm_f32_split
            rc = fn(_stream(), _p(x2), x2.stride(0), _p(gamma), _p(beta), _p(y), y.stride(0), x2.shape[0], C, eps)
        _l.check(rc, "dm4d_layernorm_f32_split")
        return y.view(x.shape[:-1] + (planes * C,))
    _req(x, "x"), _req(gamma, "gamma"), _req(beta, "beta")
    x2 = x.reshape(-1, x.shape[-1])
    y = torch.empty_like(x2)
    with _Prof("layernorm", 2.0 * y.numel() * 2, "byte", x2.shape[0]):
        rc = lib.dm4d_layernorm_bf16(_stream(), _p(x2), x2.stride(0), _p(gamma), _p(beta), _p(y), y.stride(0), x2.shape[0],
                                     x2.shape[1], eps)
    _l.check(rc, "dm4d_layernorm_bf16")
    _trace("layernorm", y, x=x2, gamma=gamma, beta=beta, eps=eps)
    return y.view(x.shape)


LOG2E = 1.4426950408889634


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, heads: int, seq: int,
              scale: Optional[float] = None, out: Optional[torch.Tensor] = None,
              kv_seq: Optional[int] = None, q_scaled: bool = False) -> torch.Tensor:
    """q/k/v: [batch*seq, >=heads*64] row-strided views (e.g. column slices of the fused QKV output).
    kv_seq: keys per batch when K/V hold more tokens than Q (frame-sharded 3-D attention); default = seq.
    q_scaled: q already carries scale * LOG2E (folded into the to_q weights, unet._TransformerBlock)."""
    lib = _l.load()
    h16 = isinstance(q, torch.Tensor) and q.dtype == F16  # precision "fp16": fp16 Q (pre-scaled) / K / V -> fp16 O
    dt = F16 if h16 else BF16
    _req(q, "q", dt), _req(k, "k", dt), _req(v, "v", dt)
    assert q.shape[0] == batch * seq and q.shape[1] == heads * 64, (q.shape, batch, seq, heads)
    kv_seq = seq if kv_seq is None else kv_seq
    assert k.shape[0] == batch * kv_seq and v.shape[0] == batch * kv_seq, (k.shape, batch, kv_seq)
    if out is None:
        out = torch.empty((batch * seq, heads * 64), dtype=dt, device=q.device)
    if h16 and not q_scaled:
        raise _l.Dm4dError("attention: fp16 operands need a pre-scaled Q (gemm(..., scale_cols=C, col_scale=scale * LOG2E))")
    if scale is None:
        scale = 0.125
    prof = KERNEL_TIMER
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    with _Prof("attention", 4.0 * batch * heads * seq * kv_seq * 64, "flop", batch * seq):
        if h16:
            rc = lib.dm4d_attention_qscaled_kv_f16(_stream(), _p(q), _p(k), _p(v), _p(out), q.stride(0), k.stride(0),
                                                   v.stride(0), out.stride(0), batch, heads, seq, kv_seq)
        elif q_scaled:
            rc = lib.dm4d_attention_qscaled_kv_bf16(_stream(), _p(q), _p(k), _p(v), _p(out), q.stride(0), k.stride(0),
                                                    v.stride(0), out.stride(0), batch, heads, seq, kv_seq)
        else:
            rc = lib.dm4d_attention_kv_bf16(_stream(), _p(q), _p(k), _p(v), _p(out), q.stride(0), k.stride(0),
                                            v.stride(0), out.stride(0), batch, heads, seq, kv_seq, scale)
    if prof is not None:
        e1.record()
        prof.append(("attn_kernel", 4.0 * batch * heads * seq * kv_seq * 64, e0, e1))
    _l.check(rc, "dm4d_attention_kv_bf16")
    if h16:
        return out
    _trace("attention", out, q=q, k=k, v=v, batch=batch, heads=heads, seq=seq, kv_seq=kv_seq, scale=scale, q_scaled=q_scaled)
    return out


def attention_split(qkv: Optional[torch.Tensor], batch: int, heads: int, seq: int, scale: Optional[float] = None, *,
                    q: Optional[torch.Tensor] = None, kv: Optional[torch.Tensor] = None, kv_seq: Optional[int] = None) -> torch.Tensor:
    """Parity precision: qkv [batch*seq, 6 C] = the planes gemm(split_out=True) leaves for a fused QKV projection
    ([q_hi | k_hi | v_hi | q_lo | k_lo | v_lo], C = heads * 64) -> attention output as a two-term operand [batch*seq, 2 C].
    Frame-sharded form (qkv = None): q [batch*seq, 2 C] = [q_hi | q_lo] of the local queries, kv [batch*kv_seq, 4 C] =
    [k_hi | v_hi | k_lo | v_lo] of the all-gathered keys (the planes of gemm(n, w_kv, split_out=True)).
    Three MFMAs per product, exact running-max softmax in fp32 (dm4d_attention_split_bf16)."""
    lib = _l.load()
    C = heads * 64
    if qkv is not None:
        _req(qkv, "qkv")
        assert qkv.shape == (batch * seq, 6 * C), (qkv.shape, batch, seq, heads)
        qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C
        ldq = ldk = ldv = qkv.stride(0)
        q_lo = k_lo = v_lo = 3 * C
        kv_seq, dev = seq, qkv.device
    else:
        _req(q, "q"), _req(kv, "kv")
        kv_seq = seq if kv_seq is None else kv_seq
        assert q.shape == (batch * seq, 2 * C) and kv.shape == (batch * kv_seq, 4 * C), (q.shape, kv.shape, batch, seq, kv_seq, heads)
        qp, kp, vp = q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 2 * C
        ldq, ldk, ldv = q.stride(0), kv.stride(0), kv.stride(0)
        q_lo, k_lo, v_lo = C, 2 * C, 2 * C
        dev = q.device
    out = torch.empty((batch * seq, 2 * C), dtype=BF16, device=dev)
    with _Prof("attention", 3 * 4.0 * batch * heads * seq * kv_seq * 64, "flop", batch * seq):  # three MFMA terms per product
        rc = lib.dm4d_attention_split_bf16(_stream(), qp, kp, vp, _p(out), ldq, ldk, ldv, out.stride(0), q_lo, k_lo, v_lo, C, batch,
                                           heads, seq, kv_seq, 0.125 if scale is None else scale)
    _l.check(rc, "dm4d_attention_split_bf16")
    return out


def softmax_rows_split(s: torch.Tensor, scale: float, n: Optional[int] = None, h16: bool = False) -> torch.Tensor:
    """Wide precisions: P = softmax(s * scale) over the first n columns of fp32 logits s [M, Np] -> parity: three planes
    [p_hi | p_lo | p_hi], bf16 [M, 3 Np]; h16 (precision "fp16"): one fp16 plane [M, Np] (columns n .. Np-1 of every plane zero)."""
    lib = _l.load()
    _req(s, "s", F32)
    M, Np = s.shape
    n = Np if n is None else int(n)
    if h16:
        p = torch.empty((M, Np), dtype=F16, device=s.device)
        _l.check(lib.dm4d_softmax_rows_f32_f16(_stream(), _p(s), s.stride(0), _p(p), p.stride(0), M, n, Np, scale),
                 "dm4d_softmax_rows_f32_f16")
        return p
    p = torch.empty((M, 3 * Np), dtype=BF16, device=s.device)
    _l.check(lib.dm4d_softmax_rows_f32_split(_stream(), _p(s), s.stride(0), _p(p), p.stride(0), M, n, Np, scale),
             "dm4d_softmax_rows_f32_split")
    return p


# bench.py sets this to a list to time individual launches with HIP events on the launch stream:
# entries are (kernel, algorithmic flops, start_event, end_event).  None = no instrumentation.
KERNEL_TIMER = None
# Same idea for every kernel family (bench.py's untimed breakdown pass): entries are (family, work, unit, start, end, rows)
# with unit "flop" or "byte" and rows = output rows of the launch.  None = no instrumentation.
PROFILE = None


class _Prof:
    """with _Prof(family, work, unit): <one launch>   -- records two events when ops.PROFILE is a list."""

    def __init__(self, family: str, work: float, unit: str, rows: int = 0):
        self.args = (family, work, unit)
        self.rows = int(rows)  # output rows (tokens / pixels) of the launch: bench.py derives the UNet level from it
        self.on = PROFILE is not None

    def __enter__(self):
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            PROFILE.append(self.args + (self.e0, self.e1, self.rows))
        return False


def softmax_rows(s: torch.Tensor, scale: float, n: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """P = softmax(s * scale) per row, bf16; s bf16 or fp32 (logits from gemm(out_f32=True)).
    n (fp32 logits only): softmax over the first n columns of every row -- a key axis padded to the GEMM's K granularity; `out`
    (same shape as s) then keeps whatever it holds behind column n (the caller's zeros)."""
    lib = _l.load()
    f32 = s.dtype == torch.float32
    _req(s, "s", torch.float32 if f32 else BF16)
    if n is not None and not f32:
        raise _l.Dm4dError("softmax_rows: a column count below the row length needs fp32 logits")
    p = torch.empty(s.shape, dtype=BF16, device=s.device) if out is None else out
    _req(p, "out", BF16)
    if p.shape != s.shape:
        raise _l.Dm4dError(f"softmax_rows: out has shape {tuple(p.shape)}, the logits {tuple(s.shape)}")
    fn = lib.dm4d_softmax_rows_f32in_bf16 if f32 else lib.dm4d_softmax_rows_bf16
    rc = fn(_stream(), _p(s), s.stride(0), _p(p), p.stride(0), s.shape[0], s.shape[1] if n is None else int(n), scale)
    _l.check(rc, "dm4d_softmax_rows_f32in_bf16" if f32 else "dm4d_softmax_rows_bf16")
    _trace("softmax_rows", p, s=s, scale=scale, n=n)
    return p


def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos: bool = True, freq_shift: float = 0.0,
                       out_f32: bool = False) -> torch.Tensor:
    lib = _l.load()
    _req(t, "t", torch.float32)
    out = torch.empty((t.shape[0], dim), dtype=F32 if out_f32 else BF16, device=t.device)
    fn = lib.dm4d_timestep_embedding_f32 if out_f32 else lib.dm4d_timestep_embedding_bf16
    rc = fn(_stream(), _p(t), _p(out), t.shape[0], dim, 1 if flip_sin_to_cos else 0, freq_shift)
    _l.check(rc, "dm4d_timestep_embedding")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    lib = _l.load()
    _req(x, "x")
    assert x.is_contiguous()
    y = torch.empty_like(x)
    _l.check(lib.dm4d_silu_bf16(_stream(), _p(x), _p(y), x.numel()), "dm4d_silu_bf16")
    _trace("silu", y, x=x)
    return y


def pack_model_input(latents, pv_lat, plucker, skel, mask, is_cond, cpad: int, use_cfg: bool,
                     frame_idx: Optional[torch.Tensor] = None, h16: bool = False) -> torch.Tensor:
    """Inputs NHWC [N, HW, c] (task level); is_cond int32 [F]; frame_idx int32 [F] selects the window's frames
    (None: F = N, identity).  Mutates the cond rows of `latents` (reference aliasing).
    fp32 task tensors (wide precisions): the result is conv_in's operand -- [hi | lo] bf16, or (h16) one fp16 plane."""
    lib = _l.load()
    dt = F32 if latents.dtype == F32 else BF16  # fp32 task tensors = parity precision: the result is conv_in's two-term operand
    for t, n in ((latents, "latents"), (pv_lat, "pv_lat"), (plucker, "plucker"), (mask, "mask")):
        _req(t, n, dt)
        assert t.is_contiguous()
    if skel is not None:
        _req(skel, "skel", dt)
    _req(is_cond, "is_cond", torch.int32)
    F, HW = is_cond.shape[0], latents.shape[1]
    if frame_idx is not None:
        _req(frame_idx, "frame_idx", torch.int32)
        assert frame_idx.shape[0] == F
    else:
        assert latents.shape[0] == F
    assert not h16 or dt == F32
    out = torch.empty(((2 if use_cfg else 1) * F, HW, 2 * cpad if (dt == F32 and not h16) else cpad), dtype=F16 if h16 else BF16,
                      device=latents.device)
    fn = lib.dm4d_pack_model_input_f32_f16 if h16 else (lib.dm4d_pack_model_input_f32_split if dt == F32 else lib.dm4d_pack_model_input_bf16)
    rc = fn(_stream(), _p(latents), _p(pv_lat), _p(plucker), _p(skel), _p(mask), _p(is_cond), _p(frame_idx), _p(out), F, HW, cpad,
            1 if use_cfg else 0)
    _l.check(rc, "dm4d_pack_model_input")
    return out


def cfg_ddim_step(latents, noise_pred, coef, is_cond, use_cfg: bool, guidance_scale: float, v_prediction: bool,
                  frame_idx: Optional[torch.Tensor] = None):
    """In-place DDIM update of rows frame_idx of `latents` [N,HW,4] from noise_pred [cfg*F, HW, ldn]."""
    lib = _l.load()
    dt = F32 if latents.dtype == F32 else BF16
    _req(latents, "latents", dt), _req(noise_pred, "noise_pred", dt), _req(coef, "coef", torch.float32)
    _req(is_cond, "is_cond", torch.int32)
    F, HW = is_cond.shape[0], latents.shape[1]
    if frame_idx is not None:
        _req(frame_idx, "frame_idx", torch.int32)
    rc = (lib.dm4d_cfg_ddim_step_f32 if dt == F32 else lib.dm4d_cfg_ddim_step_bf16)(_stream(), _p(latents), _p(noise_pred), noise_pred.stride(-2), _p(coef),
                                     _p(is_cond), _p(frame_idx), F, HW, 1 if use_cfg else 0, guidance_scale,
                                     1 if v_prediction else 0)
    _l.check(rc, "dm4d_cfg_ddim_step_bf16")
    return latents


def cfg_linear_step(latents, x0_prev, noise_pred, coef, is_cond, use_cfg: bool, guidance_scale: float,
                    frame_idx: Optional[torch.Tensor] = None):
    """In-place linear multistep update (x' = a x + b m + c p; p' = d x + e m) of rows frame_idx of `latents` and `x0_prev`
    [N,HW,4] from noise_pred [cfg*F, HW, ldn]; coef [F,8] fp32 rows from scheduler.step_rows."""
    lib = _l.load()
    dt = F32 if latents.dtype == F32 else BF16
    _req(latents, "latents", dt), _req(x0_prev, "x0_prev", dt), _req(noise_pred, "noise_pred", dt), _req(coef, "coef", torch.float32)
    _req(is_cond, "is_cond", torch.int32)
    assert x0_prev.shape == latents.shape and coef.shape[-1] == 8 and coef.is_contiguous()
    F, HW = is_cond.shape[0], latents.shape[1]
    if frame_idx is not None:
        _req(frame_idx, "frame_idx", torch.int32)
    rc = (lib.dm4d_cfg_linear_step_f32 if dt == F32 else lib.dm4d_cfg_linear_step_bf16)(_stream(), _p(latents), _p(x0_prev), _p(noise_pred), noise_pred.stride(-2), _p(coef),
                                       _p(is_cond), _p(frame_idx), F, HW, 1 if use_cfg else 0, guidance_scale)
    _l.check(rc, "dm4d_cfg_linear_step_bf16")
    return latents


def cfg_multistep_step(latents, states, noise_pred, coef, is_cond, use_cfg: bool, guidance_scale: float,
                       frame_idx: Optional[torch.Tensor] = None):
    """In-place general linear multistep update (UniPC with its corrector, DEIS; dm4d.h: dm4d_cfg_multistep_step_bf16) of rows frame_idx
    of `latents` and of the stored tensors `states` (a list of 1-3 tensors shaped like latents, zero at the start of a call) from
    noise_pred [cfg*F, HW, ldn]; coef [F,16] fp32 rows from scheduler.step_rows."""
    lib = _l.load()
    dt = F32 if latents.dtype == F32 else BF16
    _req(latents, "latents", dt), _req(noise_pred, "noise_pred", dt), _req(coef, "coef", torch.float32), _req(is_cond, "is_cond", torch.int32)
    assert 1 <= len(states) <= 3 and coef.shape[-1] == 16 and coef.is_contiguous()
    for t in states:
        _req(t, "state", dt)
        assert t.shape == latents.shape and t.is_contiguous()
    F, HW = is_cond.shape[0], latents.shape[1]
    if frame_idx is not None:
        _req(frame_idx, "frame_idx", torch.int32)
    st = list(states) + [None] * (3 - len(states))
    fn = lib.dm4d_cfg_multistep_step_f32 if dt == F32 else lib.dm4d_cfg_multistep_step_bf16
    rc = fn(_stream(), _p(latents), _p(st[0]), _p(st[1]), _p(st[2]), _p(noise_pred), noise_pred.stride(-2), _p(coef), _p(is_cond),
            _p(frame_idx), F, HW, 1 if use_cfg else 0, guidance_scale)
    _l.check(rc, "dm4d_cfg_multistep_step")
    return latents


def nchw_to_nhwc(x: torch.Tensor, cpad: Optional[int] = None) -> torch.Tensor:
    lib = _l.load()
    _req(x, "x")
    assert x.is_contiguous()
    B, C, H, W = x.shape
    cpad = cpad or C
    y = torch.empty((B, H, W, cpad), dtype=BF16, device=x.device)
    _l.check(lib.dm4d_nchw_to_nhwc_bf16(_stream(), _p(x), _p(y), B, C, H * W, cpad), "dm4d_nchw_to_nhwc_bf16")
    return y


def nhwc_to_nchw(x: torch.Tensor, C: Optional[int] = None) -> torch.Tensor:
    lib = _l.load()
    dt = F32 if x.dtype == F32 else BF16
    _req(x, "x", dt)
    assert x.is_contiguous()
    B, H, W, ld = x.shape
    C = C or ld
    y = torch.empty((B, C, H, W), dtype=dt, device=x.device)
    fn = lib.dm4d_nhwc_to_nchw_f32 if dt == F32 else lib.dm4d_nhwc_to_nchw_bf16
    _l.check(fn(_stream(), _p(x), _p(y), B, C, H * W, ld), "dm4d_nhwc_to_nchw")
    return y


def vae_sample(moments: torch.Tensor, noise: torch.Tensor, channels: int, scale: float) -> torch.Tensor:
    """moments [..., >=2C] (mean | logvar), noise [..., C] -> (mean + std * noise) * scale, [..., C]."""
    lib = _l.load()
    dt = F32 if moments.dtype == F32 else BF16
    _req(moments, "moments", dt), _req(noise, "noise", dt)
    assert noise.is_contiguous() and noise.shape[-1] == channels
    M = noise.numel() // channels
    out = torch.empty_like(noise)
    rc = (lib.dm4d_vae_sample_f32 if dt == F32 else lib.dm4d_vae_sample_bf16)(_stream(), _p(moments), moments.stride(-2), _p(noise), _p(out), M, channels, scale)
    _l.check(rc, "dm4d_vae_sample_bf16")
    return out


def scale_pad(x: torch.Tensor, cpad: int, scale: float) -> torch.Tensor:
    lib = _l.load()
    _req(x, "x")
    C = x.shape[-1]
    M = x.numel() // C
    assert x.is_contiguous()
    y = torch.empty(x.shape[:-1] + (cpad,), dtype=BF16, device=x.device)
    _l.check(lib.dm4d_scale_pad_bf16(_stream(), _p(x), C, _p(y), cpad, M, C, scale), "dm4d_scale_pad_bf16")
    return y


def resize_to_nhwc(x: torch.Tensor, size: Tuple[int, int], mode: str, out_f32: bool = False) -> torch.Tensor:
    """x fp32 NCHW on the device -> bf16 (out_f32: fp32) NHWC [B,h,w,C]; mode 'bilinear' | 'nearest' (F.interpolate semantics)."""
    lib = _l.load()
    _req(x, "x", torch.float32)
    assert x.is_contiguous() and mode in ("bilinear", "nearest")
    B, C, H, W = x.shape
    h, w = size
    y = torch.empty((B, h, w, C), dtype=F32 if out_f32 else BF16, device=x.device)
    rc = (lib.dm4d_resize_nchw_f32_to_nhwc_f32 if out_f32 else lib.dm4d_resize_nchw_f32_to_nhwc_bf16)(_stream(), _p(x), _p(y), B, C, H, W, h, w, 1 if mode == "bilinear" else 0)
    _l.check(rc, "dm4d_resize_nchw_f32_to_nhwc_bf16")
    return y


def resize_aa(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """F.interpolate(x, size=size, mode="bilinear", antialias=True) for fp32 NCHW on the device (the result writer's mosaic)."""
    lib = _l.load()
    _req(x, "x", F32)
    assert x.is_contiguous() and x.ndim == 4
    N, C, H, W = x.shape
    h, w = size
    y = torch.empty((N, C, h, w), dtype=F32, device=x.device)
    _l.check(lib.dm4d_resize_aa_nchw_f32(_stream(), _p(x), _p(y), N * C, H, W, h, w), "dm4d_resize_aa_nchw_f32")
    return y


def camera_rows(Ks: torch.Tensor, poses: torch.Tensor) -> torch.Tensor:
    """Per-frame camera constants of dm4d_plucker_latent_bf16, on the host in fp32 as the reference prepares them
    (ray_utils.py:56-59,101-105): [K^-1 | R | T | -R^T T] with [R | T] = inverse(pose)[:3].  -> [N, 24] fp32 (CPU)."""
    Ks, poses = Ks.detach().float().cpu(), poses.detach().float().cpu()
    ext = torch.inverse(poses)
    R, T = ext[:, :3, :3], ext[:, :3, 3:]
    o = -R.mT @ T
    return torch.cat([torch.inverse(Ks).reshape(-1, 9), R.reshape(-1, 9), T.reshape(-1, 3), o.reshape(-1, 3)], dim=1).contiguous()


def plucker_latents(Ks: torch.Tensor, poses: torch.Tensor, image_size: Tuple[int, int], latent_size: Tuple[int, int],
                    device, out_f32: bool = False) -> torch.Tensor:
    """Pluecker maps at latent resolution from the cameras -> bf16 (out_f32: fp32) NHWC [N, h, w, 6] on `device` (see dm4d.h)."""
    lib = _l.load()
    cams = camera_rows(Ks, poses).to(device)
    (H, W), (h, w) = image_size, latent_size
    y = torch.empty((cams.shape[0], h, w, 6), dtype=F32 if out_f32 else BF16, device=device)
    rc = (lib.dm4d_plucker_latent_f32 if out_f32 else lib.dm4d_plucker_latent_bf16)(_stream(), _p(cams), _p(y), cams.shape[0], H, W, h, w)
    _l.check(rc, "dm4d_plucker_latent_bf16")
    return y


def postprocess_images(x: torch.Tensor, channels: int = 3) -> torch.Tensor:
    """NHWC [B,H,W,ld] -> NCHW [B,channels,H,W] with (x/2+0.5).clamp(0,1)."""
    lib = _l.load()
    dt = F32 if x.dtype == F32 else BF16
    _req(x, "x", dt)
    assert x.is_contiguous()
    B, H, W, ld = x.shape
    y = torch.empty((B, channels, H, W), dtype=dt, device=x.device)
    fn = lib.dm4d_postprocess_images_f32 if dt == F32 else lib.dm4d_postprocess_images_bf16
    _l.check(fn(_stream(), _p(x), _p(y), B, channels, H * W, ld), "dm4d_postprocess_images")
    return y
