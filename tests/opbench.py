"""Per-kernel micro-benchmarks at the real UNet shapes (72x40 latents, F=16/24, CFG): prints achieved
TFLOP/s (MFMA kernels) or GB/s (HBM-bound kernels).  Run on a GPU box: `python tests/opbench.py`."""
from __future__ import annotations

import math
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffuman4d_amd.host import ops  # noqa: E402

BF = torch.bfloat16
DEV = "cuda"


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3  # seconds


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(BF)


def bench_gemm(M, N, K, geglu=False, residual=True, tag=""):
    a, w = rnd(M, K), rnd(2 * N if geglu else N, K, scale=1 / math.sqrt(K))
    b = rnd(2 * N if geglu else N)
    res = rnd(M, N) if residual else None
    t = timeit(lambda: ops.gemm(a, w, bias=b, residual=res, geglu=geglu))
    fl = 2.0 * M * K * (2 * N if geglu else N)
    print(f"gemm{tag:10s} M={M:6d} N={N:5d} K={K:5d} geglu={int(geglu)}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)


def bench_ff(M, C=320, hidden=1280, tag=""):
    """Level-0 feed-forward: the fused launch vs gemm(GEGLU) + gemm(residual) (+ the LayerNorm both forms follow)."""
    n, x = rnd(M, C), rnd(M, C)
    ff = ops.FeedForward(rnd(2 * hidden, C, scale=1 / math.sqrt(C)), rnd(2 * hidden), rnd(C, hidden, scale=1 / math.sqrt(hidden)), rnd(C))
    fl = 2.0 * M * 3 * hidden * C
    lnp = (rnd(C), rnd(C), 1e-5)  # norm3 in front of the feed-forward: folded into the fused launch, its own launch otherwise
    for rep in range(2):  # alternate the forms: whatever is timed first runs on a colder chip
        for fused in (True, False):
            ops.FF_FUSED = fused
            t = timeit(lambda: ff(n, x, ln=lnp))
            print(f"ff{tag:8s} M={M:6d} C={C} hidden={hidden} {'fused+ln   ' if fused else 'ln+two-gemm'} round {rep}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
    ops.FF_FUSED = True


def bench_ff_proj(M, C=320, hidden=1280, tag=""):
    """Tail of a level-0 transformer block: attention output projection + residual + norm3 + feed-forward + residual in one launch
    vs gemm(residual) + the fused LayerNorm / feed-forward launch."""
    a, x = rnd(M, C), rnd(M, C)
    wo, bo = rnd(C, C, scale=1 / math.sqrt(C)), rnd(C)
    ff = ops.FeedForward(rnd(2 * hidden, C, scale=1 / math.sqrt(C)), rnd(2 * hidden), rnd(C, hidden, scale=1 / math.sqrt(hidden)), rnd(C))
    fl = 2.0 * M * (3 * hidden + C) * C
    lnp = (rnd(C), rnd(C), 1e-5)
    for rep in range(2):
        for fused in (True, False):
            ops.FF_PROJ_FUSED = fused
            t = timeit(lambda: ff.after_attention(a, wo, bo, x, lnp))
            print(f"ffproj{tag:8s} M={M:6d} C={C} hidden={hidden} {'one launch      ' if fused else 'gemm + fused ff '} round {rep}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
    ops.FF_PROJ_FUSED = True


def bench_ff_proj_h16(M, C=320, hidden=1280, tag=""):
    """The same tail in the fp16 precision: one launch (fp32 h kept in the accumulators) vs the four launches it replaces."""
    F16 = torch.float16
    r16 = lambda *s, scale=1.0: (torch.randn(*s, device="cuda") * scale).to(F16)  # noqa: E731
    a, x = r16(M, C), torch.randn(M, C, device="cuda")
    wo, bo = r16(C, C, scale=1 / math.sqrt(C)), r16(C)
    ff = ops.FeedForward(r16(2 * hidden, C, scale=1 / math.sqrt(C)), r16(2 * hidden), r16(C, hidden, scale=1 / math.sqrt(hidden)), r16(C))
    fl = 2.0 * M * (3 * hidden + C) * C
    lnp = (r16(C), r16(C), 1e-5)
    for rep in range(2):
        for fused in (True, False):
            ops.FF_PROJ_FUSED = fused
            t = timeit(lambda: ff.after_attention_f16(a, wo, bo, x, lnp, False))
            print(f"ffproj16{tag:8s} M={M:6d} C={C} hidden={hidden} {'one launch      ' if fused else 'four launches   '} round {rep}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)
    ops.FF_PROJ_FUSED = True


def bench_conv(B, H, W, Cin, Cout, stride=1, upsample=False, tag=""):
    x = rnd(B, H, W, Cin)
    wt = rnd(Cout, 9 * Cin, scale=1 / math.sqrt(9 * Cin))
    b = rnd(Cout)
    rb = rnd(B, Cout)
    t = timeit(lambda: ops.conv3x3(x, wt, bias=b, rowbias=rb, stride=stride, upsample=upsample))
    Ho, Wo = ops.conv_out_hw(H, W, stride, 1, upsample)
    fl = 2.0 * B * Ho * Wo * 9 * Cin * Cout
    print(f"conv{tag:10s} B={B:3d} {H}x{W} {Cin:5d}->{Cout:5d} s{stride} up{int(upsample)}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)


QS = os.environ.get("DM4D_BENCH_QSCALED", "1") != "0"  # the entry the model uses (Q pre-scaled); 0 = scale inside the kernel


def bench_attn(batch, heads, L, tag=""):
    C = heads * 64
    qkv = rnd(batch * L, 3 * C)
    if QS:
        qkv[:, :C] *= 0.125 * ops.LOG2E
    t = timeit(lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch, heads, L, q_scaled=QS), iters=10)
    fl = 4.0 * batch * heads * L * L * 64
    print(f"attn{tag:10s} b={batch:3d} h={heads:3d} L={L:6d}  {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)


def bench_gn(B, HW, C, tag=""):
    x = rnd(B, HW, C)
    g, bt = rnd(C), rnd(C)
    t = timeit(lambda: ops.groupnorm(x, g, bt, 32, 1e-5, silu=True))
    by = 2.0 * B * HW * C * 2  # algorithmic: read once + write once, bf16
    from diffuman4d_amd.host import lib
    lib.load().dm4d_tune_set_groupnorm_resident(0)  # A/B: the statistics + apply kernel pair on the same shape
    t2 = timeit(lambda: ops.groupnorm(x, g, bt, 32, 1e-5, silu=True))
    lib.load().dm4d_tune_set_groupnorm_resident(1)
    print(f"gn  {tag:10s} B={B:3d} HW={HW:6d} C={C:5d}  {t*1e6:9.1f} us  {by/t/1e9:7.1f} GB/s(alg)   two-launch {t2*1e6:7.1f} us",
          flush=True)


def bench_ln(M, C, tag=""):
    x = rnd(M, C)
    g, bt = rnd(C), rnd(C)
    t = timeit(lambda: ops.layernorm(x, g, bt))
    by = 2.0 * M * C * 2
    print(f"ln  {tag:10s} M={M:6d} C={C:5d}  {t*1e6:9.1f} us  {by/t/1e9:7.1f} GB/s(alg)", flush=True)


def vendor_yardstick():
    """The vendor libraries on the same shapes, as a yardstick only (nothing in the product calls them): hipBLASLt / rocBLAS
    through torch.nn.functional.linear (bias epilogue, no GEGLU / residual fusion) and the flash attention torch ships
    (scaled_dot_product_attention restricted to its fused backends)."""
    import torch.nn.functional as F
    print("vendor yardstick: torch", torch.__version__, flush=True)
    B = 32
    for lvl, (hw, c) in enumerate([(2880, 320), (720, 640), (180, 1280), (45, 1280)]):
        M = B * hw
        for tag, N, K in ((f"qkv L{lvl}", 3 * c, c), (f"out L{lvl}", c, c), (f"ff1 L{lvl} (2N, no GEGLU)", 8 * c, c),
                          (f"ff2 L{lvl}", c, 4 * c)):
            a, w, b = rnd(M, K), rnd(N, K, scale=1 / math.sqrt(K)), rnd(N)
            t = timeit(lambda: F.linear(a, w, b))
            print(f"F.linear {tag:26s} M={M:6d} N={N:5d} K={K:5d}  {t*1e6:9.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF/s", flush=True)
    try:
        from torch.nn.attention import SDPBackend, sdpa_kernel
        for tag, batch, heads, L in (("2D L0", 32, 5, 2880), ("3D L1 F16", 2, 10, 11520), ("3D L1 F24", 2, 10, 17280),
                                     ("3D L2 F16", 2, 20, 2880), ("3D L1 128", 2, 10, 65536)):
            q, k, v = (rnd(batch, heads, L, 64) for _ in range(3))
            with sdpa_kernel([SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION]):
                t = timeit(lambda: F.scaled_dot_product_attention(q, k, v), iters=5, warmup=2)
            print(f"SDPA     {tag:26s} b={batch:3d} h={heads:3d} L={L:6d}  {t*1e6:9.1f} us  {4.0*batch*heads*L*L*64/t/1e12:7.1f} TF/s",
                  flush=True)
    except Exception as e:  # a yardstick, not a dependency
        print("SDPA yardstick unavailable:", type(e).__name__, str(e)[:200], flush=True)


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only == "vendor":
        vendor_yardstick()
        return
    if only == "gemm":
        B = 32
        for lvl, (hw, c) in enumerate([(2880, 320), (720, 640), (180, 1280), (45, 1280)]):
            M = B * hw
            bench_gemm(M, 3 * c, c, residual=False, tag=f" qkv L{lvl}")
            bench_gemm(M, c, c, tag=f" out L{lvl}")
            bench_gemm(M, c, c, residual=False, tag=f" pin L{lvl}")
            bench_gemm(M, 4 * c, c, geglu=True, residual=False, tag=f" ff1 L{lvl}")
            bench_gemm(M, c, 4 * c, tag=f" ff2 L{lvl}")
        bench_conv(B, 72, 40, 320, 320, tag=" L0")
        bench_conv(B, 72, 40, 960, 320, tag=" L0 up")
        bench_conv(B, 36, 20, 640, 640, tag=" L1")
        bench_conv(B, 36, 20, 1920, 640, tag=" L1 up")
        bench_conv(B, 18, 10, 1280, 1280, tag=" L2")
        bench_conv(B, 9, 5, 1280, 1280, tag=" L3")
        return
    if only == "conv":  # the stride-1 3x3 convolutions of a UNet call: one task (CFG batch 32 / 48) and a stack of two; the VAE's
        for B in (32, 64, 96):
            bench_conv(B, 72, 40, 320, 320, tag=f" L0 B{B}")
            bench_conv(B, 72, 40, 960, 320, tag=f" L0up B{B}")
            bench_conv(B, 72, 40, 640, 320, tag=f" L0u2 B{B}")
            bench_conv(B, 36, 20, 640, 640, tag=f" L1 B{B}")
            bench_conv(B, 36, 20, 1920, 640, tag=f" L1up B{B}")
            bench_conv(B, 36, 20, 320, 640, tag=f" L1in B{B}")
            bench_conv(B, 18, 10, 1280, 1280, tag=f" L2 B{B}")
            bench_conv(B, 18, 10, 2560, 1280, tag=f" L2up B{B}")
        bench_conv(8, 576, 320, 128, 128, tag=" vae128")
        bench_conv(8, 288, 160, 256, 256, tag=" vae256")
        bench_conv(8, 144, 80, 512, 512, tag=" vae512")
        return
    if only == "ff":
        bench_ff(32 * 2880, tag=" L0 F16")
        bench_ff(48 * 2880, tag=" L0 F24")
        return
    if only == "ffproj":
        bench_ff_proj(32 * 2880, tag=" L0 F16")
        bench_ff_proj(48 * 2880, tag=" L0 F24")
        return
    if only == "ffproj16":
        bench_ff_proj_h16(32 * 2880, tag=" L0 F16")
        bench_ff_proj_h16(48 * 2880, tag=" L0 F24")
        return
    if only == "attn":
        print("attn q_scaled:", QS, flush=True)
        bench_attn(32, 5, 2880, " 2D L0")
        bench_attn(48, 5, 2880, " 2D L0 F24")
        bench_attn(2, 10, 11520, " 3D L1 F16")
        bench_attn(2, 10, 17280, " 3D L1 F24")
        bench_attn(2, 20, 2880, " 3D L2 F16")
        bench_attn(2, 20, 4320, " 3D L2 F24")
        bench_attn(2, 20, 720, " 3D mid")
        bench_attn(2, 10, 65536, " 3D L1 128")
        return
    if only == "l3":
        for B in (32, 48):
            bench_conv(B, 9, 5, 1280, 1280, tag=f" L3 B{B}")
            bench_conv(B, 9, 5, 2560, 1280, tag=f" L3 up B{B}")
        return
    if only == "gn":
        for B in (32, 48):
            bench_gn(B, 2880, 320, f" L0 B{B}")
            bench_gn(B, 2880, 640, f" L0c B{B}")
            bench_gn(B, 2880, 960, f" L0cc B{B}")
            bench_gn(B, 720, 320, f" L1in B{B}")
            bench_gn(B, 720, 640, f" L1 B{B}")
            bench_gn(B, 720, 1280, f" L1c B{B}")
            bench_gn(B, 720, 1920, f" L1cc B{B}")
            bench_gn(B, 180, 1280, f" L2 B{B}")
            bench_gn(B, 180, 1920, f" L2c B{B}")
            bench_gn(B, 180, 2560, f" L2cc B{B}")
            bench_gn(B, 45, 1280, f" L3 B{B}")
            bench_gn(B, 45, 2560, f" L3c B{B}")
        return
    print("device:", torch.cuda.get_device_name(0), "attn q_scaled:", QS, flush=True)
    B = 32  # F=16, CFG
    # GEMMs of one transformer block per level
    for lvl, (hw, c) in enumerate([(2880, 320), (720, 640), (180, 1280), (45, 1280)]):
        M = B * hw
        bench_gemm(M, 3 * c, c, residual=False, tag=f" qkv L{lvl}")
        bench_gemm(M, c, c, tag=f" out L{lvl}")
        bench_gemm(M, 4 * c, c, geglu=True, residual=False, tag=f" ff1 L{lvl}")
        bench_gemm(M, c, 4 * c, tag=f" ff2 L{lvl}")
    # convs
    bench_conv(B, 72, 40, 320, 320, tag=" L0")
    bench_conv(B, 72, 40, 960, 320, tag=" L0 up")
    bench_conv(B, 72, 40, 320, 320, stride=2, tag=" L0 down")
    bench_conv(B, 36, 20, 640, 640, tag=" L1")
    bench_conv(B, 36, 20, 1920, 640, tag=" L1 up")
    bench_conv(B, 18, 10, 1280, 1280, tag=" L2")
    bench_conv(B, 18, 10, 2560, 1280, tag=" L2 up")
    bench_conv(B, 9, 5, 1280, 1280, tag=" L3")
    bench_conv(B, 9, 5, 2560, 1280, tag=" L3 up")
    bench_conv(B, 18, 10, 1280, 1280, upsample=True, tag=" L2 ups")
    bench_conv(B, 72, 40, 32, 320, tag=" conv_in")
    bench_conv(B, 72, 40, 320, 4, tag=" conv_out")
    # attention: 2-D L0, 3-D L1/L2/mid for F=16 and F=24
    bench_attn(32, 5, 2880, " 2D L0")
    bench_attn(2, 10, 11520, " 3D L1 F16")
    bench_attn(2, 10, 17280, " 3D L1 F24")
    bench_attn(2, 20, 2880, " 3D L2 F16")
    bench_attn(2, 20, 4320, " 3D L2 F24")
    bench_attn(2, 20, 720, " 3D mid")
    bench_attn(2, 10, 65536, " 3D L1 128")
    # norms
    bench_gn(B, 2880, 320, " L0")
    bench_gn(B, 720, 640, " L1")
    bench_gn(B, 180, 1280, " L2")
    bench_ln(B * 2880, 320, " L0")
    bench_ln(B * 720, 640, " L1")


if __name__ == "__main__":
    main()
