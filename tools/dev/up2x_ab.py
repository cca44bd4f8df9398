"""Upsample2D: phase-decomposed conv_up2x vs the gather kernel, UNet + VAE shapes.   python tools/dev/up2x_ab.py"""
import math
import sys

import torch

sys.path.insert(0, '.')
from diffuman4d_amd.host import ops  # noqa: E402

BF = torch.bfloat16


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(BF)


def timeit(fn, it=6):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for (B, H, W, Cin, Cout) in [(32, 36, 20, 640, 640), (32, 18, 10, 1280, 1280), (32, 9, 5, 1280, 1280), (48, 36, 20, 640, 640),
                             (48, 18, 10, 1280, 1280), (48, 9, 5, 1280, 1280), (1, 72, 40, 512, 512), (1, 144, 80, 512, 512),
                             (1, 288, 160, 256, 256)]:
    x, wt, b = rnd(B, H, W, Cin), rnd(Cout, 9 * Cin, scale=1 / math.sqrt(9 * Cin)), rnd(Cout)
    wp = ops.conv_up2x_prepare(wt)
    a = ops.conv3x3(x, wt, bias=b, upsample=True)
    c = ops.conv_up2x(x, wp, bias=b)
    rel = float((a.float() - c.float()).norm() / a.float().norm())
    t0, t1 = timeit(lambda: ops.conv3x3(x, wt, bias=b, upsample=True)), timeit(lambda: ops.conv_up2x(x, wp, bias=b))
    fl = 2.0 * B * 4 * H * W * 9 * Cin * Cout
    print(f"B{B} {H}x{W} -> x2, {Cin}->{Cout}: gather {t0:8.1f} us ({fl/t0/1e6:5.0f} TF/s nominal)  phases {t1:8.1f} us "
          f"({fl/t1/1e6:5.0f} nominal, {fl*4/9/t1/1e6:5.0f} executed)  {t0/t1:.2f}x   rel diff {rel:.2e}", flush=True)
