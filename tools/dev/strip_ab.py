"""Strip convolution, form 1 (round-1 kernel) vs form 2 (VALU-free main loop): bit-identity on edge-case and UNet shapes,
then timing of both on the UNet's stride-1 convs.   python tools/dev/strip_ab.py [--time-only]"""
import math
import sys

import torch

sys.path.insert(0, '.')
from diffuman4d_amd.host import lib as L, ops  # noqa: E402

BF = torch.bfloat16
lib = L.load()


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(BF)


def make(B, H, W, Cin, Cout, rowbias=True, residual=False):
    x, wt = rnd(B, H, W, Cin), rnd(Cout, 9 * Cin, scale=1 / math.sqrt(9 * Cin))
    b = rnd(Cout)
    rb = rnd(B, Cout) if rowbias else None
    res = rnd(B, H, W, Cout) if residual else None
    return lambda: ops.conv3x3(x, wt, bias=b, rowbias=rb, residual=res)


def both(fn):
    out = []
    for form in (1, 2):
        lib.dm4d_tune_set_strip_form(form)
        out.append(fn().clone())
    lib.dm4d_tune_set_strip_form(2)
    return out


def timeit(fn, it=8):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


EDGE = [  # (B, H, W, Cin, Cout, rowbias, residual)
    (2, 18, 10, 64, 128, True, False), (3, 9, 5, 128, 64, False, True), (10, 36, 20, 64, 320, True, False),
    (256, 9, 5, 64, 640, False, True), (16, 36, 20, 64, 1024, False, False), (11, 36, 20, 128, 640, True, True),
    (2, 5800, 1, 64, 640, False, False), (5, 9, 5, 512, 320, True, True), (3, 9, 5, 640, 200, False, False),
    (9, 8, 8, 512, 128, False, True), (1, 72, 40, 320, 320, True, False), (7, 3, 3, 64, 64, False, False),
    (2, 1, 7, 64, 96, False, False), (3, 2, 2, 128, 320, True, False)]
UNET = [(72, 40, 320, 320), (72, 40, 960, 320), (72, 40, 640, 320), (36, 20, 320, 640), (36, 20, 640, 640),
        (36, 20, 1920, 640), (36, 20, 1280, 640), (36, 20, 960, 640), (18, 10, 640, 1280), (18, 10, 1280, 1280),
        (18, 10, 2560, 1280), (18, 10, 1920, 1280), (9, 5, 1280, 1280), (9, 5, 2560, 1280)]

bad = 0
if "--time-only" not in sys.argv:
    for forced in (0, 31, 32, 33, 34, 35):
        lib.dm4d_tune_set_gemm_config(forced)
        for (B, H, W, Cin, Cout, rb, res) in EDGE:
            try:
                a, b = both(make(B, H, W, Cin, Cout, rb, res))
            except L.Dm4dError as ex:
                print(f"cfg {forced:2d} B{B} {H}x{W} {Cin}->{Cout}: skipped ({str(ex)[:60]})")
                continue
            same = torch.equal(a, b)
            bad += not same
            print(f"cfg {forced:2d} B{B} {H}x{W} {Cin}->{Cout} rb={int(rb)} res={int(res)}: {'identical' if same else 'DIFFERENT'}"
                  + ("" if same else f" max|d|={(a.float() - b.float()).abs().max().item():.3e}"), flush=True)
    lib.dm4d_tune_set_gemm_config(0)
    for B in (32, 48):
        for (H, W, Cin, Cout) in UNET:
            a, b = both(make(B, H, W, Cin, Cout))
            same = torch.equal(a, b)
            bad += not same
            print(f"auto  B{B} {H}x{W} {Cin}->{Cout}: {'identical' if same else 'DIFFERENT'}", flush=True)
    print("MISMATCHES:", bad, flush=True)
tot = [0.0, 0.0]
for B in (32, 48):
    for (H, W, Cin, Cout) in UNET:
        fn = make(B, H, W, Cin, Cout)
        t = []
        for form in (1, 2):
            lib.dm4d_tune_set_strip_form(form)
            t.append(timeit(fn))
        lib.dm4d_tune_set_strip_form(2)
        fl = 2.0 * B * H * W * 9 * Cin * Cout
        tot[0] += t[0]
        tot[1] += t[1]
        print(f"B{B} {H}x{W} {Cin:4d}->{Cout:4d}  form1 {t[0]:7.1f} us ({fl/t[0]/1e6:6.0f} TF/s)   form2 {t[1]:7.1f} us ({fl/t[1]/1e6:6.0f} TF/s)"
              f"   {t[0]/t[1]:.3f}x", flush=True)
print(f"sum form1 {tot[0]/1e3:.2f} ms  form2 {tot[1]/1e3:.2f} ms  {tot[0]/tot[1]:.3f}x")
sys.exit(1 if bad else 0)
