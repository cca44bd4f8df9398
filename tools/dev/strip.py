"""Correctness + timing of the strip conv kernel configurations (31-34) against the torch fp32 conv and config 1."""
import math
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from diffuman4d_amd.host import lib as L, ops  # noqa: E402

BF = torch.bfloat16
lib = L.load()
IDS = [int(a) for a in sys.argv[1:]] or [31, 32, 33, 34]


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(BF)


def ref_conv(x, wt, b, rb):
    B, H, W, C = x.shape
    co = wt.shape[0]
    w4 = wt.float().view(co, 3, 3, C).permute(0, 3, 1, 2)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w4, b.float(), padding=1)
    y = y + rb.float()[:, :, None, None]
    return y.permute(0, 2, 3, 1)


bad = 0
for (B, h, w, ci, co) in [(1, 8, 8, 64, 64), (3, 9, 5, 128, 192), (2, 7, 3, 64, 100), (5, 4, 1, 64, 64), (2, 18, 10, 320, 128),
                          (1, 72, 40, 64, 320), (7, 5, 9, 192, 72)]:
    x, wt = rnd(B, h, w, ci), rnd(co, 9 * ci, scale=1 / math.sqrt(9 * ci))
    b, rb = rnd(co), rnd(B, co)
    ref = ref_conv(x, wt, b, rb)
    for i in IDS:
        lib.dm4d_tune_set_gemm_config(i)
        try:
            y = ops.conv3x3(x, wt, bias=b, rowbias=rb)
        except L.Dm4dError as e:
            print("cfg", i, "rejected", (B, h, w, ci, co), e)
            continue
        err = (y.float() - ref).abs().max().item()
        ok = err < 3e-2
        bad += not ok
        print(f"cfg {i} shape {(B, h, w, ci, co)} max err {err:.4f} {'ok' if ok else 'FAIL'}")
lib.dm4d_tune_set_gemm_config(0)
print("FAILURES", bad)


def timeit(f, it=8):
    f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for B in (32, 48):
    for (h, w, ci, co) in [(72, 40, 320, 320), (72, 40, 960, 320), (72, 40, 640, 320), (36, 20, 320, 640), (36, 20, 640, 640),
                           (36, 20, 1920, 640), (36, 20, 1280, 640), (18, 10, 640, 1280), (18, 10, 1280, 1280),
                           (18, 10, 2560, 1280), (9, 5, 1280, 1280), (9, 5, 2560, 1280)]:
        x, wt = rnd(B, h, w, ci), rnd(co, 9 * ci, scale=1 / math.sqrt(9 * ci))
        b, rb = rnd(co), rnd(B, co)
        fl = 2.0 * B * h * w * 9 * ci * co
        cells = []
        for i in [0] + IDS:
            lib.dm4d_tune_set_gemm_config(i)
            t = timeit(lambda: ops.conv3x3(x, wt, bias=b, rowbias=rb))
            cells.append(f"{i}:{t:7.1f}us {fl / t / 1e6:5.0f}TF")
        lib.dm4d_tune_set_gemm_config(0)
        print(f"B{B} {h}x{w} {ci}->{co}: " + "  ".join(cells), flush=True)
