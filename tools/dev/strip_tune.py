"""Tile choice for the stride-1 strip convolution (form 2) on the UNet / VAE shapes: forced ids 31-37 vs the heuristic.
python tools/dev/strip_tune.py [--cold]
--cold: every timed launch is a single one after a 1 GiB cache flush and a re-touch of the input (the state a layer meets
inside a UNet pass: producer-warm activations, cold weights and output); median of 7."""
import math
import statistics
import sys

import torch

sys.path.insert(0, '.')
from diffuman4d_amd.host import lib as L, ops  # noqa: E402

BF = torch.bfloat16
lib = L.load()
IDS = (31, 32, 33, 34, 35, 36, 37)
NAMES = {31: "128x128", 32: "256x128", 33: "128x64", 34: "256x256", 35: "256x320", 36: "128x160", 37: "256x160"}


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


COLD = "--cold" in sys.argv
if COLD:
    FLUSH = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    _hot = timeit

    def timeit(fn, it=7):  # noqa: F811
        ts = []
        for _ in range(it):
            FLUSH.fill_(1.0)
            X_CUR[0].mul_(1.0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        return statistics.median(ts)
X_CUR = [None]

UNET = [(72, 40, 320, 320, 7), (72, 40, 960, 320, 1), (72, 40, 640, 320, 2), (36, 20, 320, 640, 1), (36, 20, 640, 640, 6),
        (36, 20, 1920, 640, 1), (36, 20, 1280, 640, 1), (36, 20, 960, 640, 1), (18, 10, 640, 1280, 1), (18, 10, 1280, 1280, 6),
        (18, 10, 2560, 1280, 2), (18, 10, 1920, 1280, 1)]
tot_a = tot_b = 0.0
for B in (32, 48):
    for (H, W, Cin, Cout, cnt) in UNET:
        x, wt = rnd(B, H, W, Cin), rnd(Cout, 9 * Cin, scale=1 / math.sqrt(9 * Cin))
        b, rb = rnd(Cout), rnd(B, Cout)
        X_CUR[0] = x
        fn = lambda: ops.conv3x3(x, wt, bias=b, rowbias=rb)  # noqa: E731
        lib.dm4d_tune_set_gemm_config(0)
        ref = fn().clone()
        t0 = timeit(fn)
        cells, best, bid = [], t0, 0
        for i in IDS:
            lib.dm4d_tune_set_gemm_config(i)
            try:
                same = torch.equal(fn(), ref)
                t = timeit(fn)
            except L.Dm4dError:
                cells.append(f"{NAMES[i]}:   n/a ")
                continue
            if t < best:
                best, bid = t, i
            cells.append(f"{NAMES[i]}:{t:7.1f}{'' if same else '!'}")
        lib.dm4d_tune_set_gemm_config(0)
        tot_a += cnt * t0
        tot_b += cnt * best
        fl = 2.0 * B * H * W * 9 * Cin * Cout
        print(f"B{B} {H}x{W} {Cin:4d}->{Cout:4d} auto {t0:7.1f} us ({fl/t0/1e6:5.0f} TF/s) | " + " ".join(cells) +
              f" | best {NAMES.get(bid, 'auto')} {t0/best:.3f}x", flush=True)
print(f"weighted per (F=16 + F=24) UNet pair: auto {tot_a/1e3:.2f} ms, best {tot_b/1e3:.2f} ms")
