"""Linear layers: gemm_lin2_kernel ids (61-64) vs the heuristic's choice on the UNet's shapes: bit-identity, then timing.
python tools/dev/lin_ab.py"""
import math
import sys

import torch

sys.path.insert(0, '.')
from diffuman4d_amd.host import lib as L, ops  # noqa: E402

BF = torch.bfloat16
lib = L.load()
IDS = (61, 63, 64, 65, 67, 69)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(BF)


def make(M, N, K, geglu, res, split=False):
    wt = rnd(2 * N if geglu else N, K, scale=1 / math.sqrt(K))
    b = rnd(2 * N if geglu else N)
    r = rnd(M, N) if res else None
    if split:
        a1, a2 = rnd(M, K // 2), rnd(M, K - K // 2)
        return lambda: ops.gemm(a1, wt, bias=b, residual=r, geglu=geglu, a2=a2)
    a = rnd(M, K)
    return lambda: ops.gemm(a, wt, bias=b, residual=r, geglu=geglu)


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


def run(tag, fn, flops, cnt, tot):
    lib.dm4d_tune_set_gemm_config(0)
    ref = fn().clone()
    t0 = timeit(fn)
    cells, best = [], t0
    for i in IDS:
        lib.dm4d_tune_set_gemm_config(i)
        try:
            out = fn().clone()
        except L.Dm4dError:
            cells.append(f"{i}:   n/a     ")
            continue
        same = torch.equal(out, ref)
        t = timeit(fn)
        best = min(best, t)
        cells.append(f"{i}:{t:7.1f}{'=' if same else '!DIFF'}")
        tot["bad"] += not same
    lib.dm4d_tune_set_gemm_config(0)
    tot["auto"] += cnt * t0
    tot["best"] += cnt * best
    print(f"{tag:34s} auto {t0:7.1f} us ({flops/t0/1e6:5.0f} TF/s) | " + " ".join(cells) + f" | best {t0/best:.3f}x", flush=True)


tot = {"auto": 0.0, "best": 0.0, "bad": 0}
# edge cases first (ragged M / N tails, split A, no bias)
for (M, N, K, g, r, sp) in [(1000, 320, 320, False, True, False), (777, 200, 640, False, False, False), (4096, 640, 1280, True, False, False),
                            (300, 1280, 2560, False, True, True), (92160, 320, 640, False, True, True), (64, 64, 64, False, False, False)]:
    run(f"edge M{M} N{N} K{K} g{int(g)} sp{int(sp)}", make(M, N, K, g, r, sp), 2.0 * M * K * (2 * N if g else N), 0, tot)
for B in (32, 48):
    print(f"===== B = {B} =====")
    for lvl, (h, w, c) in enumerate([(72, 40, 320), (36, 20, 640), (18, 10, 1280), (9, 5, 1280)]):
        M = B * h * w
        for tag, N, K, geglu, res in (("proj/out", c, c, False, True), ("qkv", 3 * c, c, False, False),
                                      ("ff1", 4 * c, c, True, False), ("ff2", c, 4 * c, False, True)):
            cnt = {"proj/out": 3, "qkv": 1, "ff1": 1, "ff2": 1}[tag] * (5 if lvl < 3 else 1)
            run(f"L{lvl} {tag} M{M} N{N} K{K}", make(M, N, K, geglu, res), 2.0 * M * K * (2 * N if geglu else N), cnt, tot)
print(f"weighted Linear total per (F=16 + F=24) UNet pair: auto {tot['auto']/1e3:.2f} ms, best-per-shape {tot['best']/1e3:.2f} ms; MISMATCHES {tot['bad']}")
sys.exit(1 if tot["bad"] else 0)
