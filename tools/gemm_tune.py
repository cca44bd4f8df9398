#!/usr/bin/env python
"""Sweep every GEMM / conv kernel configuration over the UNet's shapes (72x40 latents, F = 16 and 24)
through the tuning hook dm4d_tune_set_gemm_config and print the best id per shape.
Run on a GPU box:  python tools/gemm_tune.py > gpurun_out/gemm_tune.log
The heuristic in csrc/gemm.hip::choose_cfg is written from this table (profiles/r01_gemm_tune.log)."""
from __future__ import annotations

import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffuman4d_amd.host import lib as L, ops  # noqa: E402

H16 = len(sys.argv) > 3 and sys.argv[3] == "f16"  # the fp16 precision's launches: fp16 operands, fp32 residual / row bias, fp32 or fp16 result
BF = torch.float16 if H16 else torch.bfloat16
IDS = [1, 2, 3, 4, 13, 14, 20, 31, 32, 33, 34, 35, 36, 37, 46, 61, 63, 64, 65, 67, 69]  # every id launch_by_id knows (csrc/gemm.hip); unsupported shapes report n/a


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(BF)


IT = int(sys.argv[2]) if len(sys.argv) > 2 else 6  # launches per timing (`gemm_tune.py vae 12`: the VAE's convolutions)


def timeit(f, it=None):
    it = it or IT
    try:
        f()
    except L.Dm4dError:
        return None
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


def sweep(name, fn, flops):
    lib = L.load()
    res = {}
    for i in [0] + IDS:
        lib.dm4d_tune_set_gemm_config(i)
        res[i] = timeit(fn)
    lib.dm4d_tune_set_gemm_config(0)
    ok = {k: v for k, v in res.items() if v is not None and k != 0}
    best = min(ok, key=ok.get)
    cells = " ".join(f"{i}:{(res[i] or 0):6.0f}" for i in IDS)
    print(f"{name:34s} auto {res[0]:7.1f}us ({flops/res[0]/1e6:6.0f} TF/s) best id {best:2d} {ok[best]:7.1f}us "
          f"({flops/ok[best]/1e6:6.0f} TF/s) gain {res[0]/ok[best]:.2f}x | {cells}", flush=True)
    return res[0], ok[best]


def vae():
    """The SD VAE's stride-1 3x3 convolutions on a micro-batch of 8 images of 576 x 320 (encoder: full / half / quarter / eighth
    resolution; the decoder runs the same shapes plus the wide first layers)."""
    B = 8
    shapes = [(576, 320, 128, 128), (288, 160, 128, 256), (288, 160, 256, 256), (144, 80, 256, 512), (144, 80, 512, 512), (72, 40, 512, 512),
              (576, 320, 256, 128), (288, 160, 512, 256)]
    for (h, w, ci, co) in shapes:
        x, wt = rnd(B, h, w, ci), rnd(co, 9 * ci, scale=1 / math.sqrt(9 * ci))
        b = rnd(co)
        sweep(f"vae conv {h}x{w} {ci}->{co}", lambda: ops.conv3x3(x, wt, bias=b), 2.0 * B * h * w * 9 * ci * co)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "vae":
        return vae()
    tot_auto = tot_best = 0.0
    batches = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (32, 48)  # 64,96: the launches of a 2-task stack
    for B in batches:
        print(f"===== B = {B} =====")
        for lvl, (h, w, c) in enumerate([(72, 40, 320), (36, 20, 640), (18, 10, 1280), (9, 5, 1280)]):
            M = B * h * w
            for tag, N, K, geglu, res in (("proj/out", c, c, False, True), ("qkv", 3 * c, c, False, False),
                                          ("ff1", 4 * c, c, True, False), ("ff2", c, 4 * c, False, True)):
                a, wt = rnd(M, K), rnd(2 * N if geglu else N, K, scale=1 / math.sqrt(K))
                b = rnd(2 * N if geglu else N)
                r = (torch.randn(M, N, device="cuda") if H16 else rnd(M, N)) if res else None  # fp16 precision: the fp32 residual stream
                cnt = {"proj/out": 3, "qkv": 1, "ff1": 1, "ff2": 1}[tag] * (5 if lvl < 3 else 1)
                t0, t1 = sweep(f"gemm L{lvl} {tag} M{M} N{N} K{K}", lambda: ops.gemm(a, wt, bias=b, residual=r, geglu=geglu, out_f32=H16 and res),
                               2.0 * M * K * (2 * N if geglu else N))
                tot_auto += cnt * t0
                tot_best += cnt * t1
        convs = [(72, 40, 320, 320, 1, False, 9), (72, 40, 960, 320, 1, False, 1), (72, 40, 640, 320, 1, False, 2),
                 (72, 40, 320, 320, 2, False, 1), (36, 20, 320, 640, 1, False, 1), (36, 20, 640, 640, 1, False, 8),
                 (36, 20, 1920, 640, 1, False, 1), (36, 20, 1280, 640, 1, False, 1), (36, 20, 960, 640, 1, False, 1),
                 (36, 20, 640, 640, 2, False, 1), (36, 20, 640, 640, 1, True, 1), (18, 10, 640, 1280, 1, False, 1),
                 (18, 10, 1280, 1280, 1, False, 8), (18, 10, 2560, 1280, 1, False, 2), (18, 10, 1920, 1280, 1, False, 1),
                 (18, 10, 1280, 1280, 2, False, 1), (18, 10, 1280, 1280, 1, True, 1), (9, 5, 1280, 1280, 1, False, 11),
                 (9, 5, 2560, 1280, 1, False, 3), (9, 5, 1280, 1280, 1, True, 1)]
        for (h, w, ci, co, st, up, cnt) in convs:
            x, wt = rnd(B, h, w, ci), rnd(co, 9 * ci, scale=1 / math.sqrt(9 * ci))
            b, rb = rnd(co), (torch.randn(B, co, device="cuda") if H16 else rnd(B, co))  # fp16 precision: conv1's fp32 time-embedding row, fp16 result
            ho, wo = ops.conv_out_hw(h, w, st, 1, up)
            t0, t1 = sweep(f"conv {h}x{w} {ci}->{co} s{st} up{int(up)}",
                           lambda: ops.conv3x3(x, wt, bias=b, rowbias=rb, stride=st, upsample=up),
                           2.0 * B * ho * wo * 9 * ci * co)
            tot_auto += cnt * t0
            tot_best += cnt * t1
    print(f"weighted total per (F=16 + F=24) UNet pair: auto {tot_auto/1e3:.2f} ms, best-per-shape {tot_best/1e3:.2f} ms")


if __name__ == "__main__":
    main()
