#!/usr/bin/env python
"""Schedule A/B of the hand-placed strip-convolution stream (tools/conv64/cgen.py options): conv64-only libraries built here, timed
interleaved in one process on the GPU box against the product library's 8-wave kernels (DM4D_CONV64=0), with the s_memtime stamps of
timing builds (cycles per 64-channel step).

    python tools/conv64/cab.py build            # tools/conv64/variants/conv_<name>[_t].so for every entry of variants.json
    python tools/conv64/cab.py run [name ...]   # GPU: parity vs the product library + interleaved timing
    python tools/conv64/cab.py cycles [name ...]
"""
from __future__ import annotations

import ctypes
import json
import os
import statistics
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
VDIR = HERE / "variants"
sys.path.insert(0, str(HERE))
VARIANTS = json.loads((HERE / "variants.json").read_text()) if (HERE / "variants.json").exists() else {"base": {}}
SHAPES = [("L0 B64", 64, 72, 40, 320, 320), ("L0up B64", 64, 72, 40, 960, 320), ("L1 B64", 64, 36, 20, 640, 640), ("L1up B64", 64, 36, 20, 1920, 640),
          ("L2 B64", 64, 18, 10, 1280, 1280), ("L2 B96", 96, 18, 10, 1280, 1280)]


def build():
    import cgen
    VDIR.mkdir(exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc"
    api = "/tmp/conv64_api.o"
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{ROOT / 'diffuman4d_amd' / 'csrc'}", f"-I{HERE}"]
    subprocess.run(base + ["-c", str(ROOT / "diffuman4d_amd/csrc/api.hip"), "-o", api], check=True, stderr=subprocess.DEVNULL)
    jobs = []
    for name, opts in VARIANTS.items():
        for timing in (False, True):
            tag = name + ("_t" if timing else "")
            d = Path(f"/tmp/conv64_var/{tag}")
            d.mkdir(parents=True, exist_ok=True)
            (d / "conv64_asm.inc").write_text(cgen.emit_file(dict(opts, timing=timing)))
            cmd = base + [f'-DCONV64_INC="{d}/conv64_asm.inc"'] + (["-DCONV64_TIMING"] if timing else []) + ["-c", str(HERE / "conv64_tu.hip"), "-o", str(d / "tu.o")]
            jobs.append((tag, cmd, d / "tu.o"))
    from concurrent.futures import ThreadPoolExecutor

    def one(job):
        tag, cmd, obj = job
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(VDIR / f"conv_{tag}.so"), str(obj), api], check=True, stderr=subprocess.DEVNULL)
        return tag
    with ThreadPoolExecutor(max_workers=6) as ex:
        for tag in ex.map(one, jobs):
            print("built", tag, flush=True)


def load(tag):
    lib = ctypes.CDLL(str(VDIR / f"conv_{tag}.so"))
    f = lib.dm4d_conv64_test
    f.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p]
    f.restype = ctypes.c_int
    return lib, f


def tensors(B, H, W, Cin, Cout):
    import math
    import torch
    x = (torch.randn(B, H, W, Cin, device="cuda")).to(torch.bfloat16)
    wt = (torch.randn(Cout, 9 * Cin, device="cuda") / math.sqrt(9 * Cin)).to(torch.bfloat16)
    b = torch.randn(Cout, device="cuda").to(torch.bfloat16)
    return x, wt, b, torch.empty(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)


def run(names, rounds=7, iters=6):
    import torch
    os.environ["DM4D_CONV64"] = "0"  # the product library as the 8-wave yardstick
    sys.path.insert(0, str(ROOT))
    from diffuman4d_amd.host import ops
    torch.manual_seed(0)
    libs = {n: load(n) for n in names}
    for tag, B, H, W, Cin, Cout in SHAPES:
        x, wt, b, y = tensors(B, H, W, Cin, Cout)
        st = torch.cuda.current_stream().cuda_stream
        ref = ops.conv3x3(x, wt, bias=b)
        calls = {"8wave": lambda: ops.conv3x3(x, wt, bias=b)}
        for n in names:
            f = libs[n][1]
            calls[n] = (lambda f=f: f(st, x.data_ptr(), B, H, W, Cin, wt.data_ptr(), y.data_ptr(), Cout, b.data_ptr()))
            y.zero_()
            assert calls[n]() == 0
            torch.cuda.synchronize()
            assert torch.equal(y, ref), (n, tag, float((y.float() - ref.float()).abs().max()))  # same products in the same order
        times = {n: [] for n in calls}
        for _ in range(rounds):
            for n, c in calls.items():
                c()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    c()
                e.record()
                torch.cuda.synchronize()
                times[n].append(s.elapsed_time(e) / iters)
        fl = 2.0 * B * H * W * 9 * Cin * Cout
        print(f"{tag:10s} TF/s  " + "  ".join(f"{n}: {fl / (statistics.median(times[n]) * 1e-3) / 1e12:7.1f}" for n in calls), flush=True)


def cycles(names):
    import torch
    for tag, B, H, W, Cin, Cout in SHAPES[:5]:
        x, wt, b, y = tensors(B, H, W, Cin, Cout)
        bn = 128 if Cout % 128 == 0 else 160
        nwg = ((B * H * W + 255) // 256) * ((Cout + bn - 1) // bn)
        steps = 9 * (Cin // 64)
        mf = 32 if bn == 128 else 40
        for n in names:
            lib, f = load(n + "_t")
            dbg = torch.zeros(nwg * 4 * 3, dtype=torch.int64, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            lib.dm4d_conv64_set_debug.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            lib.dm4d_conv64_set_debug(st, dbg.data_ptr())
            for _ in range(3):
                assert f(st, x.data_ptr(), B, H, W, Cin, wt.data_ptr(), y.data_ptr(), Cout, b.data_ptr()) == 0
            torch.cuda.synchronize()
            d = dbg.view(nwg * 4, 3).cpu()
            pro, loop = (d[:, 1] - d[:, 0]).float(), (d[:, 2] - d[:, 1]).float()
            print(f"{tag:10s} {n:12s} prologue {pro.median():7.0f} cycles  step {loop.median() / steps:7.1f} cycles (min {loop.min() / steps:7.1f}, p90 "
                  f"{loop.quantile(0.9) / steps:7.1f}) = {mf * 32 / (loop.median() / steps):.3f} of the matrix pipe; {nwg} workgroups", flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1]
    names = sys.argv[2:] or list(VARIANTS)
    {"build": build, "run": lambda: run(names), "cycles": lambda: cycles(names)}[cmd]()
