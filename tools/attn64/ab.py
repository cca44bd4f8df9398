#!/usr/bin/env python
"""Schedule A/B of the hand-placed attention stream: builds attention-only libraries from tools/attn64/gen.py options (here, no GPU),
times them interleaved inside one process on the GPU box, and reads the s_memtime stamps of a timing build (cycles per 64-key step).

    python tools/attn64/ab.py build                      # tools/attn64/variants/attn_<name>.so for every entry of VARIANTS
    python tools/attn64/ab.py run [name ...]             # GPU: parity vs torch SDPA + interleaved timing, median of rounds
    python tools/attn64/ab.py cycles [name ...]          # GPU: timing builds (<name>_t): prologue / per-step cycles from s_memtime
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
SHAPES = [("2D L0", 32, 5, 2880), ("2D L0x2", 64, 5, 2880), ("3D L1 F16", 2, 10, 11520), ("3D L1 F16x2", 4, 10, 11520), ("3D L1 F24", 2, 10, 17280),
          ("3D L2 F16x2", 4, 20, 2880), ("3D 128sq", 1, 10, 65536)]


def build():
    import gen
    VDIR.mkdir(exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc"
    api = "/tmp/attn64_api.o"
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{ROOT / 'diffuman4d_amd' / 'csrc'}"]
    subprocess.run(base + ["-c", str(ROOT / "diffuman4d_amd/csrc/api.hip"), "-o", api], check=True)
    jobs = []
    for name, opts in VARIANTS.items():
        for timing in (False, True):
            tag = name + ("_t" if timing else "")
            d = Path(f"/tmp/attn64_var/{tag}")
            d.mkdir(parents=True, exist_ok=True)
            (d / "attn64_asm.inc").write_text(gen.emit_file(dict(opts, timing=timing)))
            obj = d / "attention.o"
            cmd = base + ["-mllvm", "-amdgpu-mfma-vgpr-form=1", f'-DATTN64_INC="{d}/attn64_asm.inc"'] + (["-DATTN64_TIMING"] if timing else []) + \
                ["-c", str(ROOT / "diffuman4d_amd/csrc/attention.hip"), "-o", str(obj)]
            jobs.append((tag, cmd, obj))
    from concurrent.futures import ThreadPoolExecutor

    def one(job):
        tag, cmd, obj = job
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(VDIR / f"attn_{tag}.so"), str(obj), api], check=True, stderr=subprocess.DEVNULL)
        return tag
    with ThreadPoolExecutor(max_workers=6) as ex:
        for tag in ex.map(one, jobs):
            print("built", tag, flush=True)


def load(tag):
    lib = ctypes.CDLL(str(VDIR / f"attn_{tag}.so"))
    f = lib.dm4d_attention_qscaled_kv_bf16
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 4 + [ctypes.c_int] * 4
    f.restype = ctypes.c_int
    return lib, f


def run(names, rounds=7, iters=6):
    import torch
    torch.manual_seed(0)
    libs = {n: load(n) for n in names}
    res = {n: {} for n in names}
    for tag, b, h, L in SHAPES:
        C = h * 64
        qkv = (torch.randn(b * L, 3 * C, device="cuda")).to(torch.bfloat16)
        qkv[:, :C] *= 0.125 * 1.4426950408889634
        out = torch.empty(b * L, C, device="cuda", dtype=torch.bfloat16)
        st = torch.cuda.current_stream().cuda_stream

        def call(f):
            rc = f(st, qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C, out.data_ptr(), 3 * C, 3 * C, 3 * C, C, b, h, L, L)
            assert rc == 0
        ref = None
        if L <= 20000:  # parity against torch's SDPA on the same numbers (first batch / head only)
            q = qkv[:L, :64].float() / (0.125 * 1.4426950408889634)
            k, v = qkv[:L, C:C + 64].float(), qkv[:L, 2 * C:2 * C + 64].float()
            ref = torch.nn.functional.scaled_dot_product_attention(q[None, None], k[None, None], v[None, None])[0, 0]
        times = {n: [] for n in names}
        for n in names:
            out.zero_()
            call(libs[n][1])
            torch.cuda.synchronize()
            if ref is not None:
                e = float((out[:L, :64].float() - ref).norm() / ref.norm())
                assert e < 4e-3, (n, tag, e)
        for _ in range(rounds):
            for n in names:
                f = libs[n][1]
                call(f)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    call(f)
                e.record()
                torch.cuda.synchronize()
                times[n].append(s.elapsed_time(e) / iters)
        fl = 4.0 * b * h * L * L * 64
        row = "  ".join(f"{n}: {fl / (statistics.median(times[n]) * 1e-3) / 1e12:7.1f}" for n in names)
        print(f"{tag:12s} b={b:3d} h={h:3d} L={L:6d}  TF/s  {row}", flush=True)
        for n in names:
            res[n][tag] = fl / (statistics.median(times[n]) * 1e-3) / 1e12
    return res


def cycles(names):
    import torch
    for tag, b, h, L in (("2D L0", 32, 5, 2880), ("3D L1 F16x2", 4, 10, 11520), ("3D 128sq", 1, 10, 65536)):
        C = h * 64
        qkv = (torch.randn(b * L, 3 * C, device="cuda")).to(torch.bfloat16)
        qkv[:, :C] *= 0.125 * 1.4426950408889634
        out = torch.empty(b * L, C, device="cuda", dtype=torch.bfloat16)
        nwg = ((L + 255) // 256) * b * h
        for n in names:
            lib, f = load(n + "_t")
            dbg = torch.zeros(nwg * 4 * 3, dtype=torch.int64, device="cuda")
            lib.dm4d_attn64_set_debug.argtypes = [ctypes.c_void_p]
            lib.dm4d_attn64_set_debug(dbg.data_ptr())
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                assert f(st, qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C, out.data_ptr(), 3 * C, 3 * C, 3 * C, C, b, h, L, L) == 0
            torch.cuda.synchronize()
            d = dbg.view(nwg * 4, 3).cpu()
            pro, loop = (d[:, 1] - d[:, 0]).float(), (d[:, 2] - d[:, 1]).float()
            nt = L // 64
            print(f"{tag:12s} {n:14s} prologue {pro.median():8.0f} cycles   step {loop.median() / nt:7.1f} cycles (min {loop.min() / nt:7.1f}, p90 "
                  f"{loop.quantile(0.9) / nt:7.1f})   = {(1024 + 128) / (loop.median() / nt):.3f} of the matrix pipe incl. row sums", flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1]
    names = sys.argv[2:] or list(VARIANTS)
    if cmd == "build":
        build()
    elif cmd == "run":
        run(names)
    elif cmd == "cycles":
        cycles(names)
    elif cmd == "one":  # one variant, one shape, many launches: the command a rocprofv3 pass profiles
        import torch
        tag, b, h, L = SHAPES[int(os.environ.get("AB_SHAPE", "6"))]
        C = h * 64
        qkv = torch.randn(b * L, 3 * C, device="cuda").to(torch.bfloat16)
        qkv[:, :C] *= 0.125 * 1.4426950408889634
        out = torch.empty(b * L, C, device="cuda", dtype=torch.bfloat16)
        lib, f = load(names[0])
        for _ in range(12):
            f(torch.cuda.current_stream().cuda_stream, qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C, out.data_ptr(), 3 * C, 3 * C, 3 * C, C, b, h, L, L)
        torch.cuda.synchronize()
        print("ran", names[0], tag)
