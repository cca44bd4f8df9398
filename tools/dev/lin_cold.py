"""Linear layers, tile ids under IN-MODEL cache conditions: before every timed launch the caches are flushed (1 GiB written),
then the activation operand is re-touched the way its producer kernel would leave it (weights, bias, residual and the output
stay cold).  A timing loop that repeats one launch keeps weights and output lines hot and flatters big single-workgroup tiles;
inside a UNet pass a layer meets cold weights.  Median of 7 single launches per (shape, id); ids are bit-identical.
python tools/dev/lin_cold.py [--ids=61,65,68] [--levels=0,1] [--batches=32]"""
import math
import statistics
import sys

import torch

sys.path.insert(0, '.')
from diffuman4d_amd.host import lib as L, ops  # noqa: E402

BF = torch.bfloat16
lib = L.load()
IDS = (1, 14, 46, 61, 63, 64, 65, 67, 69)
for a in sys.argv[1:]:  # --ids=61,65,68  --levels=0,1  --batches=32
    if a.startswith("--ids="):
        IDS = tuple(int(x) for x in a[6:].split(","))
LEVELS = next((tuple(int(x) for x in a[9:].split(",")) for a in sys.argv[1:] if a.startswith("--levels=")), (0, 1, 2, 3))
BATCHES = next((tuple(int(x) for x in a[10:].split(",")) for a in sys.argv[1:] if a.startswith("--batches=")), (32, 48))
FLUSH = torch.empty(1 << 28, dtype=torch.float32, device="cuda")  # 1 GiB > L2 + Infinity Cache


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).to(BF)


def make(M, N, K, geglu, res):
    wt = rnd(2 * N if geglu else N, K, scale=1 / math.sqrt(K))
    b = rnd(2 * N if geglu else N)
    r = rnd(M, N) if res else None
    a = rnd(M, K)
    return (lambda: ops.gemm(a, wt, bias=b, residual=r, geglu=geglu)), a


def time_cold(fn, a, reps=7):
    ts = []
    for _ in range(reps):
        FLUSH.fill_(1.0)
        a.mul_(1.0)  # the producer's write: A is as warm as a just-written tensor of its size can be
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts)


def run(tag, fn, a, flops, cnt, tot):
    lib.dm4d_tune_set_gemm_config(0)
    fn()
    t0 = time_cold(fn, a)
    cells, best, best_id = [], t0, 0
    for i in IDS:
        lib.dm4d_tune_set_gemm_config(i)
        try:
            fn()
        except L.Dm4dError:
            cells.append(f"{i}:  n/a ")
            continue
        t = time_cold(fn, a)
        if t < best:
            best, best_id = t, i
        cells.append(f"{i}:{t:6.1f}")
    lib.dm4d_tune_set_gemm_config(0)
    tot["auto"] += cnt * t0
    tot["best"] += cnt * best
    print(f"{tag:32s} auto {t0:6.1f} us ({flops/t0/1e6:5.0f} TF/s) | " + " ".join(cells) + f" | best {best_id:2d} {t0/best:.3f}x", flush=True)


tot = {"auto": 0.0, "best": 0.0}
for B in BATCHES:
    print(f"===== B = {B} (cold weights / output, producer-warm A) =====")
    for lvl, (h, w, c) in enumerate([(72, 40, 320), (36, 20, 640), (18, 10, 1280), (9, 5, 1280)]):
        if lvl not in LEVELS:
            continue
        M = B * h * w
        for tag, N, K, geglu, res in (("proj/out", c, c, False, True), ("qkv", 3 * c, c, False, False),
                                      ("ff1", 4 * c, c, True, False), ("ff2", c, 4 * c, False, True)):
            cnt = {"proj/out": 4, "qkv": 2, "ff1": 1, "ff2": 1}[tag] * (5 if lvl < 3 else 1)
            fn, a = make(M, N, K, geglu, res)
            run(f"L{lvl} {tag} M{M} N{N} K{K}", fn, a, 2.0 * M * K * (2 * N if geglu else N), cnt, tot)
print(f"weighted Linear total per (F=16 + F=24) UNet pair, cold: auto {tot['auto']/1e3:.2f} ms, best-per-shape {tot['best']/1e3:.2f} ms")
