"""Fixed vs per-K cost of the Linear GEMM path: time(K) at fixed M, N for a few (M, N); prints us and the linear fit
t = t0 + k*K (t0 = prologue + epilogue + launch + tail quantisation, k = main-loop slope)."""
import math, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from diffuman4d_amd.host import ops
from opbench import timeit, rnd

def run(M, N, K, bias=True, res=False, geglu=False):
    a, w = rnd(M, K), rnd(2 * N if geglu else N, K, scale=1 / math.sqrt(K))
    b = rnd(2 * N if geglu else N) if bias else None
    r = rnd(M, N) if res else None
    return timeit(lambda: ops.gemm(a, w, bias=b, residual=r, geglu=geglu), iters=20) * 1e6

for (M, N, geglu) in ((92160, 320, False), (92160, 960, False), (92160, 1280, True), (23040, 640, False), (23040, 1920, False), (46080, 320, False)):
    ts = []
    for K in (320, 640, 1280, 2560):
        t = run(M, N, K, geglu=geglu)
        ts.append((K, t))
    k = (ts[-1][1] - ts[0][1]) / (ts[-1][0] - ts[0][0])
    t0 = ts[0][1] - k * ts[0][0]
    nn = 2 * N if geglu else N
    print(f"M={M} N={N} geglu={int(geglu)}: " + " ".join(f"K={K}:{t:7.1f}us" for K, t in ts) +
          f" | t0={t0:6.1f}us slope={k*1e3:6.2f} ns/K  main-loop rate={2.0*M*nn/k/1e6:7.1f} TF/s", flush=True)
# epilogue options at K=320
for (M, N) in ((92160, 320), (92160, 960)):
    print(f"M={M} N={N} K=320: plain {run(M,N,320,bias=False):.1f}  +bias {run(M,N,320):.1f}  +bias+res {run(M,N,320,res=True):.1f} us", flush=True)
