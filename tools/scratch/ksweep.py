import sys, math, torch
sys.path.insert(0, '.')
from diffuman4d_amd.host import ops
BF=torch.bfloat16
def rnd(*s, scale=1.0): return (torch.randn(*s, device='cuda')*scale).to(BF)
def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
M=92160
for N in (960, 320, 1280):
    for K in (64, 128, 320, 640, 1280, 2560):
        a=rnd(M,K); w=rnd(N,K,scale=1/math.sqrt(K)); 
        t=timeit(lambda: ops.gemm(a,w))
        print(f"M={M} N={N} K={K}: {t:8.1f} us  {2*M*N*K/t/1e6:7.1f} TF/s  io={(M*K+M*N)*2/t/1e6:6.2f} TB/s")
