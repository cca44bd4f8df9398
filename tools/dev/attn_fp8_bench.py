"""fp8 vs bf16 attention on the bench shapes (72x40 latents and one 128x128-latent sequence): time incl. the pack kernels."""
import sys

import torch

sys.path.insert(0, '.')
from diffuman4d_amd.host import ops  # noqa: E402


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


for (b, h, L) in [(32, 5, 2880), (2, 10, 11520), (2, 10, 17280), (2, 20, 4320), (48, 5, 2880), (1, 10, 65536)]:
    C = h * 64
    qkv = (torch.randn(b * L, 3 * C, device="cuda")).to(torch.bfloat16)
    qkv[:, :C] *= 0.125 * ops.LOG2E
    f = lambda fp8: (ops.attention_fp8 if fp8 else ops.attention)(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], b, h, L, q_scaled=True)  # noqa: E731
    a, c = f(False), f(True)
    rel = float((a.float() - c.float()).norm() / a.float().norm())
    t0, t1 = timeit(lambda: f(False)), timeit(lambda: f(True))
    fl = 4.0 * b * h * L * L * 64
    print(f"b={b:2d} h={h:2d} L={L:6d}: bf16 {t0:8.1f} us ({fl/t0/1e6:5.0f} TF/s)  fp8 {t1:8.1f} us ({fl/t1/1e6:5.0f} TF/s incl. pack)  "
          f"{t0/t1:.2f}x  rel diff {rel:.3e}", flush=True)
