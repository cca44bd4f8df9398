"""One kernel shape, five launches: the target of the rocprofv3 --pmc passes of tools/dev/pmc.sh."""
import math
import sys

import torch

sys.path.insert(0, '.')
from diffuman4d_amd.host import ops  # noqa: E402

BF = torch.bfloat16


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device='cuda') * scale).to(BF)


def gemm(M, N, K, geglu=False, res=True):
    a, w = rnd(M, K), rnd(2 * N if geglu else N, K, scale=1 / math.sqrt(K))
    b = rnd(2 * N if geglu else N)
    r = rnd(M, N) if res else None
    return lambda: ops.gemm(a, w, bias=b, residual=r, geglu=geglu)


def conv(B, H, W, Cin, Cout):
    x, wt, b, rb = rnd(B, H, W, Cin), rnd(Cout, 9 * Cin, scale=1 / math.sqrt(9 * Cin)), rnd(Cout), rnd(B, Cout)
    return lambda: ops.conv3x3(x, wt, bias=b, rowbias=rb)


def attn(b, h, L):
    C = h * 64
    qkv = rnd(b * L, 3 * C)
    qkv[:, :C] *= 0.125 * ops.LOG2E
    return lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], b, h, L, q_scaled=True)


def gn(B, HW, C):
    x, g, bt = rnd(B, HW, C), rnd(C), rnd(C)
    return lambda: ops.groupnorm(x, g, bt, 32, 1e-5, silu=True)


M0, M1 = 32 * 2880, 32 * 720
KERNELS = {
    'conv_l0': lambda: conv(32, 72, 40, 320, 320), 'conv_l1': lambda: conv(32, 36, 20, 640, 640),
    'conv_l0_up': lambda: conv(32, 72, 40, 960, 320),
    'attn_l1': lambda: attn(2, 10, 11520), 'attn_l0': lambda: attn(32, 5, 2880),
    'gemm_qkv_l0': lambda: gemm(M0, 960, 320, res=False), 'gemm_out_l0': lambda: gemm(M0, 320, 320),
    'gemm_ff1_l0': lambda: gemm(M0, 1280, 320, geglu=True, res=False), 'gemm_ff2_l0': lambda: gemm(M0, 320, 1280),
    'gemm_ff1_l1': lambda: gemm(M1, 2560, 640, geglu=True, res=False), 'gemm_ff2_l1': lambda: gemm(M1, 640, 2560),
    'gn_l0': lambda: gn(32, 2880, 320), 'gn_l2': lambda: gn(32, 180, 1280),
}
if len(sys.argv) > 2:  # force one GEMM / conv kernel configuration id (tuning hook)
    from diffuman4d_amd.host import lib as _l
    _l.load().dm4d_tune_set_gemm_config(int(sys.argv[2]))
f = KERNELS[sys.argv[1]]()
for _ in range(5):
    f()
torch.cuda.synchronize()
