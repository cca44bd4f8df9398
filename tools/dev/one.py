import sys, math, torch
sys.path.insert(0, '.')
from diffuman4d_amd.host import ops
BF=torch.bfloat16
def rnd(*s, scale=1.0): return (torch.randn(*s, device='cuda')*scale).to(BF)
which=sys.argv[1]
if which=='conv_l1':
    x=rnd(32,36,20,640); wt=rnd(640,9*640,scale=1/math.sqrt(9*640)); b=rnd(640); rb=rnd(32,640)
    f=lambda: ops.conv3x3(x,wt,bias=b,rowbias=rb)
elif which=='attn_l1':
    C=640; qkv=rnd(2*11520,3*C)
    f=lambda: ops.attention(qkv[:,:C],qkv[:,C:2*C],qkv[:,2*C:],2,10,11520)
elif which=='gemm_ff2_l1':
    a=rnd(23040,2560); w=rnd(640,2560,scale=1/50); b=rnd(640); r=rnd(23040,640)
    f=lambda: ops.gemm(a,w,bias=b,residual=r)
for _ in range(5): f()
torch.cuda.synchronize()
