#!/bin/bash
# first contact of the hand-placed attention kernel: parity of every attention case, then old vs new per shape
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== opcheck attn (attn64 on)"; timeout 900 python tests/opcheck.py attn_qs h16_attn 2>&1 | tail -45
echo "=== opbench attn: 8-wave kernel (DM4D_ATTN64=0)"; DM4D_ATTN64=0 timeout 300 python tests/opbench.py attn 2>&1 | tail -10
echo "=== opbench attn: attn64"; timeout 300 python tests/opbench.py attn 2>&1 | tail -10
echo "=== opbench attn: 8-wave kernel again"; DM4D_ATTN64=0 timeout 300 python tests/opbench.py attn 2>&1 | tail -10
echo "=== opbench attn: attn64 again"; timeout 300 python tests/opbench.py attn 2>&1 | tail -10
} > gpurun_out/r06_attn64_first.log 2>&1
tail -70 gpurun_out/r06_attn64_first.log
