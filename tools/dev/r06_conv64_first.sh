#!/bin/bash
export TMPDIR=/tmp
{
echo "=== opcheck conv (every eligible shape on conv64: DM4D_CONV64=2)"; DM4D_CONV64=2 timeout 900 python tests/opcheck.py conv h16_conv 2>&1 | tail -50
echo "=== opcheck conv (default heuristic)"; timeout 900 python tests/opcheck.py conv_strip conv_big conv_batch 2>&1 | tail -14
for r in 1 2; do
echo "=== opbench conv: 8-wave kernels (DM4D_CONV64=0)"; DM4D_CONV64=0 timeout 300 python tests/opbench.py conv 2>&1 | grep -v amdgpu.ids | tail -12
echo "=== opbench conv: conv64"; timeout 300 python tests/opbench.py conv 2>&1 | grep -v amdgpu.ids | tail -12
done
} > gpurun_out/r06_conv64_first.log 2>&1
tail -120 gpurun_out/r06_conv64_first.log
