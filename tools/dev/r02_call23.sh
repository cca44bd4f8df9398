#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c23; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/dev/attn_fp8_bench.py > $O/nw8.log 2>&1
DM4D_FP8_NW=4 timeout 300 python tools/dev/attn_fp8_bench.py > $O/nw4.log 2>&1
DM4D_FP8_NW=4 timeout 300 python tests/opcheck.py attn_fp8 > $O/opcheck_nw4.log 2>&1
echo "== 8 waves"; grep -v amdgpu $O/nw8.log; echo "== 4 waves"; grep -v amdgpu $O/nw4.log; grep -v amdgpu $O/opcheck_nw4.log | tail -9
