#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c21; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attn_fp8" ) > $O/pytest_fp8.log 2>&1
timeout 300 python tests/opcheck.py attn_fp8 > $O/opcheck_fp8.log 2>&1
timeout 300 python tools/dev/attn_fp8_bench.py > $O/attn_fp8_bench.log 2>&1
tail -30 $O/pytest_fp8.log | cut -c1-300; grep -v amdgpu $O/opcheck_fp8.log | tail -12; grep -v amdgpu $O/attn_fp8_bench.log
