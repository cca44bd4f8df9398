#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c7; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python tools/dev/forced_cfg_check.py 35 46 > $O/forced.log 2>&1
( timeout 600 python tests/opcheck.py gemm; timeout 600 python tests/opcheck.py conv ) > $O/opcheck.log 2>&1
timeout 900 python tools/gemm_tune.py > $O/gemm_tune.log 2>&1
cat $O/forced.log | grep -v amdgpu; grep -c PASS $O/opcheck.log; grep -v PASS $O/opcheck.log | grep -v amdgpu | head -5; tail -1 $O/gemm_tune.log
