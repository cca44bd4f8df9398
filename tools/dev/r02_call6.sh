#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c6; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python tools/dev/forced_cfg_check.py 41 42 43 44 45 > $O/forced.log 2>&1
timeout 900 python tools/gemm_tune.py > $O/gemm_tune.log 2>&1
grep -c PASS $O/forced.log; grep -v PASS $O/forced.log | head; tail -2 $O/gemm_tune.log
