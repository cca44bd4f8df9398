#!/bin/bash
# GroupNorm with shifted statistics: every gn_* parity case (incl. |mean| = 50 and 200 sigma), per-shape timing vs the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c37; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gn_" ) > $O/pytest_gn.log 2>&1; tail -3 $O/pytest_gn.log
timeout 200 python tests/opbench.py gn > $O/opbench_gn.log 2>&1; grep "^gn" $O/opbench_gn.log | cut -c 1-80
