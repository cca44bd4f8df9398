#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c24; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gn_" ) > $O/pytest_gn.log 2>&1
timeout 300 python tests/opbench.py gn > $O/opbench_gn.log 2>&1
timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-vae > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_gn.log; grep -v amdgpu $O/opbench_gn.log; cut -c1-300 $O/bench.json; echo; grep -o '"kernel_breakdown_one_step".\{0,420\}' $O/bench.json
