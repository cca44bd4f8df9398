#!/bin/bash
# phase-decomposed Upsample2D: op parity, A/B vs the gather kernel, model parity, bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c19; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv" ) > $O/pytest_conv.log 2>&1
timeout 600 python tools/dev/up2x_ab.py > $O/up2x_ab.log 2>&1
( timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -x ) > $O/pytest_model.log 2>&1
timeout 600 python bench.py --steps 24 --warmup 4 > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|Error|error" $O/pytest_conv.log | tail -5; cat $O/up2x_ab.log | grep -v amdgpu; grep -E "passed|failed" $O/pytest_model.log | tail -3; cut -c1-330 $O/bench.json; echo; grep -o '"kernel_breakdown_one_step".\{0,300\}' $O/bench.json; grep -o '"parity".\{0,400\}' $O/bench.json; grep -o '"vae".\{0,300\}' $O/bench.json
