#!/bin/bash
# strip convolution form 2 (static DMA issue, reads drained before each barrier): bit-identity vs form 1 (twice), timing, tests, bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c14; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/dev/strip_ab.py > $O/strip_ab.log 2>&1; echo "rc=$?" >> $O/strip_ab.log
timeout 600 python tools/dev/strip_ab.py > $O/strip_ab_2.log 2>&1; echo "rc=$?" >> $O/strip_ab_2.log
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv" ) > $O/pytest_conv.log 2>&1
( timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -x ) > $O/pytest_model.log 2>&1
timeout 600 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-vae > $O/bench.json 2> $O/bench.err
grep -E "DIFFERENT|MISMATCH|rc=|skipped" $O/strip_ab.log | head -20; grep -E "^B|^sum" $O/strip_ab.log; echo; grep -E "DIFFERENT|MISMATCH|rc=|^sum" $O/strip_ab_2.log; tail -3 $O/pytest_conv.log; tail -3 $O/pytest_model.log; cut -c1-330 $O/bench.json; echo; grep -o '"kernel_breakdown_one_step".\{0,500\}' $O/bench.json
