#!/bin/bash
# strip convolution form 2 (VALU-free main loop): bit-identity vs form 1, timing, and the same without the scheduling directives
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c12; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/dev/strip_ab.py > $O/strip_ab.log 2>&1; echo "rc=$?" >> $O/strip_ab.log
cp diffuman4d_amd/libdm4d.so /tmp/new.so
cp tools/dev/libdm4d_nosched.so diffuman4d_amd/libdm4d.so
timeout 600 python tools/dev/strip_ab.py --time-only > $O/strip_ab_nosched.log 2>&1
cp /tmp/new.so diffuman4d_amd/libdm4d.so
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv" ) > $O/pytest_conv.log 2>&1
timeout 600 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-vae > $O/bench.json 2> $O/bench.err
grep -E "DIFFERENT|MISMATCH|rc=|skipped" $O/strip_ab.log | head -20; grep -E "^B|^sum" $O/strip_ab.log; echo; grep -E "^B|^sum" $O/strip_ab_nosched.log; tail -3 $O/pytest_conv.log; cut -c1-330 $O/bench.json; echo; grep -o '"kernel_breakdown_one_step".\{0,500\}' $O/bench.json
