#!/bin/bash
# Epilogue round 2: split-K partials through the staged write-out, branch-free residual read-back (STRAIGHT, per-kernel budget)
# vs the same build without it vs the previous build; 256x256 Linear tiles (ids 66 / 67) in the id sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c29; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm or conv or geglu or logits or up2x or linear" ) > $O/pytest_ops.log 2>&1
tail -2 $O/pytest_ops.log
timeout 400 python tools/dev/lin_ab.py > $O/lin_ab.log 2>&1; tail -42 $O/lin_ab.log | cut -c 1-250
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for v in base straight0 new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  timeout 300 python tests/opbench.py gemm > $O/opbench_$v.log 2>&1
done
paste <(cut -c 1-75 $O/opbench_base.log) <(cut -c 50-75 $O/opbench_straight0.log) <(cut -c 50-75 $O/opbench_new.log)
for v in base straight0 new base straight0 new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-vae --task-streams 1 > $O/bench_$v.json 2>> $O/bench.err
  python -c "
import json,re
s=open('$O/bench_$v.json').read(); d=json.loads(s.strip().splitlines()[-1]); m=re.search(r'\"kernel_breakdown_one_step\": (\{.*?\}\})', s)
print('$v', d['ms_per_step'], d['value'], m.group(1)[:200])"
done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
