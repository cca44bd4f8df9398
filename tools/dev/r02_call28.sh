#!/bin/bash
# Transposed-accumulator epilogue (bias as the first MFMA k step, 4-column staging stores, bf16 staging for plain layers):
# op parity, per-shape A/B against the previous build, whole-step A/B, vendor-library yardstick
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c28; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm or conv or geglu or logits or up2x or linear" ) > $O/pytest_ops.log 2>&1
tail -3 $O/pytest_ops.log
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for v in base new; do
  if [ $v = base ]; then cp tools/dev/libdm4d_base.so diffuman4d_amd/libdm4d.so; else cp /tmp/new.so diffuman4d_amd/libdm4d.so; fi
  timeout 300 python tests/opbench.py gemm > $O/opbench_$v.log 2>&1
done
paste <(cut -c 1-75 $O/opbench_base.log) <(cut -c 50-75 $O/opbench_new.log)
for v in base new base new; do
  if [ $v = base ]; then cp tools/dev/libdm4d_base.so diffuman4d_amd/libdm4d.so; else cp /tmp/new.so diffuman4d_amd/libdm4d.so; fi
  timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-vae --task-streams 1 > $O/bench_$v.json 2>> $O/bench.err
  python -c "
import json,re
s=open('$O/bench_$v.json').read(); d=json.loads(s.strip().splitlines()[-1]); m=re.search(r'\"kernel_breakdown_one_step\": (\{.*?\}\})', s)
print('$v', d['ms_per_step'], d['value'], m.group(1)[:200])"
done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
timeout 300 python tests/opbench.py vendor > $O/vendor.log 2>&1; cat $O/vendor.log
( timeout 420 python -m pytest tests/test_model_gpu.py -m gpu -q -x ) > $O/pytest_model.log 2>&1; tail -2 $O/pytest_model.log
