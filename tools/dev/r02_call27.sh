#!/bin/bash
# Linear heuristic with id 65 (second form on the 74 KB K-slab-32 geometry): gemm parity cases + whole-step A/B vs the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c27; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gemm or geglu or logits" ) > $O/pytest_gemm.log 2>&1
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for v in base new base new; do
  if [ $v = base ]; then cp tools/dev/libdm4d_base.so diffuman4d_amd/libdm4d.so; else cp /tmp/new.so diffuman4d_amd/libdm4d.so; fi
  timeout 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-vae --task-streams 1 > $O/bench_$v.json 2>> $O/bench.err
  python -c "
import json,re
s=open('$O/bench_$v.json').read(); d=json.loads(s.strip().splitlines()[-1]); m=re.search(r'\"kernel_breakdown_one_step\": (\{.*?\}\})', s)
print('$v', d['ms_per_step'], d['value'], m.group(1)[:140])"
done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
timeout 400 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-vae > $O/bench_s2.json 2>> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench_s2.json').read().strip().splitlines()[-1]); print('new, 2 streams:', d['ms_per_step'], d['value'])"
( timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -x ) > $O/pytest_model.log 2>&1
tail -2 $O/pytest_gemm.log; grep -E "passed|failed" $O/pytest_model.log | tail -1
