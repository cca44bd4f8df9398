#!/bin/bash
# GroupNorm: wave-parallel group reduction in the single-launch kernel; Linear heuristic with ids 67 / 62: parity, per-shape and whole-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c30; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm or geglu or groupnorm or gn or linear" ) > $O/pytest_ops.log 2>&1
tail -2 $O/pytest_ops.log
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for v in prev new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  timeout 300 python tests/opbench.py gn > $O/opbench_gn_$v.log 2>&1
done
paste <(cut -c 1-75 $O/opbench_gn_prev.log) <(cut -c 44-75 $O/opbench_gn_new.log)
for v in prev new prev new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-vae --task-streams 1 > $O/bench_$v.json 2>> $O/bench.err
  python -c "
import json,re
s=open('$O/bench_$v.json').read(); d=json.loads(s.strip().splitlines()[-1]); m=re.search(r'\"kernel_breakdown_one_step\": (\{.*?\}\})', s)
print('$v', d['ms_per_step'], d['value'], m.group(1)[:330])"
done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae > $O/bench_s2.json 2>> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench_s2.json').read().strip().splitlines()[-1]); print('new, 2 task streams:', d['ms_per_step'], d['value'])"
( timeout 420 python -m pytest tests/test_model_gpu.py -m gpu -q -x ) > $O/pytest_model.log 2>&1; grep -E "passed|failed" $O/pytest_model.log | tail -1
