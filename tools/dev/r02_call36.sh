#!/bin/bash
# attention: form chosen by key count (5 up to 8192 keys, 1 beyond): parity of every attention case, whole-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c36; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attn and not fp8" ) > $O/pytest_attn.log 2>&1; tail -2 $O/pytest_attn.log
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for v in prev new prev new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-vae --task-streams 1 > $O/bench_$v.json 2>> $O/bench.err
  python -c "
import json,re
s=open('$O/bench_$v.json').read(); d=json.loads(s.strip().splitlines()[-1]); m=re.search(r'\"kernel_breakdown_one_step\": (\{.*?\}\})', s)
print('$v', d['ms_per_step'], d['value'], d['roofline']['achieved'], m.group(1)[-90:])"
done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vae > $O/bench_s2.json 2>> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench_s2.json').read().strip().splitlines()[-1]); print('new, 2 task streams:', d['ms_per_step'], d['value'], d['roofline']['frac'])"
