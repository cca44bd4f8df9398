#!/bin/bash
# Linear heuristic refined with the cold-cache sweep: whole-step A/B against the call-30 build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c32; mkdir -p $O
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for v in prev new prev new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-vae --task-streams 1 > $O/bench_$v.json 2>> $O/bench.err
  python -c "
import json,re
s=open('$O/bench_$v.json').read(); d=json.loads(s.strip().splitlines()[-1]); m=re.search(r'\"kernel_breakdown_one_step\": (\{.*?\}\})', s)
print('$v', d['ms_per_step'], d['value'], m.group(1)[:120])"
done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
