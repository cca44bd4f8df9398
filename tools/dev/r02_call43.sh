#!/bin/bash
# Linear heuristic experiment: id 61 for the residual layers with K <= 640 (cold-sweep winner), whole-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c43; mkdir -p $O
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/cur.so
for v in cur resid61 cur resid61; do
  if [ $v = cur ]; then cp /tmp/cur.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  timeout 100 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-vae --task-streams 1 > $O/bench_$v.json 2>> $O/bench.err
  python -c "
import json,re
s=open('$O/bench_$v.json').read(); d=json.loads(s.strip().splitlines()[-1]); m=re.search(r'\"linear\": (\{.*?\})', s)
print('$v', d['ms_per_step'], m.group(1))"
done
cp /tmp/cur.so diffuman4d_amd/libdm4d.so
