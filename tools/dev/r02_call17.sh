#!/bin/bash
# in-situ bound of hiding the strip convolution's DMA latency: bench step with the wait-free (WRONG results) build vs the real one
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c17; mkdir -p $O
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for v in new nowait new nowait; do
  if [ $v = nowait ]; then cp tools/dev/libdm4d_nowait.so diffuman4d_amd/libdm4d.so; else cp /tmp/new.so diffuman4d_amd/libdm4d.so; fi
  timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-vae --task-streams 1 > $O/bench_$v.json 2>> $O/bench.err
  python -c "
import json,re
s=open('$O/bench_$v.json').read(); d=json.loads(s.strip().splitlines()[-1]); m=re.search(r'\"kernel_breakdown_one_step\": (\{.*?\}\})', s)
print('$v', d['ms_per_step'], d['value'], m.group(1)[:330])"
done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
