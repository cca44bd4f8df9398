#!/bin/bash
# strip convolution form 2 with 4 weight-slab buffers (256x128, 128x64): bit-identity, per-launch timing, whole-step A/B vs the 2-buffer build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c18; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/dev/strip_ab.py > $O/strip_ab.log 2>&1; echo "rc=$?" >> $O/strip_ab.log
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "conv" ) > $O/pytest_conv.log 2>&1
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for v in nsb2 new nsb2 new; do
  if [ $v = nsb2 ]; then cp tools/dev/libdm4d_nsb2.so diffuman4d_amd/libdm4d.so; else cp /tmp/new.so diffuman4d_amd/libdm4d.so; fi
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
grep -E "DIFFERENT|MISMATCH|rc=" $O/strip_ab.log | head; grep -E "^B32|^sum" $O/strip_ab.log; tail -2 $O/pytest_conv.log
