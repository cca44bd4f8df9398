#!/bin/bash
# Linear form 2 (gemm_lin2_kernel) A/B per shape; whole-step A/B of the strip convolution forms in one process environment
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c15; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/dev/lin_ab.py > $O/lin_ab.log 2>&1; echo "rc=$?" >> $O/lin_ab.log
for f in 1 2 1 2; do
  timeout 400 python tools/dev/bench_with_form.py $f --steps 20 --warmup 4 --no-cpu-baseline --no-vae --task-streams 1 > $O/bench_form${f}_s1.json 2>> $O/bench.err
  python - "$O/bench_form${f}_s1.json" $f <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["phases"]["kernel_breakdown_one_step"] if "phases" in d else None
import re
s=open(sys.argv[1]).read(); m=re.search(r'"kernel_breakdown_one_step": (\{.*?\}\})', s)
print("form", sys.argv[2], "streams 1:", d["ms_per_step"], "ms/step", m.group(1)[:260] if m else "")
P
done
for f in 1 2; do
  timeout 400 python tools/dev/bench_with_form.py $f --steps 24 --warmup 4 --no-cpu-baseline --no-vae > $O/bench_form${f}_s2.json 2>> $O/bench.err
  python -c "
import json,sys
d=json.loads(open('$O/bench_form${f}_s2.json').read().strip().splitlines()[-1]); print('form $f streams 2:', d['ms_per_step'], 'ms/step', d['value'])"
done
cat $O/lin_ab.log
