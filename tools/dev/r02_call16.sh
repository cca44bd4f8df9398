#!/bin/bash
# fitted GELU in the GEGLU epilogue + Linear form-2 heuristic: op parity, model parity, per-shape timing, bench A/B vs the erf build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c16; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gemm or geglu" ) > $O/pytest_gemm.log 2>&1
( timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -x ) > $O/pytest_model.log 2>&1
timeout 300 python tests/opbench.py > $O/opbench.log 2>&1
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for v in erf new erf new; do
  if [ $v = erf ]; then cp tools/dev/libdm4d_erf.so diffuman4d_amd/libdm4d.so; else cp /tmp/new.so diffuman4d_amd/libdm4d.so; fi
  timeout 400 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-vae > $O/bench_$v.json 2>> $O/bench.err
  python -c "
import json,re
s=open('$O/bench_$v.json').read(); d=json.loads(s.strip().splitlines()[-1]); m=re.search(r'\"kernel_breakdown_one_step\": (\{.*?\}\})', s)
print('$v', d['ms_per_step'], d['value'], m.group(1)[:120], json.dumps(d.get('parity'))[:200])"
done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
tail -3 $O/pytest_gemm.log; tail -3 $O/pytest_model.log; grep -E "^gemm" $O/opbench.log
