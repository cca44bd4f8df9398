#!/bin/bash
# round-6 records on the final tree: stacks / streams A/B, rocprofv3 profiles (kernel table + PMC), the driver's bench command
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
[ -n "${SKIP_AB:-}" ] || {
for round in 1 2; do
for cfg in "3 2" "3 4" "2 4" "2 2"; do
  set -- $cfg
  timeout 600 python bench.py $Q --task-streams $1 --task-batch $2 --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('round $round streams $1 batch $2', 'ms_per_step', d['ms_per_step'], 'value', d['value'])
"
done; done
} > gpurun_out/r06_task_batch_streams.log 2>&1
[ -n "${SKIP_AB:-}" ] || cat gpurun_out/r06_task_batch_streams.log
bash tools/profile_bench.sh ${TAG:-r06b} fast > gpurun_out/${TAG:-r06b}_profile_fast.out 2>&1; tail -25 gpurun_out/${TAG:-r06b}_profile_fast.out
bash tools/profile_bench.sh ${TAG:-r06b}_fp16 fp16 stats-only > gpurun_out/${TAG:-r06b}_profile_fp16.out 2>&1; tail -12 gpurun_out/${TAG:-r06b}_profile_fp16.out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; tail -c 300 gpurun_out/r06_bench_final.err; cut -c1-400 gpurun_out/r06_bench_final.json
