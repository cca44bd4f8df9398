#!/bin/bash
# task batching (tasks of a round stacked into one window call): bitwise equality with the tasks run alone, whole-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c40; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -k "task_batching" ) > $O/pytest_tb.log 2>&1; tail -2 $O/pytest_tb.log
for cfg in "1 2" "2 2" "2 1" "1 2" "2 2" "4 1" "3 2"; do
  set -- $cfg
  timeout 300 python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-vae --task-batch $1 --task-streams $2 > $O/bench_b$1_s$2.json 2>> $O/bench.err
  python -c "
import json
d=json.loads(open('$O/bench_b$1_s$2.json').read().strip().splitlines()[-1]); print('task-batch $1 streams $2:', d['ms_per_step'], 'ms/step', d['value'], 'latents/s')"
done
