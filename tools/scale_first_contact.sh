#!/bin/bash
# First contact with an 8-GPU MI355X node (none was available to rounds 1-6: RCCL has only ever run at world size 1 here).
# Runs, in order and each to its own log under gpurun_out/scale/:
#   1. the driver's scaling lines        python bench.py --gpus N --steps 20 --warmup 5        N = 1, 2, 4, 8   (grid mode at N > 1)
#   2. the same grid job, hybrid deal    python bench.py --gpus N --mode hybrid ...            N = 4, 8         (frame-sharded tail waves)
#   3. the latency mode                  python bench.py --gpus N --mode frame-shard ...       N = 2, 4, 8      (BASELINE.json configs[3])
#   4. the grid job without task stacks  python bench.py --gpus N --task-batch 1 ...           N = 8            (every line above runs the runner's
#                                        defaults, three streams of 2-task stacks per rank: at 8 ranks a round deals 19 / 6 tasks per rank,
#                                        and whether stacks still pay at that depth is the first thing to read off)
#   5. the GPU tests that need > 1 rank  python -m pytest tests/test_bench_gpu.py -m gpu -q
# and prints one summary row per line: n_gpus, mode, latents/s (secondary.grid for the 1 -> N curve), ms_per_step.
# Usage: bash tools/scale_first_contact.sh [steps] [warmup]        (HSA_ENABLE_IPC_MODE_LEGACY=0 is exported: dmabuf IPC for RCCL)
set -u
steps=${1:-20}; warm=${2:-5}
out=gpurun_out/scale; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONUNBUFFERED=1
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $ngpu"
run() {  # $1 = N, $2 = mode flag (may be empty), $3 = tag
  [ "$1" -le "$ngpu" ] || { echo "skip $3: needs $1 GPUs"; return; }
  timeout 1200 python bench.py --gpus $1 --steps $steps --warmup $warm $2 > $out/$3.json 2> $out/$3.err || echo "$3: exit code $?"
  python - "$out/$3.json" "$3" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    g = d.get("secondary", {}).get("grid", {})
    print(f"{sys.argv[2]:24s} n_gpus={d['n_gpus']} mode={d['config']['mode']:28s} value={d['value']:9.3f} lat/s  grid={g.get('latents_per_s')}  ms/step={d['ms_per_step']}")
except Exception as e:
    print(sys.argv[2], "no JSON line:", e)
PY
}
for n in 1 2 4 8; do run $n "" grid_n$n; done
for n in 4 8; do run $n "--mode hybrid" hybrid_n$n; done
for n in 2 4 8; do run $n "--mode frame-shard" frameshard_n$n; done
run 8 "--task-batch 1" grid_n8_single_tasks
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -q > $out/pytest_multi.log 2>&1; tail -3 $out/pytest_multi.log
