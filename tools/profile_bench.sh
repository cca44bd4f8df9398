#!/bin/bash
# Profiles of the bench command on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/profile_bench.sh <tag> [precision] [stats-only]'
#   precision: fast (default) | fp16 | parity = bench.py --precision ...; a third argument limits the run to step 1 (the kernel table)
# 1. rocprofv3 --kernel-trace --stats      -> gpurun_out/<tag>_kernel_stats.txt   (per-kernel time table)
# 2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only, as MI355X_MICROARCH.md's HBM section
#    prescribes)                            -> gpurun_out/<tag>_attn_traffic_pmc.json (per-launch averages for attn_kernel)
# 3. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -> gpurun_out/<tag>_attn_mfma_busy_pmc.json (MFMA-busy fraction at the
#    clock the chip ran, that clock, and their product against the nominal peak) + the same for the conv / Linear kernels
# Copy the files into profiles/ to have them judged.
set -u
# every rocprofv3 run is under `timeout`: after a fault in the profiled process the tool waits for ever (23 GPU-minutes lost once)
TAG=${1:-r01}
PREC=${2:-fast}
STATS_ONLY=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
# one stack of tasks in flight: the bench's roofline figures come from its one-stack-at-a-time pass, and with two task streams the
# kernel intervals of different stacks overlap in the trace.  --steps 4 --warmup 2: every pass of the run (set-up, warm-up, timed, prune,
# roofline, breakdown) is then made of whole stacks of the default task_batch (2), so every launch of a kernel carries the same rows and the
# trace's average duration per kernel is the average of ONE launch shape per layer
BENCH="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --task-streams 1 --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128 --precision $PREC"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o stats -- $BENCH > "$OUT.bench.log" 2> "$OUT.stats.err"
DB=$(find "$OUT" -name "stats*results.db" | head -1)
python tools/profile_summary.py "$DB" "rocprofv3 --kernel-trace --stats -- $BENCH ($TAG; units of 2 F=16 + 1 F=24 window calls, all in stacks of the default task_batch: 2 set-up + 2 warm-up + 4 timed + 2 + 2 prune-pass + 4 roofline-pass + 2 breakdown, plus weight-init kernels)" \
  > gpurun_out/${TAG}_kernel_stats.txt
tail -1 "$OUT.bench.log" > gpurun_out/${TAG}_bench_under_rocprof.json
if [ -n "$STATS_ONLY" ]; then head -16 gpurun_out/${TAG}_kernel_stats.txt; rm -rf "$OUT"; exit 0; fi
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT" -o pmc_$C -- $BENCH > /dev/null 2> "$OUT.pmc_$C.err"
done
python tools/pmc_summary.py "$OUT" attn64_kernel FETCH_SIZE WRITE_SIZE > gpurun_out/${TAG}_attn_traffic_pmc.json
# which stacks the launches carried (bench.py reads the traffic figure only for the stack size it runs)
python - gpurun_out/${TAG}_attn_traffic_pmc.json gpurun_out/${TAG}_bench_under_rocprof.json <<'PY'
import json, sys
t, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
t["task_batch"] = b["config"]["task_batch"]
json.dump(t, open(sys.argv[1], "w"))
PY
# the same HBM-traffic figures for the other MFMA families (strip convolutions, Linear layers, the fused level-0 block tail)
for K in conv_strip2_kernel gemm_lin2_kernel ff_proj_fused_kernel gn_apply_kernel gn_stats_kernel; do
  echo "== $K"; python tools/pmc_summary.py "$OUT" $K FETCH_SIZE WRITE_SIZE
done > gpurun_out/${TAG}_other_traffic_pmc.txt 2>&1
# 3. MFMA-busy cycles of the attention kernel against the cycles the chip actually clocked (its own pass, kernel trace only)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT" -o pmc_MFMA -- $BENCH > /dev/null 2> "$OUT.pmc_MFMA.err"
DBM=$(find "$OUT" -name "pmc_MFMA*results.db" | head -1)
python tools/mfma_busy_summary.py "$DBM" attn64_kernel > gpurun_out/${TAG}_attn_mfma_busy_pmc.json
for K in conv_strip2_kernel gemm_lin2_kernel ff_proj_fused_kernel; do python tools/mfma_busy_summary.py "$DBM" $K; done > gpurun_out/${TAG}_other_mfma_busy_pmc.txt
head -12 gpurun_out/${TAG}_kernel_stats.txt
cat gpurun_out/${TAG}_attn_traffic_pmc.json gpurun_out/${TAG}_attn_mfma_busy_pmc.json gpurun_out/${TAG}_other_mfma_busy_pmc.txt
# the raw rocprofv3 databases are tens of MB per pass: gpurun merges at most 64 MiB back, so only the summaries above are kept
rm -rf "$OUT"
