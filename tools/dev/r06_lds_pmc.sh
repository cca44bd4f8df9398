#!/bin/bash
# LDS counters of the strip-convolution and Linear loops on the bench command (verdict item 3's record): LDS instructions issued,
# bank-conflict cycles, LDS-busy cycles, per launch (separate --pmc passes, kernel trace only)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --task-streams 1 --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/prof_lds
rm -rf "$OUT"
for C in SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT" -o pmc_$C -- $BENCH > /dev/null 2> "$OUT.pmc_$C.err" || echo "pass $C failed"
done
for K in conv_strip2_kernel gemm_lin2_kernel ff_proj_fused_kernel attn64_kernel; do
  echo "== $K (the avg_kb field is the counter's per-launch average)"; python tools/pmc_summary.py "$OUT" $K SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES
done > gpurun_out/r06c_lds_pmc.txt 2>&1
cat gpurun_out/r06c_lds_pmc.txt
rm -rf "$OUT"
