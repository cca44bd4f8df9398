#!/bin/bash
# clock and MFMA-busy of attn64_kernel (and of the 8-wave kernel) on the L = 65 536 shape, plus the ablation cycle table
export TMPDIR=/tmp; cd /tmp; cd $GRAFT_REPO_ROOT
python tools/attn64/ab.py cycles > gpurun_out/r06_attn64_ablation.log 2>&1
for v in base; do
  rm -rf gpurun_out/pmc_$v
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_$v -o pmc -- python tools/attn64/ab.py one $v > /dev/null 2> gpurun_out/pmc_$v.err
  DB=$(find gpurun_out/pmc_$v -name "pmc*results.db" | head -1)
  python tools/mfma_busy_summary.py "$DB" attn64_kernel > gpurun_out/r06_attn64_${v}_mfma_busy_pmc.json
  rm -rf gpurun_out/pmc_$v
done
rm -rf gpurun_out/pmc_old
DM4D_ATTN64=0 timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_old -o pmc -- python tools/attn64/ab.py one base > /dev/null 2> gpurun_out/pmc_old.err
DB=$(find gpurun_out/pmc_old -name "pmc*results.db" | head -1)
python tools/mfma_busy_summary.py "$DB" attn_kernel > gpurun_out/r06_attn8w_mfma_busy_pmc.json
rm -rf gpurun_out/pmc_old
cat gpurun_out/r06_attn64_ablation.log gpurun_out/r06_attn64_base_mfma_busy_pmc.json gpurun_out/r06_attn8w_mfma_busy_pmc.json
