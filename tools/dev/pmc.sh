cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in ${KERNELS:-conv_l1 attn_l1}; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d gpurun_out/pmc -o ${k}_a -- python tools/dev/one.py $k > /dev/null 2>gpurun_out/pmc_${k}_a.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 -d gpurun_out/pmc -o ${k}_b -- python tools/dev/one.py $k > /dev/null 2>gpurun_out/pmc_${k}_b.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d gpurun_out/pmc -o ${k}_c -- python tools/dev/one.py $k > /dev/null 2>gpurun_out/pmc_${k}_c.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc -o ${k}_d -- python tools/dev/one.py $k > /dev/null 2>gpurun_out/pmc_${k}_d.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TA_BUSY_avr -d gpurun_out/pmc -o ${k}_e -- python tools/dev/one.py $k > /dev/null 2>gpurun_out/pmc_${k}_e.err
done
ls -la gpurun_out/pmc; tail -3 gpurun_out/pmc_conv_l1_b.err
