#!/bin/bash
# second sweep of the configuration ids on stacked launches: two passes of 12 launches per timing (the first sweep's +-5 % spread), then the
# same shapes as the fp16 precision launches them
export TMPDIR=/tmp
for pass in 1 2; do timeout 900 python tools/gemm_tune.py 64,96 12 > gpurun_out/r06_gemm_tune_stacks_p$pass.log 2>&1; tail -1 gpurun_out/r06_gemm_tune_stacks_p$pass.log; done
for pass in 1 2; do timeout 900 python tools/gemm_tune.py 64,96 12 f16 > gpurun_out/r06_gemm_tune_stacks_f16_p$pass.log 2>&1; tail -1 gpurun_out/r06_gemm_tune_stacks_f16_p$pass.log; done
