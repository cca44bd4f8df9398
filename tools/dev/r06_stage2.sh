#!/bin/bash
# round-6 records: the three tests that failed in the first full-suite run, the whole-task fp16 / parity cases, rocprofv3 profiles
export TMPDIR=/tmp
{
echo "=== re-run of the suite's failures"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_bench_gpu.py -m gpu -q -k "multistep_step or hybrid" --durations=5 2>&1 | tail -12
echo "=== whole-task cases"; timeout 900 python tests/modelcheck.py fp16_demo3d fp16_demo4dtiny fp16_multiround par_demo4dtiny demo4dtiny golden_pndm par_golden_pndm fp16_golden_pndm opreplay_unet_sd21 2>&1 | grep -E "^\s+\[|PASS|FAIL|ERROR|modelcheck:" 
} > gpurun_out/r06_modelcheck_fp16_tasks.log 2>&1
tail -30 gpurun_out/r06_modelcheck_fp16_tasks.log
bash tools/profile_bench.sh r06 fast > gpurun_out/r06_profile_fast.out 2>&1; tail -25 gpurun_out/r06_profile_fast.out
bash tools/profile_bench.sh r06_fp16 fp16 stats-only > gpurun_out/r06_profile_fp16.out 2>&1; tail -12 gpurun_out/r06_profile_fp16.out
