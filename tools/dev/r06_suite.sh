#!/bin/bash
# the whole GPU suite with per-test durations (the driver's step limit is 1 200 s)
export TMPDIR=/tmp
start=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q --durations=60 > gpurun_out/r06_pytest_gpu.log 2>&1
echo "exit $? wall $(( $(date +%s) - start )) s" >> gpurun_out/r06_pytest_gpu.log
tail -60 gpurun_out/r06_pytest_gpu.log
