#!/bin/bash
# stacks of four end to end (the loop of r06_e2e2.sh whose argument order argparse refused), then the GEMM / conv configuration sweep on
# the launches of a 2-task stack (CFG batch 64 / 96): is choose_cfg, tuned on single tasks, still the best per shape?
export TMPDIR=/tmp
P="sampler.plucker_on_device=true data.plucker=cameras"
C="--writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3"
for cfg in "3 4" "2 4"; do
  set -- $cfg
  timeout 900 python tools/e2e_demo.py --exp demo_4d --fast-vae --prune $C --gpu-streams $1 --task-batch $2 $P model.precision=fast > gpurun_out/r06b_e2e_demo_4d_fast_s$1_b$2.json 2> gpurun_out/r06b_e2e_fast_s$1_b$2.err
  cut -c1-700 gpurun_out/r06b_e2e_demo_4d_fast_s$1_b$2.json
done
timeout 900 python tools/gemm_tune.py 64,96 > gpurun_out/r06_gemm_tune_stacks.log 2>&1; tail -3 gpurun_out/r06_gemm_tune_stacks.log | cut -c1-200
