#!/bin/bash
# the CLI path end to end on the full 48 x 150 grid with the runner's defaults (3 streams of 2-task stacks),
# fast and fp16 precisions, fast (cached moments, lazy decode, pruned cond rows, device results) and strict (the reference's per-task
# work) configurations
export TMPDIR=/tmp
F="--fast-vae --prune --writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3 sampler.plucker_on_device=true data.plucker=cameras"
S="--writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3"
for prec in fast fp16; do
  timeout 900 python tools/e2e_demo.py --exp demo_4d $F model.precision=$prec > gpurun_out/r06_e2e_demo_4d_${prec}.json 2> gpurun_out/r06_e2e_${prec}.err
  cut -c1-700 gpurun_out/r06_e2e_demo_4d_${prec}.json; tail -1 gpurun_out/r06_e2e_${prec}.err | cut -c1-200
done
for prec in fast fp16; do
  timeout 1500 python tools/e2e_demo.py --exp demo_4d $S model.precision=$prec > gpurun_out/r06_e2e_demo_4d_${prec}_strict.json 2> gpurun_out/r06_e2e_${prec}_strict.err
  cut -c1-700 gpurun_out/r06_e2e_demo_4d_${prec}_strict.json; tail -1 gpurun_out/r06_e2e_${prec}_strict.err | cut -c1-200
done
