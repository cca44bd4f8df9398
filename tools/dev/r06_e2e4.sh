#!/bin/bash
# the fast configuration end to end once more, on the FINAL tree (after the tile choices for stacked launches), both precisions
export TMPDIR=/tmp
P="sampler.plucker_on_device=true data.plucker=cameras"
C="--writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3"
for prec in fast fp16; do
  timeout 900 python tools/e2e_demo.py --exp demo_4d --fast-vae --prune $C $P model.precision=$prec > gpurun_out/r06c_e2e_demo_4d_${prec}.json 2> gpurun_out/r06c_e2e_${prec}.err
  cut -c1-700 gpurun_out/r06c_e2e_demo_4d_${prec}.json
done
