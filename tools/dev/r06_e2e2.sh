#!/bin/bash
# the CLI path end to end on the full 48 x 150 grid, final tree of round 6 (runner defaults: 3 streams of 2-task stacks).  Fast configuration
# (cached moments, lazy decode, pruned cond rows) in both precisions, and the STRICT one (the reference's per-task VAE work, nothing pruned) with
# the conditioning prepared as in round 4's strict record (Pluecker maps from the cameras on the device) -- r06_e2e.sh's strict runs built the
# Pluecker maps on the host (data.plucker default), which is 1 700 s of host stage time and not what round 4's 17.6 latents/s measured
export TMPDIR=/tmp
P="sampler.plucker_on_device=true data.plucker=cameras"
C="--writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3"
for prec in fast fp16; do
  timeout 900 python tools/e2e_demo.py --exp demo_4d --fast-vae --prune $C $P model.precision=$prec > gpurun_out/r06b_e2e_demo_4d_${prec}.json 2> gpurun_out/r06b_e2e_${prec}.err
  cut -c1-700 gpurun_out/r06b_e2e_demo_4d_${prec}.json
done
for prec in fast fp16; do
  timeout 1200 python tools/e2e_demo.py --exp demo_4d $C $P model.precision=$prec > gpurun_out/r06b_e2e_demo_4d_${prec}_strict.json 2> gpurun_out/r06b_e2e_${prec}_strict.err
  cut -c1-700 gpurun_out/r06b_e2e_demo_4d_${prec}_strict.json
done
# stacks of four (bench A/B of the same tree, profiles/r06_task_batch_streams.log: 3 x 4 +0.9 %, 2 x 4 +1.8 % over the default 3 x 2): end to end
for cfg in "3 4" "2 4"; do
  set -- $cfg
  timeout 900 python tools/e2e_demo.py --exp demo_4d --fast-vae --prune $C --gpu-streams $1 --task-batch $2 $P model.precision=fast > gpurun_out/r06b_e2e_demo_4d_fast_s$1_b$2.json 2> gpurun_out/r06b_e2e_fast_s$1_b$2.err
  cut -c1-700 gpurun_out/r06b_e2e_demo_4d_fast_s$1_b$2.json
done
