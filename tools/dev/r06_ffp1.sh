#!/bin/bash
# ff_fused experiment: weight fragments of product 1 read D k steps ahead by hand (DM4D_FF_P1_DEPTH; tools/dev/libdm4d_ffp1d<D>.so built
# here with tools/dev/build_variant.sh) against the shipped schedule: bit-identity (the opcheck cases assert it against the unfused forms),
# per-launch time, and a bench step for the best
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/base.so
{
for v in base ffp1d3 ffp1d4 ffp1d6; do
  if [ $v = base ]; then cp /tmp/base.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  echo "=== $v: parity"; timeout 600 python tests/opcheck.py ff_fused ff_proj_fused h16_ff_proj 2>&1 | grep -E "FAIL|ERROR|opcheck:|Error"
done
for round in 1 2; do
for v in base ffp1d3 ffp1d4 ffp1d6; do
  if [ $v = base ]; then cp /tmp/base.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  echo "=== $v: opbench (round $round)"; timeout 300 python tests/opbench.py ffproj 2>&1 | grep "one launch"
done; done
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
for v in base ffp1d4 ffp1d6; do
  if [ $v = base ]; then cp /tmp/base.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  timeout 600 python bench.py $Q 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown_one_step',{})
print('round $round fast $v', 'ms_per_step', d['ms_per_step'], 'linear', kb.get('linear',{}).get('ms'), 'linear.L0', kb.get('linear.L0',{}).get('ms'))
"
done; done
cp /tmp/base.so diffuman4d_amd/libdm4d.so
} > gpurun_out/r06_ffp1.log 2>&1
cat gpurun_out/r06_ffp1.log
