#!/bin/bash
# strip convolution: the zero row of the strip row's own parity (the build) against the single even zero row (tools/dev/libdm4d_zrow0.so):
# parity (bit-identical by construction: zeros), per-shape timing, bench step, LDS bank-conflict counter
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/new.so
{
echo "=== parity (new)"; timeout 900 python tests/opcheck.py conv h16_conv par_conv 2>&1 | grep -E "FAIL|ERROR|opcheck:|Error"
for round in 1 2; do
for v in base new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_zrow0.so diffuman4d_amd/libdm4d.so; fi
  echo "=== $v: opbench conv (round $round)"; timeout 300 python tests/opbench.py conv 2>&1 | grep -E "B64|vae" | cut -c1-120
done; done
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
for prec in fast fp16; do
for v in base new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_zrow0.so diffuman4d_amd/libdm4d.so; fi
  timeout 600 python bench.py $Q --precision $prec --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown_one_step',{})
print('round $round $prec $v', 'ms_per_step', d['ms_per_step'], 'conv', kb.get('conv3x3',{}).get('ms'), 'conv.L0', kb.get('conv3x3.L0',{}).get('ms'), 'conv.L1', kb.get('conv3x3.L1',{}).get('ms'), 'conv.L2', kb.get('conv3x3.L2',{}).get('ms'))
"
done; done; done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
echo "=== model cases (new)"; timeout 900 python tests/modelcheck.py unet_sd21_72x40_f16 vae_sd_576x320 fp16_vae_sd_576x320 task_stack_spatial 2>&1 | grep -E "PASS|FAIL|ERROR|modelcheck:" | cut -c1-130
} > gpurun_out/r06_zrow.log 2>&1
cat gpurun_out/r06_zrow.log
