#!/bin/bash
# fp16 precision: the stand-alone fp32 -> fp16 conversions (inputs of the down- / up-sampling convolutions) eight channels per thread
# (to_f16_vec8_kernel): parity, bench step
export TMPDIR=/tmp
{
echo "=== parity"; timeout 600 python tests/opcheck.py h16_split 2>&1 | grep -E "PASS|FAIL|ERROR|opcheck:|Error"
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
  timeout 600 python bench.py $Q --precision fp16 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown_one_step',{})
print('round $round fp16', 'ms_per_step', d['ms_per_step'], 'split', kb.get('split',{}), 'linear', kb.get('linear',{}).get('ms'), 'conv', kb.get('conv3x3',{}).get('ms'))
"
done
echo "=== fp16 model cases"; timeout 900 python tests/modelcheck.py fp16_unet_sd21_72x40_f16 fp16_demo3d fp16_vae_sd fp16_golden_pose 2>&1 | grep -E "PASS|FAIL|ERROR|modelcheck:" | cut -c1-130
} > gpurun_out/r06_split8.log 2>&1
cat gpurun_out/r06_split8.log
