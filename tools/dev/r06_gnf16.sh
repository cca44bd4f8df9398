#!/bin/bash
# fp16 precision: conv1 -> fp16 -> norm2 (dm4d_groupnorm_nhwc_f16_f16; DM4D_H16_CONV1_F16=0: fp32 in between): parity cases, what it does to the
# whole-task distances, bench A/B inside one call
export TMPDIR=/tmp
{
echo "=== opcheck"; timeout 900 python tests/opcheck.py h16_gn h16_conv_l0 h16_conv_9x5 2>&1 | grep -E "PASS|FAIL|ERROR|opcheck:|Error"
for a in 0 1; do
echo "=== fp16 model cases, DM4D_H16_CONV1_F16=$a"; DM4D_H16_CONV1_F16=$a timeout 1200 python tests/modelcheck.py fp16_unet_sd21 fp16_golden_spatial fp16_golden_temporal_v fp16_golden_pose fp16_demo3d fp16_demo4dtiny fp16_multiround 2>&1 | grep -E "PASS|FAIL|ERROR|modelcheck:" | cut -c1-120
done
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
for a in 0 1; do
  DM4D_H16_CONV1_F16=$a timeout 600 python bench.py $Q --precision fp16 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown_one_step',{})
print('round $round fp16 conv1_f16=$a', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'conv', kb.get('conv3x3',{}).get('ms'), 'groupnorm', kb.get('groupnorm',{}).get('ms'), 'parity', json.dumps(d.get('parity',{}))[:300])
"
done; done
} > gpurun_out/r06_gnf16.log 2>&1
cat gpurun_out/r06_gnf16.log
