#!/bin/bash
# the fp16 precision's fused block tail (ff_proj_fused_h16_kernel): parity cases, the whole-task fp16 cases, bench A/B inside one call
# (DM4D_FF_PROJ_FUSED=0: the four launches it replaces)
export TMPDIR=/tmp
{
echo "=== opcheck"; timeout 900 python tests/opcheck.py h16_ff_proj h16_gemm_resid ff_proj_fused_128 2>&1 | grep -E "PASS|FAIL|ERROR|opcheck:|Error"
echo "=== fp16 model cases"; timeout 1200 python tests/modelcheck.py fp16_unet_sd21 fp16_golden fp16_demo3d fp16_demo4dtiny fp16_multiround 2>&1 | grep -E "^\s+\[|PASS|FAIL|ERROR|modelcheck:"
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
for a in 0 1; do
  DM4D_FF_PROJ_FUSED=$a timeout 600 python bench.py $Q --precision fp16 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown_one_step',{})
print('round $round fp16 ff_proj_fused=$a', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'linear', kb.get('linear',{}).get('ms'), 'linear.L0', kb.get('linear.L0',{}).get('ms'), 'layernorm', kb.get('layernorm',{}).get('ms'))
"
done; done
} > gpurun_out/r06_ffh16.log 2>&1
cat gpurun_out/r06_ffh16.log
