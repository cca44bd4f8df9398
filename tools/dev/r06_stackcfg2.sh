#!/bin/bash
# second set of configuration choices for stacked launches (the build) against the first set (tools/dev/libdm4d_cfg1.so = the previous commit)
# (tools/dev/libdm4d_cfg1.so = gemm.hip + gemm_h16.hip with -DDM4D_NO_STACK_CFG): results must not change (every id is bit-identical),
# bench step A/B inside one call, fast and fp16
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/new.so
{
echo "=== parity (new)"; timeout 900 python tests/opcheck.py conv_batch_invariance gemm_ conv_l0 h16_conv_l0 h16_gemm_qkv 2>&1 | grep -E "FAIL|ERROR|opcheck:|Error"
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
for prec in fast fp16; do
for v in base new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_cfg1.so diffuman4d_amd/libdm4d.so; fi
  timeout 600 python bench.py $Q --precision $prec --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown_one_step',{})
print('round $round $prec $v', 'ms_per_step', d['ms_per_step'], 'linear', kb.get('linear',{}).get('ms'), 'linear.L0', kb.get('linear.L0',{}).get('ms'), 'conv', kb.get('conv3x3',{}).get('ms'), 'conv.L0', kb.get('conv3x3.L0',{}).get('ms'))
"
done; done; done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
echo "=== model cases (new)"; timeout 900 python tests/modelcheck.py task_stack unet_sd21_72x40_f16 fp16_task_stack fp16_unet_sd21_72x40_f16 2>&1 | grep -E "PASS|FAIL|ERROR|modelcheck:" | cut -c1-130
} > gpurun_out/r06_stackcfg2.log 2>&1
cat gpurun_out/r06_stackcfg2.log
