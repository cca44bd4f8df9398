#!/bin/bash
# fp16 precision: straight-line read-back of the epilogue for fp32 side inputs (the build) against the task-by-task loop
# (tools/dev/libdm4d_h16loop.so = gemm_h16.hip with -DDM4D_H16_EPI_LOOP): parity, bench step A/B inside one call
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/new.so
{
echo "=== parity (new)"; timeout 900 python tests/opcheck.py h16_gemm h16_conv h16_ff 2>&1 | grep -E "FAIL|ERROR|opcheck:|Error"
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
for v in loop new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_h16loop.so diffuman4d_amd/libdm4d.so; fi
  timeout 600 python bench.py $Q --precision fp16 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown_one_step',{})
print('round $round fp16 $v', 'ms_per_step', d['ms_per_step'], 'linear', kb.get('linear',{}).get('ms'), 'conv', kb.get('conv3x3',{}).get('ms'))
"
done; done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
echo "=== fp16 model cases (new)"; timeout 1200 python tests/modelcheck.py fp16_unet_sd21 fp16_golden_spatial fp16_demo3d fp16_demo4dtiny fp16_vae_sd fp16_task_stack fp16_unet_frame 2>&1 | grep -E "PASS|FAIL|ERROR|modelcheck:" | cut -c1-130
} > gpurun_out/r06_h16epi.log 2>&1
cat gpurun_out/r06_h16epi.log
