#!/bin/bash
# LayerNorm with R rows per wave (the build: R = 2; tools/dev/libdm4d_ln4.so: R = 4) against one row per wave (libdm4d_ln1.so): parity
# (bitwise per row by construction), bench step, fast and fp16
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/new.so
{
echo "=== parity (R = 2)"; timeout 600 python tests/opcheck.py layernorm ln h16_layernorm h16_ln par_ln par_layernorm 2>&1 | grep -E "PASS|FAIL|ERROR|opcheck:|Error" | cut -c1-110
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
for prec in fast fp16; do
for v in ln1 new ln4; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  timeout 600 python bench.py $Q --precision $prec --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown_one_step',{})
print('round $round $prec $v', 'ms_per_step', d['ms_per_step'], 'layernorm', kb.get('layernorm',{}), 'L0', kb.get('layernorm.L0',{}).get('ms'))
"
done; done; done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
echo "=== model cases (R = 2)"; timeout 600 python tests/modelcheck.py unet_sd21_72x40_f16 task_stack_spatial fp16_unet_sd21_72x40_f16 par_unet_sd21_72x40_f16 2>&1 | grep -E "PASS|FAIL|ERROR|modelcheck:" | cut -c1-130
} > gpurun_out/r06_ln.log 2>&1
cat gpurun_out/r06_ln.log
