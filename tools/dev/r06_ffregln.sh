#!/bin/bash
# the fast precision's block tail with norm3 from the accumulators and h as the second product's initial accumulators (FF_REG_LN, the build)
# against round 3's form (tools/dev/libdm4d_fftile.so = -DFF_TILE_LN): parity, per launch, bench step, model cases
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/new.so
{
echo "=== parity (new)"; timeout 600 python tests/opcheck.py ff_proj_fused ff_fused h16_ff_proj 2>&1 | grep -E "PASS|FAIL|ERROR|opcheck:|Error"
for round in 1 2; do
for v in tile new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_fftile.so diffuman4d_amd/libdm4d.so; fi
  echo "=== $v: opbench (round $round)"; timeout 300 python tests/opbench.py ffproj 2>&1 | grep "one launch"
done; done
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
for v in tile new; do
  if [ $v = new ]; then cp /tmp/new.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_fftile.so diffuman4d_amd/libdm4d.so; fi
  timeout 600 python bench.py $Q 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d.get('kernel_breakdown_one_step',{})
print('round $round fast $v', 'ms_per_step', d['ms_per_step'], 'linear', kb.get('linear',{}).get('ms'), 'linear.L0', kb.get('linear.L0',{}).get('ms'), 'layernorm', kb.get('layernorm',{}).get('ms'))
"
done; done
cp /tmp/new.so diffuman4d_amd/libdm4d.so
echo "=== model cases (new)"; timeout 1200 python tests/modelcheck.py unet_sd21 golden_spatial golden_temporal_v demo3d_sd21 demo4dtiny multiround_sd21 opreplay_unet task_stack 2>&1 | grep -E "PASS|FAIL|ERROR|modelcheck:" | cut -c1-150
} > gpurun_out/r06_ffregln.log 2>&1
cat gpurun_out/r06_ffregln.log
