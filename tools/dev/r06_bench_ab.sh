#!/bin/bash
# bench A/B inside one call: 8-wave attention kernel (DM4D_ATTN64=0) against attn64, fast and fp16 precisions
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
for round in 1 2; do
for prec in fast fp16; do
for a in 0 1; do
  DM4D_ATTN64=$a timeout 600 python bench.py $Q --precision $prec 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kb=d['secondary']['kernel_breakdown_one_step'] if 'kernel_breakdown_one_step' in d.get('secondary',{}) else d.get('kernel_breakdown_one_step',{})
print('round $round prec $prec attn64=$a', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'attn', kb.get('attention',{}).get('ms'), 'one-stack', d['roofline']['measured_in'][-60:])
"
done; done; done
