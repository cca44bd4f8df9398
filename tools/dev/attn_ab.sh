#!/bin/bash
# A/B of attention builds inside one gpurun call: parity (opcheck attn*) then the UNet attention shapes (opbench attn)
# usage: TAGS="new base" tools/dev/attn_ab.sh   (libraries: tools/dev/libdm4d_<tag>.so)
cp diffuman4d_amd/libdm4d.so /tmp/cur.so
for round in 1 2; do
for v in $TAGS; do
  cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
  if [ $round = 1 ]; then echo "=== $v opcheck"; timeout 300 python tests/opcheck.py attn 2>&1 | grep -v "amdgpu.ids" ; fi
  echo "=== $v opbench q_scaled (round $round)"; timeout 300 python tests/opbench.py attn 2>&1 | grep -v "amdgpu.ids"
  if [ $round = 1 ]; then echo "=== $v opbench scale in kernel"; DM4D_BENCH_QSCALED=0 timeout 300 python tests/opbench.py attn 2>&1 | grep -v "amdgpu.ids"; fi
done
done
cp /tmp/cur.so diffuman4d_amd/libdm4d.so
