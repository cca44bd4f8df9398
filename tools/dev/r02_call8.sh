#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c8; mkdir -p $O
cp diffuman4d_amd/libdm4d.so /tmp/cur.so
for v in swp swpnosgb; do
  cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
  echo "=== $v opcheck attn" >> $O/attn_swp.log; timeout 600 python tests/opcheck.py attn 2>&1 | grep -v amdgpu.ids | grep -v "^PASS" >> $O/attn_swp.log
done
for round in 1 2; do for v in swp0 swp swpnosgb swplead2 swplead6; do
  cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
  echo "=== $v opbench attn (round $round)" >> $O/attn_swp.log; timeout 300 python tests/opbench.py attn 2>&1 | grep -v amdgpu.ids >> $O/attn_swp.log
done; done
cp /tmp/cur.so diffuman4d_amd/libdm4d.so
cat $O/attn_swp.log | grep -v "^attn q_scaled" | head -80
