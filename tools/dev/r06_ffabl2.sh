#!/bin/bash
# ff_proj_fused_kernel, second look: what its prologue + epilogue (35 % of a launch, r06_ffabl.log) are made of -- timing-only builds without
# the steady loop, then without norm3 / the epilogue as well; the norm3 row loop unrolled (bit-identical); the fp16 kernel, whose norm3 comes
# from the accumulators and which stores no h
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/base.so
{
for round in 1 2; do
for v in base ffabl_LOOP ffabl_LOOP_LN ffabl_LOOP_EPI ffabl_LOOP_LN_EPI fflnu4 fflnu8; do
  if [ $v = base ]; then cp /tmp/base.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
  echo "=== $v (round $round)"; timeout 300 python tests/opbench.py ffproj 2>&1 | grep "one launch"
done
cp /tmp/base.so diffuman4d_amd/libdm4d.so
echo "=== fp16 kernel (round $round)"; timeout 300 python tests/opbench.py ffproj16 2>&1 | grep "ffproj16"
done
for v in fflnu4 fflnu8; do cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; echo "=== $v parity"; timeout 600 python tests/opcheck.py ff_proj_fused ff_fused 2>&1 | grep -E "FAIL|ERROR|opcheck:"; done
cp /tmp/base.so diffuman4d_amd/libdm4d.so
} > gpurun_out/r06_ffabl2.log 2>&1
cat gpurun_out/r06_ffabl2.log
