#!/bin/bash
# ff_proj_fused_kernel: where a step's time goes -- timing-only builds (wrong results) with one ingredient removed each
# (tools/dev/build_variant.sh ffabl_<X> ff_fused -DFF_ABL_<X>): GEGLU VALU, weight DMA, product 1, product 2, the step barrier, the loop
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/base.so
{
for round in 1 2; do
for v in base VALU DMA P1 P2 BAR VALU_DMA VALU_DMA_BAR P1_P2 LOOP; do
  if [ $v = base ]; then cp /tmp/base.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_ffabl_$v.so diffuman4d_amd/libdm4d.so; fi
  echo "=== $v (round $round)"; timeout 300 python tests/opbench.py ffproj 2>&1 | grep "one launch"
done; done
cp /tmp/base.so diffuman4d_amd/libdm4d.so
} > gpurun_out/r06_ffabl.log 2>&1
cat gpurun_out/r06_ffabl.log
