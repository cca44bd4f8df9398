#!/bin/bash
# ablations of the Linear / strip-conv kernels: where does the time go (epilogue stores / epilogue / DMA / MFMA)?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c4; mkdir -p $O
cp diffuman4d_amd/libdm4d.so /tmp/cur.so
for round in 1 2; do for v in base abl1 abl2 abl3 abl4; do
  cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
  echo "=== $v (round $round)" >> $O/ablate.log; timeout 300 python tests/opbench.py 2>&1 | grep "^gemm\|^conv" >> $O/ablate.log
done; done
cp /tmp/cur.so diffuman4d_amd/libdm4d.so
tail -3 $O/ablate.log
