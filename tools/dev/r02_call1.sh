#!/bin/bash
# round 2, GPU call 1: validate everything new, baseline bench, cheap A/Bs (attention ring4 / young-half priority, GEMM epilogue)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c1; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?" >> $O/build.log
cp diffuman4d_amd/libdm4d.so /tmp/cur.so
( time timeout 900 python tests/opcheck.py ) > $O/opcheck.log 2>&1
( time timeout 1200 python tests/modelcheck.py ) > $O/modelcheck.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_model_gpu.py --deselect tests/test_ops_gpu.py ) > $O/pytest_rest.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 python tests/opbench.py > $O/opbench_base.log 2>&1
# attention variants: parity on the q-scaled entry (incl. the judged shapes), then the UNet shapes, two rounds
for v in ring4 prio ring4prio; do
  cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
  echo "=== $v opcheck attn_qs" >> $O/attn_ab.log; timeout 400 python tests/opcheck.py attn_qs 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/attn_ab.log
done
for round in 1 2; do for v in base ring4 prio ring4prio; do
  cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
  echo "=== $v opbench attn (round $round)" >> $O/attn_ab.log; timeout 300 python tests/opbench.py attn 2>&1 | grep -v amdgpu.ids >> $O/attn_ab.log
done; done
for round in 1 2; do for v in base epi0 epi2; do
  cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
  echo "=== $v opbench gemm (round $round)" >> $O/gemm_epi_ab.log; timeout 300 python tests/opbench.py 2>&1 | grep "^gemm" >> $O/gemm_epi_ab.log
done; done
cp tools/dev/libdm4d_epi2.so diffuman4d_amd/libdm4d.so
echo "=== epi2 opcheck gemm/conv" >> $O/gemm_epi_ab.log; (timeout 300 python tests/opcheck.py gemm; timeout 300 python tests/opcheck.py conv) 2>&1 | grep -v amdgpu.ids | grep -v "^PASS" >> $O/gemm_epi_ab.log
cp /tmp/cur.so diffuman4d_amd/libdm4d.so
tail -3 $O/opcheck.log $O/modelcheck.log $O/pytest_rest.log; cat $O/bench.json | head -c 1500
