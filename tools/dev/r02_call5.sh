#!/bin/bash
# MODE-specialised epilogue: parity (all op cases, the bitwise ones included; model cases) and speed
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c5; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?" >> $O/build.log
( time timeout 900 python tests/opcheck.py ) > $O/opcheck.log 2>&1
timeout 300 python tests/opbench.py > $O/opbench.log 2>&1
( time timeout 1200 python tests/modelcheck.py ) > $O/modelcheck.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
for k in conv_l1 gemm_ff2_l0; do
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc -o ${k}_b -- python tools/dev/one.py $k > /dev/null 2>$O/pmc_${k}_b.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc -o ${k}_d -- python tools/dev/one.py $k > /dev/null 2>$O/pmc_${k}_d.err
  python tools/dev/pmc_report.py /tmp/pmc $k gemm_kernel conv_strip > $O/pmc_$k.txt 2>&1
done
grep -c PASS $O/opcheck.log; tail -2 $O/opcheck.log | head -1; tail -12 $O/modelcheck.log | head -3; head -c 400 $O/bench.json
