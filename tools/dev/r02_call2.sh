#!/bin/bash
# round 2, GPU call 2: fill-rate probe, GEMM/conv config sweep incl. the 16-wave 256x256 tiles, PMC of two Linear shapes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c2; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?" >> $O/build.log
timeout 300 tools/probes/fill_probe > $O/fill_probe.log 2>&1
timeout 900 python tools/gemm_tune.py > $O/gemm_tune.log 2>&1
timeout 600 python -m pytest tests/test_e2e_gpu.py -q -m gpu -k fp16 > $O/pytest_fp16.log 2>&1
timeout 120 python tests/opcheck.py gemm > $O/opcheck_gemm.log 2>&1
for cfg in 14 19; do python - <<PY >> $O/opcheck_gemm.log 2>&1
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from diffuman4d_amd.host import lib as L
import opcheck
L.load().dm4d_tune_set_gemm_config($cfg)
for n in ("gemm_big_tiles","gemm_geglu","gemm_nobias","gemm_rowbias","gemm_n64_tiles","gemm_split_a"):
    try:
        e,m,t=opcheck.run_case(n); print("cfg $cfg",n,"PASS" if e<=t else "FAIL",e)
    except Exception as ex: print("cfg $cfg",n,"ERR",ex)
PY
done
rocprofv3 -L > $O/counters.txt 2>&1
cd /tmp; cd "$GRAFT_REPO_ROOT"
for k in "gemm_qkv_l0" "gemm_qkv_l0 19" "gemm_ff1_l0" "gemm_ff1_l0 19" "conv_l1"; do
  tag=$(echo $k | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/pmc -o ${tag}_a -- python tools/dev/one.py $k > /dev/null 2>$O/pmc_${tag}_a.err
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum -d /tmp/pmc -o ${tag}_c -- python tools/dev/one.py $k > /dev/null 2>$O/pmc_${tag}_c.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc -o ${tag}_d -- python tools/dev/one.py $k > /dev/null 2>$O/pmc_${tag}_d.err
  python tools/dev/pmc_report.py /tmp/pmc $tag gemm_kernel conv_strip > $O/pmc_$tag.txt 2>&1
done
du -sh $O
tail -5 $O/fill_probe.log; tail -3 $O/gemm_tune.log; tail -3 $O/pytest_fp16.log
