#!/bin/bash
# The GPU calls of round 4, one parameterised script:  gpurun -- 'bash tools/dev/r04_gpu.sh <stage> [args]'
# Every stage writes under gpurun_out/r04_<stage>*; summaries worth keeping are copied to profiles/ by hand.
set -u
stage=${1:-parity1}; shift || true
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
bench_line() {  # $1 = json file, $2 = label
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d["kernel_breakdown_one_step"]
    print(sys.argv[2], d["value"], "lat/s", d["ms_per_step"], "ms; attn frac", d["roofline"]["frac"], "linear", k["linear"]["ms"], "conv", k["conv3x3"]["ms"],
          "gn", k["groupnorm"]["ms"], "ln", k["layernorm"]["ms"], "attn", k["attention"]["ms"])
except Exception as e:
    print(sys.argv[2], "bench line unreadable:", e)
PY
}
case $stage in
  parity1)  # first contact of the parity precision: new kernels, tiny models, the judged fixtures; then the fast path's regression check
    timeout 600 python tests/opcheck.py par_ > $out/r04_opcheck_par.log 2>&1; tail -50 $out/r04_opcheck_par.log
    timeout 600 python tests/modelcheck.py par_unet_spatial par_unet_temporal par_unet_2d par_vae par_pipeline > $out/r04_modelcheck_par_tiny.log 2>&1; grep -v "^    \[" $out/r04_modelcheck_par_tiny.log | tail -30
    timeout 900 python tests/modelcheck.py par_unet_sd21 par_vae_sd par_demo3d unet_sd21_128 > $out/r04_modelcheck_par_sd.log 2>&1; tail -30 $out/r04_modelcheck_par_sd.log
    timeout 600 python tests/opcheck.py gemm > $out/r04_opcheck_gemm.log 2>&1; tail -3 $out/r04_opcheck_gemm.log
    timeout 600 python tests/opcheck.py conv > $out/r04_opcheck_conv.log 2>&1; tail -3 $out/r04_opcheck_conv.log
    timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-grid-secondary --no-vae > $out/r04_bench_parity1.json 2> $out/r04_bench_parity1.err
    bench_line $out/r04_bench_parity1.json "fast path after the epilogue change:"
    ;;
  *) echo "unknown stage $stage"; exit 2;;
esac
