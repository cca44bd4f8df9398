#!/bin/bash
# The GPU calls of round 5, one parameterised script:  gpurun -- 'bash tools/dev/r05_gpu.sh <stage> [args]'
# Every stage writes under gpurun_out/r05_<stage>*; summaries worth keeping are copied to profiles/ by hand.
set -u
stage=${1:-fp16a}; shift || true
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
bench_line() {  # $1 = json file, $2 = label
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); k = d["kernel_breakdown_one_step"]
    print(sys.argv[2], d["value"], "lat/s", d["ms_per_step"], "ms; attn frac", d["roofline"]["frac"], "linear", k["linear"]["ms"], "conv", k["conv3x3"]["ms"],
          "gn", k["groupnorm"]["ms"], "ln", k["layernorm"]["ms"], "attn", k["attention"]["ms"], "split", k.get("split", {}).get("ms"))
    for key in ("tolerance_mode", "parity_precision"):
        t = d.get("secondary", {}).get(key)
        if t:
            kk = t["kernel_breakdown_one_step"]
            print("  ", key, t["ms_per_step"], "ms (", t["task_streams"], "streams ),", t.get("ms_per_step_one_task"), "ms one task;",
                  {f: (kk[f]["ms"], kk[f]["roofline_frac"]) for f in ("linear", "conv3x3", "attention", "groupnorm", "layernorm", "split") if f in kk})
    if "parity" in d:
        print("   parity.modes:", json.dumps(d["parity"]["modes"]))
    if "cpu_baseline" in d:
        print("   cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"].get("seconds_f16_all"), d["cpu_baseline"].get("seconds_f24"))
except Exception as e:
    print(sys.argv[2], "bench line unreadable:", e)
PY
}
quiet() { grep -v "^/opt\|Denoising\|warnings.warn\|^$" "$1" | cut -c1-${2:-400}; }
case $stage in
  fp16a)  # first contact of the fp16 precision: its kernels, the models, the judged fixtures; the closed parity-precision holes; then bench
    timeout 900 python tests/opcheck.py h16_ par_attn_judged > $out/r05_opcheck_h16.log 2>&1; quiet $out/r05_opcheck_h16.log | grep -v "^PASS" | tail -40
    timeout 900 python tests/modelcheck.py fp16_unet_spatial fp16_unet_temporal fp16_unet_2d fp16_unet_pose fp16_vae fp16_pipeline fp16_golden fp16_unet_frame \
        > $out/r05_modelcheck_fp16_tiny.log 2>&1; quiet $out/r05_modelcheck_fp16_tiny.log 300 | grep -v "^    \[" | tail -40
    timeout 1200 python tests/modelcheck.py fp16_unet_sd21 fp16_vae_sd fp16_demo3d fp16_multiround > $out/r05_modelcheck_fp16_sd.log 2>&1; quiet $out/r05_modelcheck_fp16_sd.log 600 | tail -24
    timeout 900 python tests/modelcheck.py par_unet_sd21_72x40_f24 par_unet_frame par_pipeline_shard par_pipeline_prune > $out/r05_modelcheck_par_holes.log 2>&1
    quiet $out/r05_modelcheck_par_holes.log 600 | tail -12
    ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench_a.json 2> $out/r05_bench_a.err ) 2> $out/r05_bench_a.time; tail -3 $out/r05_bench_a.time
    bench_line $out/r05_bench_a.json "driver command:"
    tail -5 $out/r05_bench_a.err | cut -c1-300
    ;;
  *) echo "unknown stage $stage"; exit 2;;
esac
