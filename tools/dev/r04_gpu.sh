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
  verify2)  # parity launches on their own kernel instantiations, the matched-oracle / multi-round / 128x128 cases, the new bench line
    timeout 600 python tests/opcheck.py par_ > $out/r04_v2_opcheck_par.log 2>&1; tail -2 $out/r04_v2_opcheck_par.log
    timeout 300 python tests/opcheck.py resize_aa > $out/r04_v2_opcheck_resize.log 2>&1; tail -4 $out/r04_v2_opcheck_resize.log
    timeout 900 python tests/modelcheck.py unet_spatial_matched unet_temporal_temb_matched pipeline_spatial_matched pipeline_temporal_v_matched \
        unet_sd21_72x40_f16_matched unet_sd21_72x40_f24_matched > $out/r04_v2_modelcheck_matched.log 2>&1; grep -v "^/opt" $out/r04_v2_modelcheck_matched.log | tail -14
    timeout 900 python tests/modelcheck.py multiround par_multiround par_unet_sd21_72x40 par_demo3d par_vae_sd > $out/r04_v2_modelcheck_par.log 2>&1; grep -v "^/opt" $out/r04_v2_modelcheck_par.log | tail -12
    ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r04_v2_bench_default.json 2> $out/r04_v2_bench_default.err ) 2> $out/r04_v2_bench_default.time; tail -3 $out/r04_v2_bench_default.time
    bench_line $out/r04_v2_bench_default.json "driver command:"
    python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04_v2_bench_default.json"))
    print("parity:", json.dumps(d.get("parity", {}).get("modes")), "| parity_precision:", json.dumps(d["secondary"].get("parity_precision")))
    print("latent128:", json.dumps(d["secondary"].get("latent128")))
    print("levels:", {k: (v["ms"], v["roofline_frac"]) for k, v in d["kernel_breakdown_one_step"].items() if "." in k})
    print("cpu_baseline:", d.get("cpu_baseline", {}).get("value"), "grid:", d["secondary"].get("grid", {}).get("latents_per_s"))
except Exception as e:
    print("default bench line unreadable:", e)
PY
    for rep in 1 2; do for ts in 2 3; do
      timeout 300 python bench.py --steps 8 --warmup 2 --task-streams $ts --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-latent128 > $out/r04_v2_bench_ts${ts}_$rep.json 2>/dev/null
      bench_line $out/r04_v2_bench_ts${ts}_$rep.json "task streams $ts rep $rep:"
    done; done
    timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > $out/r04_v2_pytest_e2e.log 2>&1; tail -3 $out/r04_v2_pytest_e2e.log
    ;;
  verify3)  # launch-by-launch replay of the fast precision, the matched-oracle band, the multi-round job; then the whole CLI path twice
    timeout 900 python tests/modelcheck.py opreplay_ > $out/r04_v3_modelcheck_opreplay.log 2>&1; grep -v "^/opt\|Denoising" $out/r04_v3_modelcheck_opreplay.log | cut -c1-1200 | tail -14
    timeout 900 python tests/modelcheck.py unet_spatial_matched unet_temporal_temb_matched pipeline_spatial_matched pipeline_temporal_v_matched \
        unet_sd21_72x40_f16_matched unet_sd21_72x40_f24_matched multiround par_multiround > $out/r04_v3_modelcheck_matched.log 2>&1
    grep -v "^/opt\|Denoising" $out/r04_v3_modelcheck_matched.log | cut -c1-400 | tail -24
    for m in strict fast; do
      if [ $m = fast ]; then fl="--fast-vae --prune"; else fl=""; fi
      timeout 1200 python tools/e2e_demo.py --exp demo_4d $fl --writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3 \
        --timeline $out/r04_e2e_demo4d_${m}_timeline.json sampler.plucker_on_device=true data.plucker=cameras > $out/r04_e2e_demo4d_$m.json 2> $out/r04_e2e_demo4d_$m.err
      cat $out/r04_e2e_demo4d_$m.json; tail -2 $out/r04_e2e_demo4d_$m.err | cut -c1-300
    done
    rm -f $out/r04_e2e_demo4d_*_timeline.json
    ;;
  verify4)  # UniPC / DEIS (kernel + reference-pipeline fixtures), parity precision against the reference pipeline's fixtures, pose encoder;
            # then the fast CLI path again with result packing kept on the device
    timeout 300 python tests/opcheck.py multistep_step par_multistep par_convd convd_ > $out/r04_v4_opcheck.log 2>&1; tail -8 $out/r04_v4_opcheck.log
    timeout 900 python tests/modelcheck.py golden_unipc golden_deis par_golden > $out/r04_v4_modelcheck_golden.log 2>&1
    grep -v "^/opt\|Denoising" $out/r04_v4_modelcheck_golden.log | cut -c1-400 | tail -32
    timeout 900 python tools/e2e_demo.py --exp demo_4d --fast-vae --prune --writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3 \
        sampler.plucker_on_device=true data.plucker=cameras > $out/r04_v4_e2e_demo4d_fast.json 2> $out/r04_v4_e2e_demo4d_fast.err
    cat $out/r04_v4_e2e_demo4d_fast.json; tail -2 $out/r04_v4_e2e_demo4d_fast.err | cut -c1-300
    ;;
  e2e5)  # where the fast CLI path loses its last 12 %: host-stage detail with 2 and 3 tasks in flight
    for gs in 2 3; do
      timeout 700 python tools/e2e_demo.py --exp demo_4d --fast-vae --prune --writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3 \
        --gpu-streams $gs --timeline $out/r04_v5_tl_$gs.json sampler.plucker_on_device=true data.plucker=cameras > $out/r04_v5_e2e_fast_gs$gs.json 2> $out/r04_v5_e2e_fast_gs$gs.err
      cat $out/r04_v5_e2e_fast_gs$gs.json
      python - $out/r04_v5_tl_$gs.json <<'PY'
import json, sys
ev = json.load(open(sys.argv[1]))
den = sorted((s, e) for n, s, e, th in ev if n == "denoise")
# union of the intervals in which NO worker is between the start of its window sweep and the end of its device wait = certain GPU idle
gpu = sorted((s, e) for n, s, e, th in ev if n in ("d.denoise_latents", "d.decode", "d.device_wait", "d.encode_scaled"))
end = max(e for _, _, e, _ in ev)
cov, cur = 0.0, None
for s, e in gpu:
    if cur is None or s > cur[1]:
        cov += (cur[1] - cur[0]) if cur else 0.0
        cur = [s, e]
    else:
        cur[1] = max(cur[1], e)
cov += (cur[1] - cur[0]) if cur else 0.0
print(f"  wall {end:.1f}s; some worker inside sweep/decode/wait/encode: {cov:.1f}s; first denoise starts at {den[0][0]:.2f}s")
rounds = {}
for n, s, e, th in ev:
    if n == "denoise":
        rounds.setdefault(th, []).append((s, e))
gaps = []
for th, iv in rounds.items():
    iv.sort()
    gaps += [(b[0] - a[1], a[1]) for a, b in zip(iv, iv[1:])]
gaps.sort(reverse=True)
print("  largest gaps between consecutive tasks of a worker (s, at):", [(round(g, 2), round(t, 1)) for g, t in gaps[:6]], "sum", round(sum(g for g, _ in gaps), 1))
PY
      rm -f $out/r04_v5_tl_$gs.json
    done
    ;;
  final)  # the record set of the final tree: the whole GPU suite, the driver's command, the strict CLI path with the default three streams
    ( time timeout 1700 python -m pytest tests -m gpu -x -q > $out/r04_pytest_gpu.log 2>&1 ) 2> $out/r04_pytest_gpu.time; tail -4 $out/r04_pytest_gpu.log; tail -3 $out/r04_pytest_gpu.time
    timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/r04_smoke.log 2>&1; tail -2 $out/r04_smoke.log | cut -c1-600
    ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r04_bench.json 2> $out/r04_bench.err ) 2> $out/r04_bench.time; tail -3 $out/r04_bench.time
    bench_line $out/r04_bench.json "driver command:"
    timeout 900 python tools/e2e_demo.py --exp demo_4d --writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3 \
        sampler.plucker_on_device=true data.plucker=cameras > $out/r04_e2e_demo4d_strict_gs3.json 2> $out/r04_e2e_demo4d_strict_gs3.err
    cat $out/r04_e2e_demo4d_strict_gs3.json
    ;;
  suite)  # the whole GPU suite (no -x: one trip reports everything), the matched demo_3d case with its ratios, the driver's command for the record
    ( time timeout 1700 python -m pytest tests -m gpu -q > $out/r04_pytest_gpu.log 2>&1 ) 2> $out/r04_pytest_gpu.time; tail -6 $out/r04_pytest_gpu.log | cut -c1-300; tail -3 $out/r04_pytest_gpu.time
    grep "demo_3d fast vs rounding-matched" $out/r04_pytest_gpu.log | cut -c1-300
    timeout 300 python tests/modelcheck.py demo3d_sd21_72x40_matched 2>&1 | grep "demo_3d\|PASS\|FAIL" | cut -c1-300 | tee $out/r04_modelcheck_demo3d_matched.log
    ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r04_bench.json 2> $out/r04_bench.err ) 2> $out/r04_bench.time; tail -3 $out/r04_bench.time
    bench_line $out/r04_bench.json "driver command:"
    ;;
  *) echo "unknown stage $stage"; exit 2;;
esac
