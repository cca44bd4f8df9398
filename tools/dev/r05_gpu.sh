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
  fp16b)  # second contact: vectorised fp32-in norms, phase-decomposed upsampling, direct operand out, the new epilogue loop; the F = 24 parity
          # case against its fp32 fixture; the driver command; then A/B of two experiments of the FAST path (variant libraries built on the
          # build host: tools/dev/build_variant.sh csouter gemm -DSTRIP2_CS_OUTER / ffabl ff_fused -DFF_ABLATE_H_ROUNDTRIP)
    timeout 900 python tests/opcheck.py h16_ > $out/r05_b_opcheck_h16.log 2>&1; quiet $out/r05_b_opcheck_h16.log | grep -v "^PASS" | tail -30
    timeout 1200 python tests/modelcheck.py fp16_unet_spatial fp16_vae fp16_pipeline_spatial fp16_unet_sd21 fp16_vae_sd fp16_demo3d fp16_multiround \
        fp16_unet_frame fp16_pipeline_shard par_unet_sd21_72x40_f24 par_unet_spatial par_pipeline_spatial > $out/r05_b_modelcheck.log 2>&1
    quiet $out/r05_b_modelcheck.log 600 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp" | tail -30
    ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench_b.json 2> $out/r05_bench_b.err ) 2> $out/r05_bench_b.time; tail -3 $out/r05_bench_b.time
    bench_line $out/r05_bench_b.json "driver command:"
    python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_bench_b.json"))
    for k in ("encode", "decode"):
        print("vae", k, d["secondary"]["vae"][k]["ms_per_image"], {f: (v["ms_per_image"], v["roofline_frac"]) for f, v in d["secondary"]["vae"][k].get("kernel_breakdown", {}).items()})
    t = d["secondary"]["tolerance_mode"]["kernel_breakdown_one_step"]
    print("tolerance levels:", {k: (v["ms"], v["roofline_frac"]) for k, v in t.items() if "." in k and k.split(".")[0] in ("linear", "conv3x3", "groupnorm", "layernorm", "split")})
except Exception as e:
    print("bench b unreadable:", e)
PY
    ab="--steps 9 --warmup 3 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
    cp diffuman4d_amd/libdm4d.so /tmp/cur.so
    for rep in 1 2; do for v in cur csouter ffabl; do
      if [ $v = cur ]; then cp /tmp/cur.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
      timeout 300 python bench.py $ab > $out/r05_ab_${v}_$rep.json 2>/dev/null; bench_line $out/r05_ab_${v}_$rep.json "A/B $v rep $rep:"
    done; done
    cp tools/dev/libdm4d_csouter.so diffuman4d_amd/libdm4d.so
    timeout 600 python tests/opcheck.py conv > $out/r05_ab_csouter_opcheck_conv.log 2>&1; tail -3 $out/r05_ab_csouter_opcheck_conv.log
    cp /tmp/cur.so diffuman4d_amd/libdm4d.so
    ;;
  fp16c)  # after the fix of the phase kernels' fp32 output offset: the fp16 precision again, then the driver command
    timeout 900 python tests/opcheck.py h16_ conv_up2x > $out/r05_c_opcheck_h16.log 2>&1; quiet $out/r05_c_opcheck_h16.log | grep -v "^PASS" | tail -30
    timeout 1200 python tests/modelcheck.py fp16_ > $out/r05_c_modelcheck_fp16.log 2>&1
    quiet $out/r05_c_modelcheck_fp16.log 600 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|^    \[golden\|^    \[pipeline" | tail -40
    ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench_c.json 2> $out/r05_bench_c.err ) 2> $out/r05_bench_c.time; tail -3 $out/r05_bench_c.time
    bench_line $out/r05_bench_c.json "driver command:"
    python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_bench_c.json"))
    t = d["secondary"]["tolerance_mode"]["kernel_breakdown_one_step"]
    print("tolerance levels:", {k: (v["ms"], v["roofline_frac"]) for k, v in t.items() if "." in k and k.split(".")[0] in ("linear", "conv3x3", "attention", "groupnorm", "layernorm", "split")})
except Exception as e:
    print("bench c unreadable:", e)
PY
    ;;
  dot2)  # A/B: attention row sums from the packed probabilities (v_dot2c against (1, 1)) -- tools/dev/build_variant.sh dot2 attention -DATTN_DOT2_SUM
    cp diffuman4d_amd/libdm4d.so /tmp/cur.so
    cp tools/dev/libdm4d_dot2.so diffuman4d_amd/libdm4d.so
    timeout 900 python tests/opcheck.py attn h16_attn > $out/r05_dot2_opcheck_attn.log 2>&1; quiet $out/r05_dot2_opcheck_attn.log | grep -v "^PASS" | tail; grep -c "^PASS" $out/r05_dot2_opcheck_attn.log
    grep "judged\|65536\|attn_qs_2d \|attn_qs_3d " $out/r05_dot2_opcheck_attn.log | cut -c1-120
    cp /tmp/cur.so diffuman4d_amd/libdm4d.so
    timeout 600 python tests/opcheck.py attn_qs_judged attn_qs_128sq attn_qs_2d attn_qs_3d h16_attn_judged > $out/r05_dot2_opcheck_attn_base.log 2>&1; grep "judged\|65536\|attn_qs_2d \|attn_qs_3d " $out/r05_dot2_opcheck_attn_base.log | cut -c1-120
    ab="--steps 9 --warmup 3 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-latent128"
    for rep in 1 2; do for v in cur dot2; do
      if [ $v = cur ]; then cp /tmp/cur.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
      timeout 300 python bench.py $ab > $out/r05_dot2_${v}_$rep.json 2>/dev/null; bench_line $out/r05_dot2_${v}_$rep.json "A/B $v rep $rep:"
    done; done
    for v in cur dot2; do
      if [ $v = cur ]; then cp /tmp/cur.so diffuman4d_amd/libdm4d.so; else cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; fi
      timeout 300 python bench.py --latent 128x128 --steps 2 --warmup 1 --task-streams 1 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128 > $out/r05_dot2_${v}_128.json 2>/dev/null
      bench_line $out/r05_dot2_${v}_128.json "A/B $v 128x128:"
    done
    cp /tmp/cur.so diffuman4d_amd/libdm4d.so
    ;;
  gnraw)  # the GroupNorm that also emits the shortcut operand; then task streams 3 / 4 / 6 in both precisions
    timeout 600 python tests/opcheck.py h16_gn h16_conv_up2x gn_ > $out/r05_gnraw_opcheck.log 2>&1; quiet $out/r05_gnraw_opcheck.log | grep -v "^PASS" | tail
    timeout 900 python tests/modelcheck.py fp16_unet_spatial fp16_vae fp16_pipeline_spatial fp16_unet_sd21_72x40_f16 fp16_demo3d fp16_vae_sd unet_spatial pipeline_spatial \
        > $out/r05_gnraw_modelcheck.log 2>&1; quiet $out/r05_gnraw_modelcheck.log 500 | grep "^PASS\|^FAIL\|^ERROR\|modelcheck:" | tail -12
    ab="--steps 12 --warmup 3 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-latent128"
    for rep in 1 2; do for ts in 3 4 6; do
      timeout 300 python bench.py $ab --task-streams $ts > $out/r05_ts${ts}_$rep.json 2>/dev/null; bench_line $out/r05_ts${ts}_$rep.json "task streams $ts rep $rep:"
    done; done
    ;;
  final1)  # the whole GPU suite, smoke(), the driver command on the final tree
    ( time timeout 2400 python -m pytest tests -m gpu -q > $out/r05_pytest_gpu.log 2>&1 ) 2> $out/r05_pytest_gpu.time; tail -5 $out/r05_pytest_gpu.log; tail -3 $out/r05_pytest_gpu.time
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/r05_smoke.log 2>&1; tail -2 $out/r05_smoke.log | cut -c1-1500
    ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench.json 2> $out/r05_bench.err ) 2> $out/r05_bench.time; tail -3 $out/r05_bench.time
    bench_line $out/r05_bench.json "driver command:"
    ;;
  final2)  # the driver command once more on the final tree (bench.py gained the tolerance mode's evidence fields after final1), then the records: rocprofv3 kernel tables of the three precisions, PMC passes of the fast one, BASELINE configs[4], the CLI path in the fp16 precision
    ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench.json 2> $out/r05_bench.err ) 2> $out/r05_bench.time; tail -3 $out/r05_bench.time
    bench_line $out/r05_bench.json "driver command:"
    bash tools/profile_bench.sh r05 fast > $out/r05_profile.log 2>&1; tail -14 $out/r05_profile.log | cut -c1-220
    bash tools/profile_bench.sh r05_fp16 fp16 stats > $out/r05_profile_fp16.log 2>&1; tail -12 $out/r05_profile_fp16.log | cut -c1-200
    bash tools/profile_bench.sh r05_parity parity stats > $out/r05_profile_parity.log 2>&1; tail -8 $out/r05_profile_parity.log | cut -c1-200
    timeout 600 python bench.py --config5 --steps 12 --warmup 3 --no-vae > $out/r05_bench_config5.json 2> $out/r05_bench_config5.err; bench_line $out/r05_bench_config5.json "config5 (48 x 225, sliding_default):"
    timeout 600 python bench.py --config5 --precision fp16 --steps 12 --warmup 3 --no-vae > $out/r05_bench_config5_fp16.json 2>/dev/null; bench_line $out/r05_bench_config5_fp16.json "config5, fp16 precision:"
    timeout 900 python tools/e2e_demo.py --exp demo_4d --fast-vae --prune --writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 3 \
        sampler.plucker_on_device=true data.plucker=cameras model.precision=fp16 > $out/r05_e2e_demo4d_fp16.json 2> $out/r05_e2e_demo4d_fp16.err
    cat $out/r05_e2e_demo4d_fp16.json | cut -c1-600; tail -2 $out/r05_e2e_demo4d_fp16.err | cut -c1-300
    ;;
  batch)  # tasks of a round stacked along M (bench.py --task-batch k: upload_plan(copies=k), bit-identical latents) against task streams of single tasks
    ab="--steps 12 --warmup 6 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
    for rep in 1 2; do
      for cfg in "1 3" "2 1" "2 2" "3 1" "3 2" "6 1"; do set -- $cfg
        timeout 300 python bench.py $ab --task-batch $1 --task-streams $2 > $out/r05_batch_b$1_s$2_$rep.json 2>/dev/null; bench_line $out/r05_batch_b$1_s$2_$rep.json "task-batch $1 x streams $2 rep $rep:"
      done
    done
    ;;
  stack)  # task stacks as a runner feature: bitwise checks, then defaults (streams x batch) by A/B on the bench and on the CLI path
    timeout 900 python tests/modelcheck.py task_stack fp16_task_stack par_task_stack > $out/r05_modelcheck_task_stack.log 2>&1; grep -c "^PASS" $out/r05_modelcheck_task_stack.log; grep "^FAIL\|^ERROR\|modelcheck:" $out/r05_modelcheck_task_stack.log | cut -c1-300
    timeout 900 python -m pytest tests/test_e2e_gpu.py -q -k "task_stacks" > $out/r05_pytest_task_stack.log 2>&1; tail -3 $out/r05_pytest_task_stack.log | cut -c1-300
    ab="--steps 24 --warmup 12 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-precision --no-tolerance-mode --no-latent128"
    for rep in 1 2; do
      for cfg in "1 3" "2 2" "3 2" "2 3" "4 2"; do set -- $cfg
        timeout 300 python bench.py $ab --task-batch $1 --task-streams $2 > $out/r05_stack_b$1_s$2_$rep.json 2>$out/r05_stack_err.log || tail -5 $out/r05_stack_err.log
        bench_line $out/r05_stack_b$1_s$2_$rep.json "batch $1 x streams $2 rep $rep:"
      done
    done
    for cfg in "1 3" "2 2" "3 2"; do set -- $cfg
      timeout 300 python bench.py $ab --precision fp16 --task-batch $1 --task-streams $2 > $out/r05_stack_fp16_b$1_s$2.json 2>$out/r05_stack_err.log || tail -5 $out/r05_stack_err.log
      bench_line $out/r05_stack_fp16_b$1_s$2.json "fp16 precision, batch $1 x streams $2:"
    done
    for cfg in "1 3" "2 2" "3 2"; do set -- $cfg
      timeout 600 python tools/e2e_demo.py --exp demo_4d_tiny --fast-vae --prune --writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 4 \
          --task-batch $1 --gpu-streams $2 sampler.plucker_on_device=true data.plucker=cameras > $out/r05_e2e_tiny_b$1_s$2.json 2> $out/r05_e2e_tiny.err || tail -5 $out/r05_e2e_tiny.err
      echo "e2e demo_4d_tiny batch $1 x streams $2:"; cut -c1-500 $out/r05_e2e_tiny_b$1_s$2.json
    done
    ;;
  final3)  # after the runner defaults became 2 streams of 2-task stacks: a canary, the driver command, the profile records on the new launch
           # shapes, then the GPU tests of everything the change touched (pipeline / sampler / runner / bench) and smoke()
    timeout 600 python -m pytest tests/test_bench_gpu.py -q -x -k "default_task_streams" > $out/r05_pytest_canary.log 2>&1 || { tail -30 $out/r05_pytest_canary.log | cut -c1-300; exit 1; }
    tail -1 $out/r05_pytest_canary.log
    ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench.json 2> $out/r05_bench.err ) 2> $out/r05_bench.time; tail -3 $out/r05_bench.time
    bench_line $out/r05_bench.json "driver command:"; tail -3 $out/r05_bench.err | cut -c1-300
    bash tools/profile_bench.sh r05b fast > $out/r05b_profile.log 2>&1; tail -14 $out/r05b_profile.log | cut -c1-220
    ( time timeout 1500 python -m pytest tests/test_bench_gpu.py tests/test_e2e_gpu.py tests/test_reference_protocol_gpu.py tests/test_entry_gpu.py tests/test_model_gpu.py -q \
        -k "not default_task_streams and (bench or e2e or reference_protocol or entry or pipeline or golden or task_stack or multiround)" > $out/r05_pytest_gpu_part2.log 2>&1 ) 2> $out/r05_pytest_gpu_part2.time
    tail -5 $out/r05_pytest_gpu_part2.log | cut -c1-300; tail -3 $out/r05_pytest_gpu_part2.time
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/r05_smoke.log 2>&1; tail -2 $out/r05_smoke.log | cut -c1-600
    ;;
  final4)  # the multi-round job in stacks against its task-by-task fixture, the UNet-call distance against the input state, the driver command
    timeout 900 python tests/modelcheck.py multiround par_multiround fp16_multiround > $out/r05_modelcheck_multiround_stacks.log 2>&1; grep "^PASS\|^FAIL\|^ERROR\|multi-round" $out/r05_modelcheck_multiround_stacks.log | cut -c1-330
    timeout 300 python tools/dev/parity_state_probe.py > $out/r05_parity_state_probe.log 2>&1; grep -v "^/opt\|warn" $out/r05_parity_state_probe.log | tail -12
    ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench.json 2> $out/r05_bench.err ) 2> $out/r05_bench.time; tail -3 $out/r05_bench.time
    bench_line $out/r05_bench.json "driver command:"; tail -3 $out/r05_bench.err | cut -c1-300
    ;;
  final5)  # the driver command with the stack-of-2 PMC records in profiles/ (roofline.traffic), then the CLI path on a 48 x 64 grid
           # (64 + 44 + 64 tasks) with the new runner defaults and with round 4's (the new ones FIRST: whatever runs first meets the colder chip)
    ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_bench.json 2> $out/r05_bench.err ) 2> $out/r05_bench.time; tail -3 $out/r05_bench.time
    bench_line $out/r05_bench.json "driver command:"; tail -2 $out/r05_bench.err | cut -c1-300
    for cfg in "2 2" "1 3"; do set -- $cfg
      timeout 230 python tools/e2e_demo.py --exp demo_4d --fast-vae --prune --writers 2 --device-results --writer-processes 12 --host-threads 8 --depth 4 \
          --task-batch $1 --gpu-streams $2 sampler.plucker_on_device=true data.plucker=cameras "sampler.tem_label_range=[0,64,1]" > $out/r05_e2e_64fr_b$1_s$2.json 2> $out/r05_e2e_64fr.err || tail -3 $out/r05_e2e_64fr.err | cut -c1-300
      echo "e2e 48 x 64 grid, batch $1 x streams $2:"; cut -c1-520 $out/r05_e2e_64fr_b$1_s$2.json
    done
    ;;
  final6)  # the driver command's GPU part with the final runner defaults (3 streams of 2-task stacks); the CPU baseline / parity objects of the
           # full command do not depend on the defaults (final5's record has them) and the round's GPU budget ends here
    timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-precision --no-latent128 > $out/r05_bench_default_3x2.json 2> $out/r05_bench_default_3x2.err
    bench_line $out/r05_bench_default_3x2.json "default (3 streams x 2-task stacks), no CPU legs:"; tail -2 $out/r05_bench_default_3x2.err | cut -c1-300
    ;;
  *) echo "unknown stage $stage"; exit 2;;
esac
