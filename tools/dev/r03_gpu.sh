#!/bin/bash
# The GPU calls of round 3, one parameterised script:  gpurun -- 'bash tools/dev/r03_gpu.sh <stage> [args]'
# Every stage writes under gpurun_out/r03_<stage>*; summaries worth keeping are copied to profiles/ by hand.
set -u
stage=${1:-baseline}; shift || true
out=gpurun_out; mkdir -p $out
export PYTHONUNBUFFERED=1
case $stage in
  baseline)  # GPU test suite on the cleaned tree, the driver's bench command, e2e demo_4d_tiny old vs new host pipeline
    ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $out/r03_pytest_gpu.log 2>&1; tail -5 $out/r03_pytest_gpu.log
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r03_bench_baseline.json 2> $out/r03_bench_baseline.err; tail -c 600 $out/r03_bench_baseline.json
    timeout 300 python tools/e2e_demo.py --exp demo_4d_tiny --fast-vae --prune --writers 8 --timeline $out/r03_e2e_tiny_old_timeline.json sampler.plucker_on_device=true data.plucker=cameras > $out/r03_e2e_tiny_old.json 2> $out/r03_e2e_tiny_old.err; cat $out/r03_e2e_tiny_old.json
    timeout 300 python tools/e2e_demo.py --exp demo_4d_tiny --fast-vae --prune --writers 2 --device-results --writer-processes 8 --host-threads 8 --timeline $out/r03_e2e_tiny_new_timeline.json sampler.plucker_on_device=true data.plucker=cameras > $out/r03_e2e_tiny_new.json 2> $out/r03_e2e_tiny_new.err; cat $out/r03_e2e_tiny_new.json
    ;;
  e2e)  # the full demo_4d grid (48 x 150, 344 tasks) end to end with the new host pipeline
    timeout 900 python tools/e2e_demo.py --exp demo_4d --fast-vae --prune --writers 2 --device-results --writer-processes ${1:-12} --host-threads ${2:-8} --depth ${3:-3} --timeline $out/r03_e2e_demo4d_timeline.json sampler.plucker_on_device=true data.plucker=cameras > $out/r03_e2e_demo4d.json 2> $out/r03_e2e_demo4d.err; cat $out/r03_e2e_demo4d.json; tail -3 $out/r03_e2e_demo4d.err
    ;;
  check)  # opcheck / modelcheck prefixes given as arguments, e.g.  check "ff_" "unet_sd21"
    timeout 900 python tests/opcheck.py "${1:-}" > $out/r03_opcheck.log 2>&1; tail -25 $out/r03_opcheck.log
    if [ -n "${2:-}" ]; then timeout 900 python tests/modelcheck.py "$2" > $out/r03_modelcheck.log 2>&1; tail -8 $out/r03_modelcheck.log; fi
    ;;
  ff)  # fused level-0 feed-forward: bit-identity + fp32 parity, per-launch timing, the judged UNet call, a bench step A/B, e2e timers
    timeout 600 python tests/opcheck.py ff_fused > $out/r03_ff_opcheck.log 2>&1; tail -8 $out/r03_ff_opcheck.log
    timeout 300 python tests/opbench.py ff > $out/r03_ff_opbench.log 2>&1; cat $out/r03_ff_opbench.log
    timeout 600 python tests/opcheck.py attn_fp8 > $out/r03_fp8_opcheck.log 2>&1; tail -3 $out/r03_fp8_opcheck.log
    timeout 900 python tests/modelcheck.py unet_sd21_72x40_f16 demo3d > $out/r03_ff_modelcheck.log 2>&1; tail -6 $out/r03_ff_modelcheck.log
    for rep in 1 2; do for f in 1 0; do
      DM4D_FF_FUSED=$f timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-grid-secondary --no-vae > $out/r03_ff_bench_f${f}_$rep.json 2>/dev/null
      python - <<PY
import json; d=json.load(open("$out/r03_ff_bench_f${f}_$rep.json")); print("FF_FUSED=$f rep $rep:", d["value"], "lat/s", d["ms_per_step"], "ms; linear", d["kernel_breakdown_one_step"]["linear"], "ln", d["kernel_breakdown_one_step"]["layernorm"])
PY
    done; done
    timeout 300 python tools/e2e_demo.py --exp demo_4d_tiny --fast-vae --prune --writers 2 --device-results --writer-processes 8 --host-threads ${1:-8} --timeline $out/r03_e2e_tiny_new2_timeline.json sampler.plucker_on_device=true data.plucker=cameras > $out/r03_e2e_tiny_new2.json 2> $out/r03_e2e_tiny_new2.err; cat $out/r03_e2e_tiny_new2.json
    ;;
  ff2)  # fused feed-forward after a schedule change: parity, per-launch timing, bench A/B
    timeout 600 python tests/opcheck.py ff_fused > $out/r03_ff_opcheck.log 2>&1; tail -8 $out/r03_ff_opcheck.log
    timeout 300 python tests/opbench.py ff > $out/r03_ff_opbench.log 2>&1; cat $out/r03_ff_opbench.log
    for rep in 1 2; do for f in 1 0; do
      DM4D_FF_FUSED=$f timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-grid-secondary --no-vae > $out/r03_ff_bench_f${f}_$rep.json 2>/dev/null
      python - <<PY
import json; d=json.load(open("$out/r03_ff_bench_f${f}_$rep.json")); print("FF_FUSED=$f rep $rep:", d["value"], "lat/s", d["ms_per_step"], "ms; linear", d["kernel_breakdown_one_step"]["linear"])
PY
    done; done
    ;;
  e2eprof)  # where the GPU time of an end-to-end run goes: kernel-time table of the whole CLI path (torch's kernels included)
    cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
    rm -rf $out/prof_e2e
    timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_e2e -o stats -- python tools/e2e_demo.py --exp demo_4d_tiny --fast-vae --prune --writers 2 --device-results --writer-processes 8 --host-threads 8 sampler.plucker_on_device=true data.plucker=cameras > $out/r03_e2e_tiny_prof.json 2> $out/r03_e2e_tiny_prof.err
    cat $out/r03_e2e_tiny_prof.json
    DB=$(find $out/prof_e2e -name "stats*results.db" | head -1)
    python tools/profile_summary.py "$DB" "rocprofv3 --kernel-trace --stats -- tools/e2e_demo.py --exp demo_4d_tiny (fast VAE, prune, device results)" > $out/r03_e2e_tiny_kernel_stats.txt
    head -40 $out/r03_e2e_tiny_kernel_stats.txt
    rm -rf $out/prof_e2e
    ;;
  micro)  # GroupNorm loads in flight + 160-wide strip tiles: parity, per-launch A/B (tools/dev/libdm4d_gnbase.so = the tree before), bench A/B, MFMA-busy PMC
    timeout 600 python tests/opcheck.py gn > $out/r03_micro_opcheck_gn.log 2>&1; tail -4 $out/r03_micro_opcheck_gn.log
    timeout 600 python tests/opcheck.py conv > $out/r03_micro_opcheck_conv.log 2>&1; tail -4 $out/r03_micro_opcheck_conv.log
    cp diffuman4d_amd/libdm4d.so tools/dev/libdm4d_new.so
    TAGS="gnbase new gnbase new" bash tools/dev/abn.sh timeout 300 python tests/opbench.py gn > $out/r03_micro_gn_ab.log 2>&1; cat $out/r03_micro_gn_ab.log
    timeout 600 python tools/dev/strip_tune.py --cold > $out/r03_micro_strip_tune_cold.log 2>&1; cat $out/r03_micro_strip_tune_cold.log
    for rep in 1 2; do for v in gnbase new; do
      cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
      timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-grid-secondary --no-vae > $out/r03_micro_bench_${v}_$rep.json 2>/dev/null
      python - <<PY
import json; d=json.load(open("$out/r03_micro_bench_${v}_$rep.json")); k=d["kernel_breakdown_one_step"]; print("$v rep $rep:", d["value"], "lat/s", d["ms_per_step"], "ms; gn", k["groupnorm"]["ms"], "conv", k["conv3x3"]["ms"], "linear", k["linear"]["ms"])
PY
    done; done
    cp tools/dev/libdm4d_new.so diffuman4d_amd/libdm4d.so
    timeout 300 python bench.py --steps 8 --warmup 2 --task-streams 3 --no-cpu-baseline --no-grid-secondary --no-vae > $out/r03_micro_bench_streams3.json 2>/dev/null; python -c "import json; d=json.load(open('$out/r03_micro_bench_streams3.json')); print('streams 3:', d['value'], d['ms_per_step'])"
    bash tools/profile_bench.sh r03m
    rm -rf $out/prof_r03m
    ;;
  verify)  # after the 160-wide tiles were wired: parity first (abort on failure), Linear 160 / 320-wide tiles cold, bench A/B, then the `final` records
    timeout 300 python tests/opcheck.py conv_s > $out/r03_verify_opcheck.log 2>&1; tail -3 $out/r03_verify_opcheck.log
    grep -q "opcheck: \([0-9]*\)/\1 passed" $out/r03_verify_opcheck.log || { echo "PARITY FAILED - stopping"; exit 1; }
    timeout 240 python tools/dev/lin_cold.py --ids=61,62,63,65,68,69 --levels=0,1 > $out/r03_verify_lin_cold.log 2>&1; cat $out/r03_verify_lin_cold.log
    cp diffuman4d_amd/libdm4d.so tools/dev/libdm4d_new.so
    for rep in 1 2; do for v in gnbase new; do
      cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
      timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-bf16 > $out/r03_verify_bench_${v}_$rep.json 2>/dev/null
      python - <<PY
import json
try:
    d=json.load(open("$out/r03_verify_bench_${v}_$rep.json")); k=d["kernel_breakdown_one_step"]; print("$v rep $rep:", d["value"], "lat/s", d["ms_per_step"], "ms; gn", k["groupnorm"]["ms"], "conv", k["conv3x3"]["ms"], "linear", k["linear"]["ms"])
except Exception as e:
    print("$v rep $rep: FAILED", e)
PY
    done; done
    cp tools/dev/libdm4d_new.so diffuman4d_amd/libdm4d.so
    ( time timeout 900 python -m pytest tests -m gpu -q -x ) > $out/r03_pytest_gpu.log 2>&1; tail -6 $out/r03_pytest_gpu.log
    grep -q " passed" $out/r03_pytest_gpu.log && ! grep -q " failed" $out/r03_pytest_gpu.log || { echo "GPU SUITE FAILED - stopping"; exit 1; }
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r03_bench.json 2> $out/r03_bench.err; tail -c 400 $out/r03_bench.json
    bash tools/profile_bench.sh r03
    rm -rf $out/prof_r03
    ;;
  final2)  # last call of round 3: parity of the new Linear tile first (abort on failure), GPU suite, the driver's bench command, profiles, a 3-stream line
    timeout 300 python tests/opcheck.py gemm_n > $out/r03_final2_opcheck.log 2>&1; tail -4 $out/r03_final2_opcheck.log
    grep -q "opcheck: \([0-9]*\)/\1 passed" $out/r03_final2_opcheck.log || { echo "PARITY FAILED - stopping"; exit 1; }
    ( time timeout 900 python -m pytest tests -m gpu -q -x ) > $out/r03_pytest_gpu.log 2>&1; tail -6 $out/r03_pytest_gpu.log
    grep -q " passed" $out/r03_pytest_gpu.log && ! grep -q " failed" $out/r03_pytest_gpu.log || { echo "GPU SUITE FAILED - stopping"; exit 1; }
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r03_bench.json 2> $out/r03_bench.err; tail -c 300 $out/r03_bench.json
    timeout 200 python bench.py --steps 8 --warmup 2 --task-streams 3 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-bf16 > $out/r03_bench_streams3.json 2>/dev/null; python -c "import json; d=json.load(open('$out/r03_bench_streams3.json')); print('streams 3:', d['value'], d['ms_per_step'])"
    timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-bf16 > $out/r03_bench_streams2.json 2>/dev/null; python -c "import json; d=json.load(open('$out/r03_bench_streams2.json')); print('streams 2:', d['value'], d['ms_per_step'], d['kernel_breakdown_one_step'])"
    bash tools/profile_bench.sh r03
    rm -rf $out/prof_r03
    ;;
  ab3)  # bench A/B of three builds inside one call: gnbase = tree at the start of this series, prev = with the 160-wide strip tiles + GroupNorm loads, new = + Linear 128x160 + level-2 rule
    for rep in 1 2; do for v in gnbase prev new; do
      cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so
      timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-bf16 > $out/r03_ab3_${v}_$rep.json 2>/dev/null
      python - <<PY
import json
try:
    d=json.load(open("$out/r03_ab3_${v}_$rep.json")); k=d["kernel_breakdown_one_step"]; print("$v rep $rep:", d["value"], "lat/s", d["ms_per_step"], "ms; gn", k["groupnorm"]["ms"], "conv", k["conv3x3"]["ms"], "linear", k["linear"]["ms"], "attn", k["attention"]["ms"])
except Exception as e:
    print("$v rep $rep: FAILED", e)
PY
    done; done
    cp tools/dev/libdm4d_new.so diffuman4d_amd/libdm4d.so
    ;;
  vaeodd)  # VAE mid-block attention with a padded key axis (latent areas that are not multiples of 32): softmax tail + odd-size VAE parity
    timeout 200 python tests/opcheck.py logits_softmax > $out/r03_vaeodd_opcheck.log 2>&1; tail -6 $out/r03_vaeodd_opcheck.log
    timeout 300 python tests/modelcheck.py vae > $out/r03_vaeodd_modelcheck.log 2>&1; tail -8 $out/r03_vaeodd_modelcheck.log
    ;;
  ffproj)  # attention output projection as a prologue of the fused feed-forward: bit-identity + fp32 parity, per-launch timing, the judged UNet calls, bench A/B
    timeout 200 python tests/opcheck.py ff_ > $out/r03_ffproj_opcheck.log 2>&1; tail -15 $out/r03_ffproj_opcheck.log
    grep -q "opcheck: \([0-9]*\)/\1 passed" $out/r03_ffproj_opcheck.log || { echo "PARITY FAILED - stopping"; exit 1; }
    timeout 200 python tests/opbench.py ffproj > $out/r03_ffproj_opbench.log 2>&1; cat $out/r03_ffproj_opbench.log
    if [ "${1:-}" = "full" ]; then
      timeout 300 python tests/modelcheck.py unet_sd21 demo3d > $out/r03_ffproj_modelcheck.log 2>&1; tail -5 $out/r03_ffproj_modelcheck.log
      grep -q "modelcheck: \([0-9]*\)/\1 passed" $out/r03_ffproj_modelcheck.log || { echo "MODEL PARITY FAILED - stopping"; exit 1; }
      for rep in 1 2; do for f in 1 0; do
        DM4D_FF_PROJ_FUSED=$f timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-grid-secondary --no-vae --no-parity-bf16 > $out/r03_ffproj_bench_f${f}_$rep.json 2>/dev/null
        python - <<PY
import json
try:
    d=json.load(open("$out/r03_ffproj_bench_f${f}_$rep.json")); k=d["kernel_breakdown_one_step"]; print("FF_PROJ_FUSED=$f rep $rep:", d["value"], "lat/s", d["ms_per_step"], "ms; linear", k["linear"])
except Exception as e:
    print("FF_PROJ_FUSED=$f rep $rep: FAILED", e)
PY
      done; done
    fi
    ;;
  final)  # the records that go to profiles/: GPU test suite, the driver's bench command, rocprofv3 stats + PMC, extension lines
    ( time timeout 1800 python -m pytest tests -m gpu -q ) > $out/r03_pytest_gpu.log 2>&1; tail -6 $out/r03_pytest_gpu.log
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r03_bench.json 2> $out/r03_bench.err; tail -c 400 $out/r03_bench.json
    bash tools/profile_bench.sh r03
    # copyBuffer attribution: the same command with 6 timed steps instead of 2 -- launches that do not scale with the step count are init
    cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
    rm -rf $out/prof_r03b
    rocprofv3 --kernel-trace --stats -d $out/prof_r03b -o stats -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --task-streams 1 --no-grid-secondary --no-vae > /dev/null 2> $out/prof_r03b.err
    DB=$(find $out/prof_r03b -name "stats*results.db" | head -1)
    python tools/profile_summary.py "$DB" "same command with --steps 6 --no-vae --no-grid-secondary" | grep -E "total kernel|copyBuffer|fillBuffer|attn_kernel" > $out/r03_copybuffer_steps6.txt
    grep -E "total kernel|copyBuffer|fillBuffer|attn_kernel" $out/r03_kernel_stats.txt > $out/r03_copybuffer_steps2.txt
    cat $out/r03_copybuffer_steps2.txt $out/r03_copybuffer_steps6.txt
    rm -rf $out/prof_r03 $out/prof_r03b
    timeout 600 python bench.py --latent 128x128 --steps 4 --warmup 1 --no-cpu-baseline --no-grid-secondary > $out/r03_bench_latent128.json 2>/dev/null; tail -c 300 $out/r03_bench_latent128.json
    timeout 300 python bench.py --config5 --steps 8 --warmup 2 --no-grid-secondary --no-vae > $out/r03_bench_config5_bf16.json 2>/dev/null; python -c "import json; d=json.load(open('$out/r03_bench_config5_bf16.json')); print('config5 bf16', d['value'], d['ms_per_step'])"
    timeout 300 python bench.py --config5 --attention fp8 --steps 8 --warmup 2 --no-grid-secondary --no-vae > $out/r03_bench_config5_fp8.json 2>/dev/null; python -c "import json; d=json.load(open('$out/r03_bench_config5_fp8.json')); print('config5 fp8', d['value'], d['ms_per_step'])"
    ;;
  *) echo "unknown stage $stage"; exit 2;;
esac
