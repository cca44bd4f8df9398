#!/bin/bash
# after the attention kernel: parity of the attention cases and of the judged model cases, then the bench line
export TMPDIR=/tmp
{
echo "=== opcheck attn*"; timeout 900 python tests/opcheck.py attn h16_attn 2>&1 | grep -v "^PASS" | tail -15
echo "=== modelcheck judged cases"; timeout 1200 python tests/modelcheck.py unet_sd21_72x40 fp16_unet_sd21_72x40 fp16_demo3d unet_frame_shard fp16_unet_frame_shard task_stack 2>&1 | grep -E "PASS|FAIL|ERROR|modelcheck:" | tail -30
echo "=== bench"; timeout 900 python bench.py > gpurun_out/r06_bench_attn64.json 2> gpurun_out/r06_bench_attn64.err; tail -c 3000 gpurun_out/r06_bench_attn64.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_attn64.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"])
kb = d.get("secondary", {}).get("kernel_breakdown_one_step") or d.get("kernel_breakdown_one_step")
print({k: v for k, v in (kb or {}).items() if "." not in k})
tm = d.get("secondary", {}).get("tolerance_mode", {})
print("tolerance_mode", tm.get("ms_per_step"), tm.get("latents_per_s"), {k: v for k, v in (tm.get("kernel_breakdown_one_step") or {}).items() if "." not in k})
PY
} > gpurun_out/r06_stage1.log 2>&1
tail -60 gpurun_out/r06_stage1.log
