#!/bin/bash
# 128x128-latent bench line, task-stream sweep on the judged config, end-to-end CLI demo (host vs device Pluecker maps)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c10; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?" >> $O/build.log
for s in 1 2 3 4; do
  timeout 300 python bench.py --steps 24 --warmup 4 --task-streams $s --no-cpu-baseline --no-vae > $O/bench_streams$s.json 2> $O/bench_streams$s.err
done
timeout 400 python bench.py --latent 128x128 --steps 4 --warmup 2 --no-cpu-baseline --no-vae > $O/bench_128.json 2> $O/bench_128.err
timeout 600 python tools/e2e_demo.py --exp demo_4d_tiny --fast-vae --prune > $O/e2e_host_plucker.json 2> $O/e2e_host.err
timeout 600 python tools/e2e_demo.py --exp demo_4d_tiny --fast-vae --prune sampler.plucker_on_device=true data.plucker=cameras > $O/e2e_dev_plucker.json 2> $O/e2e_dev.err
timeout 600 python tools/e2e_demo.py --exp demo_4d_tiny --fast-vae --prune --gpu-streams 3 sampler.plucker_on_device=true data.plucker=cameras > $O/e2e_dev_plucker_s3.json 2> $O/e2e_dev3.err
for f in $O/bench_streams*.json $O/bench_128.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("task_streams"))
except Exception as e: print(sys.argv[1], "ERR", e)
P
done
tail -n1 $O/e2e_*.json; tail -n3 $O/*.err | tail -40
