#!/bin/bash
# full validation of the tree (round 2, second session): GPU tests, smoke, judged bench line, op bench, rocprof stats + PMC traffic
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c39; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --task-streams 3 --no-cpu-baseline --no-vae > $O/bench_s3.json 2>> $O/bench.err
timeout 300 python tests/opbench.py > $O/opbench.log 2>&1
bash tools/profile_bench.sh r02d > $O/profile.log 2>&1
cp gpurun_out/r02d_kernel_stats.txt gpurun_out/r02d_attn_traffic_pmc.json gpurun_out/r02d_bench_under_rocprof.json $O/ 2>/dev/null
rm -rf gpurun_out/prof_r02d
timeout 300 python bench.py --latent 128x128 --steps 4 --warmup 2 --no-cpu-baseline --no-vae > $O/bench_128.json 2> $O/bench_128.err
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; head -c 600 $O/bench.json; echo; python -c "
import json
for f in ('bench_s3','bench_128'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'])"
head -16 $O/r02d_kernel_stats.txt; cat $O/r02d_attn_traffic_pmc.json; du -sh gpurun_out
