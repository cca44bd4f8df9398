#!/bin/bash
# full validation of the round-2 tree + the judged bench line + rocprof stats / PMC traffic
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c9; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?" >> $O/build.log
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 python tests/opbench.py > $O/opbench.log 2>&1
bash tools/profile_bench.sh r02 > $O/profile.log 2>&1
cp gpurun_out/r02_kernel_stats.txt gpurun_out/r02_attn_traffic_pmc.json gpurun_out/r02_bench_under_rocprof.json $O/ 2>/dev/null
rm -rf gpurun_out/prof_r02
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; head -c 600 $O/bench.json; echo; head -14 $O/r02_kernel_stats.txt; cat $O/r02_attn_traffic_pmc.json; du -sh gpurun_out
