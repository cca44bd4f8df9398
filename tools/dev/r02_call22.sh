#!/bin/bash
# full validation of the tree: GPU tests, smoke, judged bench line, op bench, rocprof stats + PMC traffic, extension lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c22; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?" >> $O/build.log
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 python tests/opbench.py > $O/opbench.log 2>&1
bash tools/profile_bench.sh r02b > $O/profile.log 2>&1
cp gpurun_out/r02b_kernel_stats.txt gpurun_out/r02b_attn_traffic_pmc.json gpurun_out/r02b_bench_under_rocprof.json $O/ 2>/dev/null
rm -rf gpurun_out/prof_r02b
timeout 400 python bench.py --config5 --steps 12 --warmup 3 --no-vae > $O/bench_config5.json 2> $O/bench_config5.err
timeout 400 python bench.py --steps 12 --warmup 3 --no-vae --no-cpu-baseline --attention fp8 > $O/bench_fp8_fast.json 2>> $O/bench_config5.err
timeout 400 python bench.py --latent 128x128 --steps 4 --warmup 2 --no-cpu-baseline --no-vae > $O/bench_128.json 2> $O/bench_128.err
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; head -c 500 $O/bench.json; echo; head -14 $O/r02b_kernel_stats.txt; cat $O/r02b_attn_traffic_pmc.json; head -c 400 $O/bench_config5.json; echo; head -c 300 $O/bench_fp8_fast.json; echo; head -c 300 $O/bench_128.json; echo; du -sh gpurun_out
