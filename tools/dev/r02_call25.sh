#!/bin/bash
# validation after the last source changes (epilogue out_scale skip, fp8 4-wave default, Upsampler guard) + extension lines with the 4-wave fp8 kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c25; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?" >> $O/build.log
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 400 python bench.py --config5 --steps 12 --warmup 3 --no-vae > $O/bench_config5.json 2> $O/bench_config5.err
timeout 400 python bench.py --steps 12 --warmup 3 --no-vae --no-cpu-baseline --attention fp8 > $O/bench_fp8_fast.json 2>> $O/bench_config5.err
tail -2 $O/pytest_gpu.log; tail -1 $O/smoke.log | cut -c1-200; head -c 420 $O/bench.json; echo; head -c 330 $O/bench_config5.json; echo; head -c 260 $O/bench_fp8_fast.json; echo
