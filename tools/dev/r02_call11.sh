#!/bin/bash
# single-launch GroupNorm: parity cases, op bench A/B vs the two-launch pair, model-level checks, bench line, e2e with more writers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c11; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?" >> $O/build.log
( timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gn_ or ln_" ) > $O/pytest_gn.log 2>&1
timeout 300 python tests/opbench.py gn > $O/opbench_gn.log 2>&1
( timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -x ) > $O/pytest_model.log 2>&1
timeout 600 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-vae > $O/bench.json 2> $O/bench.err
timeout 600 python tools/e2e_demo.py --exp demo_4d_tiny --fast-vae --prune --writers 8 sampler.plucker_on_device=true data.plucker=cameras > $O/e2e_w8.json 2> $O/e2e_w8.err
tail -5 $O/pytest_gn.log; cat $O/opbench_gn.log; tail -5 $O/pytest_model.log; cut -c1-400 $O/bench.json; echo; grep -o '"phases".*' $O/bench.json | cut -c1-600; cat $O/e2e_w8.json
