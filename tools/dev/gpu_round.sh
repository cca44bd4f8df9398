( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu2.log 2>&1; tail -4 gpurun_out/pytest_gpu2.log
python bench.py > gpurun_out/bench_v8.json 2> gpurun_out/bench_v8.err; python bench.py --task-streams 1 --no-cpu-baseline > gpurun_out/bench_v8_s1.json 2>/dev/null; python bench.py --task-streams 3 --steps 6 --no-cpu-baseline > gpurun_out/bench_v8_s3.json 2>/dev/null
