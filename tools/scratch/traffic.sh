cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcb -o fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>gpurun_out/pmcb_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcb -o write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>gpurun_out/pmcb_write.err
ls -la gpurun_out/pmcb
