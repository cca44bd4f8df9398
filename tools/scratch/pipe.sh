for v in 1 2 3 4 5 6; do echo "== PIPE=$v"; DM4D_GEMM_PIPE=$v python tests/opcheck.py 2>&1 | grep -E "opcheck:|FAIL|ERROR"; DM4D_GEMM_PIPE=$v python tests/opbench.py 2>&1 | grep -E "gemm|conv" | cut -c1-34,50-80 > gpurun_out/pipe$v.log; done
python tests/opbench.py 2>&1 | grep -E "gemm|conv" | cut -c1-34,50-80 > gpurun_out/pipe0.log
paste -d"|" gpurun_out/pipe0.log <(cut -c35-70 gpurun_out/pipe1.log) <(cut -c35-70 gpurun_out/pipe2.log) <(cut -c35-70 gpurun_out/pipe3.log) <(cut -c35-70 gpurun_out/pipe4.log) <(cut -c35-70 gpurun_out/pipe5.log) <(cut -c35-70 gpurun_out/pipe6.log)
