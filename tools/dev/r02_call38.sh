#!/bin/bash
# multistep scheduler (DPM-Solver++): kernel parity + full tiny-pipeline parity against the oracle with one stateful scheduler per latent
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c38; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "linear_step or pack_ddim" ) > $O/pytest_ops.log 2>&1; tail -2 $O/pytest_ops.log
( timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "pipeline_dpm or pipeline_spatial or pipeline_temporal_v" ) > $O/pytest_model.log 2>&1; grep -E "pipeline |passed|failed|Error" $O/pytest_model.log | tail -12
