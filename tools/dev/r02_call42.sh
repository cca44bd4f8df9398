#!/bin/bash
# HIP pipeline with the DPM-Solver++ rows against the fixture made by the REFERENCE's pipeline with one stateful scheduler per latent
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c42; mkdir -p $O
export TMPDIR=/tmp
( timeout 120 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "golden_dpm or golden_spatial" ) > $O/pytest.log 2>&1; grep -E "golden |passed|failed|Error" $O/pytest.log | tail -6
