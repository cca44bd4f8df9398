#!/bin/bash
# Tile choice under in-model cache conditions (cold weights / outputs, producer-warm activations): Linear ids and strip-conv tiles
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c31; mkdir -p $O
export TMPDIR=/tmp
timeout 280 python tools/dev/lin_cold.py > $O/lin_cold.log 2>&1; grep -v amdgpu.ids $O/lin_cold.log | cut -c 1-260
timeout 200 python tools/dev/strip_tune.py --cold > $O/strip_cold.log 2>&1; grep -v amdgpu.ids $O/strip_cold.log | cut -c 1-200
