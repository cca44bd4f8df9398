#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c20; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/dev/strip_tune.py > $O/strip_tune.log 2>&1
grep -v amdgpu $O/strip_tune.log
