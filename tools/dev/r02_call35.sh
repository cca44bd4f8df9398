#!/bin/bash
# attention forms 1 / 2 / 3: bit-identity + timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c35; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tools/dev/attn_form_ab.py > $O/attn_forms.log 2>&1; grep -v amdgpu.ids $O/attn_forms.log
