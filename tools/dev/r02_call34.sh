#!/bin/bash
# attention, second form (4 waves x 64 query rows, in-place score registers, block-level software pipeline): bit-identity + timing,
# default build (AGPR-form MFMAs with copies) and a build of attention.hip with -mllvm -amdgpu-mfma-vgpr-form=1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c34; mkdir -p $O
export TMPDIR=/tmp
cp diffuman4d_amd/libdm4d.so /tmp/new.so
timeout 200 python tools/dev/attn_form_ab.py > $O/attn_form_default.log 2>&1; grep -v amdgpu.ids $O/attn_form_default.log
cp tools/dev/libdm4d_vgprform.so diffuman4d_amd/libdm4d.so
timeout 200 python tools/dev/attn_form_ab.py > $O/attn_form_vgprform.log 2>&1; grep -v amdgpu.ids $O/attn_form_vgprform.log
cp /tmp/new.so diffuman4d_amd/libdm4d.so
