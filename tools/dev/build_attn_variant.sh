#!/bin/bash
# build tools/dev/libdm4d_<tag>.so = current objects + attention.hip compiled with extra flags
# usage: tools/dev/build_attn_variant.sh <tag> [-DFLAG=..]...
set -e
tag=$1; shift
mkdir -p /tmp/vb
for f in api gemm conv_direct norm elementwise; do
  if [ ! -f /tmp/vb/$f.o ] || [ diffuman4d_amd/csrc/$f.hip -nt /tmp/vb/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Idiffuman4d_amd/csrc -c diffuman4d_amd/csrc/$f.hip -o /tmp/vb/$f.o &
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -Iinclude -Idiffuman4d_amd/csrc -c diffuman4d_amd/csrc/attention.hip -o /tmp/vb/attn_$tag.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/dev/libdm4d_$tag.so /tmp/vb/api.o /tmp/vb/gemm.o /tmp/vb/conv_direct.o /tmp/vb/norm.o /tmp/vb/elementwise.o /tmp/vb/attn_$tag.o
echo built tools/dev/libdm4d_$tag.so
