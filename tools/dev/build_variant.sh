#!/bin/bash
# build tools/dev/libdm4d_<tag>.so = the current objects (diffuman4d_amd/build/*.o) with ONE source recompiled with extra flags
# usage: tools/dev/build_variant.sh <tag> <source stem: attention|gemm|norm|...> [-DFLAG=..]...
set -e
tag=$1; stem=$2; shift 2
python -m diffuman4d_amd.build > /dev/null
mkdir -p /tmp/vb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -Iinclude -Idiffuman4d_amd/csrc -c diffuman4d_amd/csrc/$stem.hip -o /tmp/vb/${stem}_$tag.o
objs=""
for f in api gemm ff_fused conv_direct attention norm elementwise; do
  if [ $f = $stem ]; then objs="$objs /tmp/vb/${stem}_$tag.o"; else objs="$objs diffuman4d_amd/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/dev/libdm4d_$tag.so $objs
echo built tools/dev/libdm4d_$tag.so
