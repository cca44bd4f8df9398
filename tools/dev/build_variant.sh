#!/bin/bash
# build tools/dev/libdm4d_<tag>.so = the current objects (diffuman4d_amd/build/*.o) with ONE source recompiled with extra flags
# usage: tools/dev/build_variant.sh <tag> <source stem: attention|gemm|norm|...> [-DFLAG=..]...
set -e
tag=$1; stem=$2; shift 2
python -m diffuman4d_amd.build > /dev/null
mkdir -p /tmp/vb
extra=""; [ $stem = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"  # build.EXTRA_FLAGS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -Iinclude -Idiffuman4d_amd/csrc -c diffuman4d_amd/csrc/$stem.hip -o /tmp/vb/${stem}_$tag.o
objs=""
for f in $(python -c "from diffuman4d_amd import build; print(' '.join(s[:-4] for s in build.SOURCES))"); do  # every object host/lib.py resolves symbols from
  if [ $f = $stem ]; then objs="$objs /tmp/vb/${stem}_$tag.o"; else objs="$objs diffuman4d_amd/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o tools/dev/libdm4d_$tag.so $objs
echo built tools/dev/libdm4d_$tag.so
