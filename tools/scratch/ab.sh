#!/bin/bash
# A/B of two builds of libdm4d.so on the attention micro-benchmark: tools/scratch/libdm4d_base.so vs the current build
cp diffuman4d_amd/libdm4d.so /tmp/new.so
echo "=== base"; cp tools/scratch/libdm4d_base.so diffuman4d_amd/libdm4d.so; python tests/opbench.py attn 2>&1 | grep "^attn"
echo "=== new";  cp /tmp/new.so diffuman4d_amd/libdm4d.so; python tests/opbench.py attn 2>&1 | grep "^attn"
