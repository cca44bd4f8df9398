#!/bin/bash
# A/B of two builds of libdm4d.so (tools/dev/libdm4d_base.so vs the current build) with the command given as $@
cp diffuman4d_amd/libdm4d.so /tmp/new.so
for round in 1 2; do
echo "=== base (round $round)"; cp tools/dev/libdm4d_base.so diffuman4d_amd/libdm4d.so; "$@"
echo "=== new (round $round)";  cp /tmp/new.so diffuman4d_amd/libdm4d.so; "$@"
done
