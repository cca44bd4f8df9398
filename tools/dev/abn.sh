#!/bin/bash
# A/B/... of several builds: tools/dev/libdm4d_<tag>.so for each tag in $TAGS, running the command given as $@
cp diffuman4d_amd/libdm4d.so /tmp/cur.so
for v in $TAGS; do echo "=== $v"; cp tools/dev/libdm4d_$v.so diffuman4d_amd/libdm4d.so; "$@"; done
cp /tmp/cur.so diffuman4d_amd/libdm4d.so
