set -e
cp diffuman4d_amd/libdm4d.so /tmp/orig.so
cp tools/scratch/libdm4d_abl.so diffuman4d_amd/libdm4d.so
for d in 0 0x40000000 0x20000000; do echo "== DM4D_DBG=$d (0x4..=no DMA, 0x2..=no MFMA)"; DM4D_DBG=$d python tests/opbench.py 2>&1 | grep -E "conv L1 |conv L2 |conv L0 |ff2 L1|qkv L2|out L0"; done
cp /tmp/orig.so diffuman4d_amd/libdm4d.so
