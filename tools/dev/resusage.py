"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr of a compile) per kernel: registers, scratch, occupancy.
usage: python tools/dev/resusage.py remarks.txt"""
import re, sys
t = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', t)[1:]
def g(b, k):
    m = re.search(k + r': (\d+)', b)
    return int(m.group(1)) if m else -1
for b in blocks:
    name = b.split('\n')[0].strip()
    v, a, s, o, l = (g(b, 'VGPRs'), g(b, 'AGPRs'), g(b, r'ScratchSize \[bytes/lane\]'), g(b, r'Occupancy \[waves/SIMD\]'),
                     g(b, r'LDS Size \[bytes/block\]'))
    print(f"{name[:100]:100s} vgpr {v:4d} agpr {a:4d} scratch {s:5d} occ {o} lds {l}")
