"""Instruction mix between consecutive s_barrier instructions of one kernel in a hipcc -S listing.
usage: python tools/dev/asm_segments.py file.s <mangled-name-substring>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M)
i = m.start(); j = s.index('.Lfunc_end', i)
lines = s[i:j].splitlines()
bars = [k for k, l in enumerate(lines) if re.match(r'\s+s_barrier', l)]
print(m.group(1), len(lines), 'lines; barriers at', bars)
for a, b in zip([0] + bars, bars + [len(lines)]):
    c = Counter(re.findall(r'^\s+([a-z_0-9]+)', '\n'.join(lines[a:b]), re.M))
    g = lambda *ks: sum(v for k, v in c.items() if any(k.startswith(x) for x in ks))
    print(f"{a:5d}-{b:5d} mfma {g('v_mfma'):3d} exp {g('v_exp'):3d} add {g('v_add_f32') + 2 * g('v_pk_add_f32'):3d} (pk {g('v_pk_add_f32'):2d}) cvt {g('v_cvt_pk'):3d} "
          f"ds_read {g('ds_read'):3d} accmov {g('v_accvgpr'):3d} v_mov {g('v_mov'):3d} nop {g('s_nop'):3d} waitcnt {g('s_waitcnt'):3d} total {b - a}")
