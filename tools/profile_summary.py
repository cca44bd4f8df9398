#!/usr/bin/env python
"""Summarise a rocprofv3 (--kernel-trace --stats) rocpd sqlite database into a per-kernel table.
usage: python tools_profile_summary.py <results.db> <title> > profiles/<name>.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# {sys.argv[2]}")
print(f"# total kernel time {tot/1e6:.2f} ms")
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for n, c, s, a, mn, mx in rows:
    n2 = re.sub(r"\(anonymous namespace\)::", "", n)[:72]
    print(f"{n2:72s} {c:6d} {s/1e6:10.3f} {a/1e3:10.1f} {mn/1e3:9.1f} {mx/1e3:9.1f} {100*s/tot:6.2f}")
