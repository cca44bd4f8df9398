#!/usr/bin/env python
"""Per-launch averages of every counter in the rocprofv3 --pmc passes written by tools/dev/pmc.sh.
usage: pmc_report.py <dir> <tag> [kernel-name substring ...]   (reads <dir>/**/<tag>_[a-e]*results.db)"""
import glob
import sqlite3
import sys

root, tag, subs = sys.argv[1], sys.argv[2], sys.argv[3:]
rows = {}
for dbf in sorted(glob.glob(f"{root}/**/{tag}_[a-e]*results.db", recursive=True)):
    db = sqlite3.connect(dbf)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    q = f"select {kcol}, counter_name, dispatch_id, sum(value) from counters_collection group by {kcol}, counter_name, dispatch_id"
    for k, c, d, v in db.execute(q):
        if subs and not any(s in k for s in subs):
            continue
        if "at::native" in k or "elementwise" in k.lower() and "dm4d" not in k:
            continue
        rows.setdefault((k.split("(")[0][:70], c), []).append(v)
names = sorted({k for k, _ in rows})
for k in names:
    print(k)
    for (kk, c), v in sorted(rows.items()):
        if kk == k:
            print(f"    {c:34s} launches={len(v):3d} avg={sum(v) / len(v):16.1f}")
