#!/usr/bin/env python
"""Per-launch average of rocprofv3 --pmc counters for one kernel.
usage: pmc_summary.py <rocprof output dir> <kernel name substring> COUNTER [COUNTER ...]
Reads the rocpd sqlite databases written by `rocprofv3 --kernel-trace --pmc C -o pmc_C` (one pass per counter).
FETCH_SIZE / WRITE_SIZE are reported in KiB as rocprofv3 emits them; bench.py applies the gfx950 correction
(HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md, HBM section)."""
import glob
import json
import sqlite3
import sys

root, kname, counters = sys.argv[1], sys.argv[2], sys.argv[3:]
out = {}
for c in counters:
    dbs = glob.glob(f"{root}/**/pmc_{c}*results.db", recursive=True)
    if not dbs:
        out[c] = None
        continue
    db = sqlite3.connect(dbs[0])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tables else None
    if view is None:
        out[c] = {"error": "no counters_collection view", "tables": tables[:20]}
        continue
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = list(db.execute(f"select dispatch_id, sum(value) from {view} where {kcol} like ? and counter_name = ? "
                           f"group by dispatch_id", (f"%{kname}%", c)))
    vals = [r[1] for r in rows]
    out[c] = {"launches": len(vals), "avg_kb": sum(vals) / max(1, len(vals))}
print(json.dumps(out, indent=1))
