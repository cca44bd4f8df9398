#!/usr/bin/env python
"""MFMA-busy fraction of one kernel from a rocprofv3 PMC pass (`--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`).
usage: mfma_busy_summary.py <results.db> <kernel name substring>

SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per v_mfma_f32_32x32x16_bf16 and SIMD (MI355X_MICROARCH.md, cycle constants), summed over
the chip's 1 024 SIMDs; GRBM_GUI_ACTIVE counts busy shader-clock cycles per XCD, summed over the 8 XCDs.  So, per launch,
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)       (fraction of the MFMA issue slots AT THE CLOCK THE CHIP RAN)
    clock_ghz = GRBM_GUI_ACTIVE / 8 / duration                                 (the chip clocks to its power budget under MFMA load)
and mfma_busy * clock_ghz / 2.4 is the fraction of the nominal 2.5 PFLOP/s peak -- the number `bench.py` derives from FLOPs and time.
Launch-weighted sums over all matching dispatches (the bench's 48 launches per step have very different sizes)."""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
kname = sys.argv[2]
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
assert "counters_collection" in tables, tables[:20]
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else "name"
tot = {}
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    rows = list(db.execute(f"select dispatch_id, sum(value) from counters_collection where {kcol} like ? and counter_name = ? "
                           f"group by dispatch_id", (f"%{kname}%", c)))
    tot[c] = (len(rows), sum(r[1] for r in rows))
dur = list(db.execute("select count(*), sum(end - start) from kernels where name like ?", (f"%{kname}%",)))[0]
n, busy = tot["SQ_VALU_MFMA_BUSY_CYCLES"]
_, gui = tot["GRBM_GUI_ACTIVE"]
out = {"kernel": kname, "launches": n, "SQ_VALU_MFMA_BUSY_CYCLES_sum": busy, "GRBM_GUI_ACTIVE_sum": gui,
       "kernel_time_ms_sum": dur[1] / 1e6 if dur[1] else None}
if gui:
    out["mfma_busy"] = round(busy / (gui / 8.0 * 1024.0), 4)
    if dur[1]:
        out["clock_ghz"] = round(gui / 8.0 / dur[1], 3)
        out["frac_of_nominal_peak"] = round(out["mfma_busy"] * out["clock_ghz"] / 2.4, 4)
print(json.dumps(out, indent=1))
