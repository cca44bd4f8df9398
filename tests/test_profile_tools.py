"""The profile post-processing tools (tools/profile_summary.py, pmc_summary.py, mfma_busy_summary.py, dev/isa_diff.py) on synthetic
rocprofv3 databases / listings: the numbers quoted in DESIGN.md come through these scripts, so their arithmetic is pinned here.  CPU only."""
import json
import sqlite3
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(*args):
    r = subprocess.run([sys.executable, *map(str, args)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    return r.stdout


def _db(path, kernels, counters=()):
    db = sqlite3.connect(path)
    db.execute("create table kernels (name text, start integer, end integer)")
    db.executemany("insert into kernels values (?, ?, ?)", kernels)
    db.execute("create table counters_collection (kernel_name text, counter_name text, dispatch_id integer, value real)")
    db.executemany("insert into counters_collection values (?, ?, ?, ?)", counters)
    db.commit()
    db.close()


def test_kernel_time_table(tmp_path):
    _db(tmp_path / "stats_results.db", [("void (anonymous namespace)::attn_kernel<true, 8>(AttnParams)", 0, 400_000),
                                         ("void (anonymous namespace)::attn_kernel<true, 8>(AttnParams)", 500_000, 1_100_000),
                                         ("gn_apply_kernel(GNParams)", 0, 250_000)])
    out = _run("tools/profile_summary.py", tmp_path / "stats_results.db", "title").splitlines()
    assert out[0] == "# title" and out[1] == "# total kernel time 1.25 ms"
    row = out[3].split()
    assert out[3].startswith("void attn_kernel<true, 8>(AttnParams)")  # the anonymous-namespace prefix is dropped
    assert row[-6:] == ["2", "1.000", "500.0", "400.0", "600.0", "80.00"]  # calls, total ms, avg / min / max us, percent


def test_hbm_traffic_summary(tmp_path):
    d = tmp_path / "prof"
    d.mkdir()
    for c, vals in (("FETCH_SIZE", (100.0, 300.0)), ("WRITE_SIZE", (50.0, 70.0))):
        rows = []
        for disp, v in enumerate(vals):  # a counter arrives as several rows per dispatch (one per XCD): they add up
            rows += [("attn_kernel<true, 8>", c, disp, v / 2), ("attn_kernel<true, 8>", c, disp, v / 2)]
        rows.append(("other_kernel", c, 9, 1e9))
        _db(d / f"pmc_{c}_results.db", [], rows)
    out = json.loads(_run("tools/pmc_summary.py", d, "attn_kernel", "FETCH_SIZE", "WRITE_SIZE"))
    assert out["FETCH_SIZE"] == {"launches": 2, "avg_kb": 200.0} and out["WRITE_SIZE"] == {"launches": 2, "avg_kb": 60.0}
    # what bench.py makes of it: HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (the gfx950 FETCH_SIZE correction)
    assert int((2 * out["FETCH_SIZE"]["avg_kb"] + out["WRITE_SIZE"]["avg_kb"]) * 1024) == 471040


def test_mfma_busy_summary(tmp_path):
    # two launches: 32 busy cycles per MFMA and SIMD summed over 1024 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs
    cyc = (200_000, 400_000)  # shader cycles each launch took
    rows, kern = [], []
    t = 0
    for disp, c in enumerate(cyc):
        rows.append(("attn_kernel<true, 8>", "SQ_VALU_MFMA_BUSY_CYCLES", disp, 0.5 * c * 1024))  # pipes busy half the time
        rows += [("attn_kernel<true, 8>", "GRBM_GUI_ACTIVE", disp, float(c))] * 8
        kern.append(("attn_kernel<true, 8>", t, t + c // 2))  # 2 cycles per ns = 2.0 GHz
        t += c
    _db(tmp_path / "pmc_MFMA_results.db", kern, rows)
    out = json.loads(_run("tools/mfma_busy_summary.py", tmp_path / "pmc_MFMA_results.db", "attn_kernel"))
    assert out["launches"] == 2 and out["mfma_busy"] == 0.5 and out["clock_ghz"] == 2.0
    assert abs(out["frac_of_nominal_peak"] - 0.5 * 2.0 / 2.4) < 1e-4


def test_isa_diff(tmp_path):
    a = tmp_path / "a.s"
    b = tmp_path / "b.s"
    a.write_text("_Z1kv: ; @k\n\ts_mov_b32 s0, 1 ; comment\n.LBB0_1:\n\tv_add_u32 v0, v0, v1\n\ts_endpgm\n"
                 "_Z1gv:\n\ts_nop 0\n\ts_endpgm\n")
    b.write_text("_Z1kv: ; @k\n\ts_mov_b32   s0, 1\n.LBB7_1:\n\tv_add_u32 v0, v0, v1\n\ts_endpgm\n"
                 "_Z1gv:\n\ts_nop 1\n\ts_endpgm\n_Z1hv:\n\ts_nop 0\n\ts_endpgm\n")
    out = _run("tools/dev/isa_diff.py", a, b).split()
    assert out == ["DIFF", "_Z1gv", "SAME", "_Z1kv", "NEW", "_Z1hv"]  # by name; labels, comments and spacing do not count
