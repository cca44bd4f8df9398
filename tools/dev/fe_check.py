"""One-process parity run of every GEMM / conv-epilogue case of tests/opcheck.py (results flushed per case)."""
import sys, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2])); sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
import torch, opcheck
torch.manual_seed(0)
out = open("gpurun_out/fe_check.log", "w")
bad = n = 0
for name in opcheck.CASES:
    if name.startswith(("attn", "gn", "ln", "softmax", "layout", "convd")):
        continue
    try:
        err, mx, tol = opcheck.run_case(name)
        ok = err <= tol and math.isfinite(err)
    except Exception as e:  # noqa: BLE001
        ok, err, tol = False, float("nan"), 0.0
        print("ERROR", name, type(e).__name__, str(e)[:200], file=out, flush=True)
    n += 1
    bad += 0 if ok else 1
    print(("PASS" if ok else "FAIL"), name, f"{err:.3e}", f"tol={tol:.1e}", file=out, flush=True)
print(f"fe_check: {n - bad}/{n} passed", file=out, flush=True)
