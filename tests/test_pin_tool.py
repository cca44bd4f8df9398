"""tools/pin_with_diffusers.py must skip cleanly where diffusers is absent (the build container and the GPU box) and must never
report a mismatch where it is present.  CPU only."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_pin_tool_skips_or_pins():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "pin_with_diffusers.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode in (0, 77), r.stdout[-2000:] + r.stderr[-2000:]
    if r.returncode == 77:
        assert "SKIP" in r.stdout
    else:
        assert "all blocks pinned" in r.stdout


def test_pin_tool_self_test_walks_the_multistep_grid():
    """The UniPC / DEIS comparison loop of the tool, run oracle-against-oracle: its configurations are all accepted by oracle/multistep.py
    and the walk itself (timestep tables, 12 steps, corrector on / off) executes -- so the check is not dead code where diffusers is absent."""
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "pin_with_diffusers.py"), "--self-test"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SELF-TEST" in r.stdout and "deviation 0.0e+00" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
