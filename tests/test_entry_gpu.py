"""GPU: the driver's entry points in ONE fresh process, build() before smoke() -- the order in which the HIP runtime
copies get loaded matters (libdm4d.so must bind to PyTorch's libamdhip64, see host/lib.py::load)."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_build_then_smoke_in_one_process():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke(); print('ENTRY-OK')"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ENTRY-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
