import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device; run them on the GPU box")
    from diffuman4d_amd.host import lib
    lib.load()  # fail loudly if libdm4d.so is missing: there is no fallback path
    return torch.device("cuda:0")
