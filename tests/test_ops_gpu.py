"""-m gpu: every HIP kernel, called through the C ABI, against a plain fp32 PyTorch reference."""
import math

import pytest

import opcheck


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(opcheck.CASES))
def test_op(name, hip_device):
    err, mx, tol = opcheck.run_case(name)
    assert math.isfinite(err) and err <= tol, f"{name}: rel_l2={err:.3e} (tol {tol:.1e}), max_abs={mx:.3e}"
