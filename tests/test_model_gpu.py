"""-m gpu: HIP UNet / VAE / full sliding_iterative_denoise vs the CPU oracle (same weights, same noise)."""
import math

import pytest

import modelcheck


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(modelcheck.CASES))
def test_model(name, hip_device):
    err, yard, tol = modelcheck.run_case(name)
    assert math.isfinite(err) and err <= tol, f"{name}: rel_l2={err:.3e} (tol {tol:.1e}); oracle-bf16 yardstick {yard:.3e}"
