"""precision "fp16" holds the checkpoint's values in fp16: a value outside fp16's range is refused at load (host/unet.py::_Weights)."""
import pytest
import torch

from diffuman4d_amd.host.unet import _Weights


def test_fp16_precision_refuses_a_weight_outside_the_fp16_range():
    ok = _Weights._to_f16(torch.tensor([1.0, -3.0e4, 2.0 ** -20], dtype=torch.bfloat16))
    assert ok.dtype == torch.float16 and bool(torch.isfinite(ok).all())
    assert float(ok[2]) == 2.0 ** -20  # a bf16 value below 2^-14 that fp16's subnormals still hold exactly
    with pytest.raises(ValueError, match="outside the fp16"):
        _Weights._to_f16(torch.tensor([1.0, 1.0e5], dtype=torch.bfloat16))
