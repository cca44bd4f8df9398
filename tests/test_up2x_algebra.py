"""CPU: the four-phase form of Upsample2D (oracle/up2x.py, what dm4d_conv_up2x_nhwc_bf16 computes) equals nearest-x2 + 3x3
convolution, including image borders, odd sizes and 1-pixel-wide images."""
import pytest
import torch

from oracle import up2x


@pytest.mark.parametrize("shape", [(2, 5, 7, 6, 4), (1, 3, 1, 9, 2), (3, 4, 8, 1, 5), (1, 2, 1, 1, 3), (2, 8, 9, 5, 8)])
def test_phase_form_equals_reference_form(shape):
    B, Cin, H, W, Cout = shape
    g = torch.Generator().manual_seed(B * 100 + H * 10 + W)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    ref = up2x.reference_form(x, w, b)
    out = up2x.conv_up2x(x, w, b)
    assert out.shape == ref.shape == (B, Cout, 2 * H, 2 * W)
    assert torch.allclose(out, ref, rtol=0, atol=1e-12), float((out - ref).abs().max())


def test_phase_weights_cover_every_tap_once():
    w = torch.arange(2 * 3 * 9, dtype=torch.float64).reshape(2, 3, 3, 3)
    wp = up2x.phase_weights(w)
    # every phase's four kernels add up to the sum of all nine taps
    total = w.sum(dim=(2, 3))
    for py in (0, 1):
        for px in (0, 1):
            assert torch.equal(wp[py, px].sum(dim=(2, 3)), total)
