"""Oracle for the phase decomposition of Upsample2D.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference op is ``Upsample2D.forward`` of the third-party dependency diffusers==0.33.1 (``/root/reference/requirements.txt:5``;
not vendored), instantiated at ``/root/reference/src/diffusers/models/unets/unet_multiview_blocks.py:620`` (up blocks) and in
the VAE decoder: ``F.interpolate(x, scale_factor=2.0, mode="nearest")`` followed by a 3x3 convolution with padding 1.  ``phase_weights`` / ``conv_up2x`` restate it as four 2x2 convolutions of the low-resolution input -- the form
``dm4d_conv_up2x_nhwc_bf16`` runs -- so that the algebra can be checked on the CPU against the reference form
(tests/test_up2x_algebra.py) independently of the HIP kernel.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# taps of the 3-tap axis that read low-resolution offset d (0: the earlier pixel, 1: the later) of output phase p
TAPS = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}


def phase_weights(w: torch.Tensor) -> torch.Tensor:
    """w [Cout, Cin, 3, 3] -> [2, 2, Cout, Cin, 2, 2] indexed [py, px, :, :, dy, dx]: sums of the taps sharing a pixel."""
    out = w.new_zeros((2, 2) + tuple(w.shape[:2]) + (2, 2))
    for py in (0, 1):
        for px in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    out[py, px, :, :, dy, dx] = sum(w[:, :, ky, kx] for ky in TAPS[py][dy] for kx in TAPS[px][dx])
    return out


def conv_up2x(x: torch.Tensor, w: torch.Tensor, bias=None) -> torch.Tensor:
    """x [B, Cin, H, W] -> [B, Cout, 2H, 2W] through the four phase convolutions: phase (py, px) reads low-resolution
    rows y - 1 + py, y + py and columns x - 1 + px, x + px (zero outside the image)."""
    B, _, H, W = x.shape
    wp = phase_weights(w)
    y = x.new_zeros((B, w.shape[0], 2 * H, 2 * W))
    for py in (0, 1):
        for px in (0, 1):
            xp = F.pad(x, (1 - px, px, 1 - py, py))  # (left, right, top, bottom)
            y[:, :, py::2, px::2] = F.conv2d(xp, wp[py, px])
    if bias is not None:
        y = y + bias[None, :, None, None]
    return y


def reference_form(x: torch.Tensor, w: torch.Tensor, bias=None) -> torch.Tensor:
    return F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, bias, padding=1)
