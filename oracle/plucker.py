"""Oracle Pluecker maps: restatement of the reference's camera-ray conditioning.  TEST INFRASTRUCTURE ONLY.

Follows ``/root/reference/src/data/utils/ray_utils.py``
  * ``:6-8``    ``normalize``: x / (|x| + 1e-8)
  * ``:11-33``  ``get_rays``: pixel grid i (rows) / j (columns) of the H x W image
  * ``:36-98``  ``get_rays_from_ij``: o = -R^T T; pixel centres (+0.5); d = normalize(R^T (K^-1 [j, i, 1]^T - T) - o)
  * ``:101-112`` ``calc_plucker_embeds``: [R | T] = inverse(pose)[:3]; embed = [d | o x d] as [B, 6, H, W]
  * ``:115-119`` ``calc_relative_poses``: poses relative to the sample's first camera (spatem_dataset.py:169-173)
and the consumer ``pipeline_diffuman4d.py:90-100`` (``F.interpolate(..., mode="bilinear")`` to the latent size, cast).
Pinned by tests/test_plucker.py against the reference's own ray_utils (importable: it needs only torch).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def calc_relative_poses(poses: torch.Tensor) -> torch.Tensor:
    return torch.matmul(torch.inverse(poses[0]), poses)


def calc_plucker_embeds(h: int, w: int, K: torch.Tensor, pose: torch.Tensor) -> torch.Tensor:
    ext = torch.inverse(pose)
    R, T = ext[:, :3, :3], ext[:, :3, 3:]
    i, j = torch.meshgrid(torch.arange(h, dtype=R.dtype), torch.arange(w, dtype=R.dtype), indexing="ij")
    xy1 = torch.stack([j + 0.5, i + 0.5, torch.ones_like(i)], dim=-1)[None, ..., None]  # [1, h, w, 3, 1]
    inv_k = torch.inverse(K.float()).type(K.dtype)[:, None, None]
    r_t = R.mT[:, None, None]
    ray_o = (-R.mT @ T)[:, None, None]
    pixel_world = r_t @ (inv_k @ xy1 - T[:, None, None])
    ray_d = (pixel_world - ray_o)[..., 0]
    ray_d = ray_d / (torch.norm(ray_d, dim=-1, keepdim=True) + 1e-8)
    ray_o = ray_o[..., 0].expand_as(ray_d)
    return torch.cat([ray_d, torch.cross(ray_o, ray_d, dim=-1)], dim=-1).permute(0, 3, 1, 2)


def plucker_latents(h_img: int, w_img: int, K: torch.Tensor, pose: torch.Tensor, latent_size, dtype=torch.bfloat16) -> torch.Tensor:
    """What the UNet is fed: the full-resolution map resized on the host in fp32, then cast (pipeline_diffuman4d.py:90-100)."""
    return F.interpolate(calc_plucker_embeds(h_img, w_img, K, pose), size=tuple(latent_size), mode="bilinear").to(dtype)
