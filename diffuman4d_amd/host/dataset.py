"""Synthetic stand-in for ``SpaTemDataset`` with the same ``get_item`` output contract.

The real dataset (``/root/reference/src/data/spatem_dataset.py``: PIL decode, mask-bbox crop, bicubic
resize, camera parsing) is host-side file I/O and out of scope for this round.  This
class produces tensors with the documented shapes and value ranges (``spatem_dataset.py:191-228``:
pixel / skeleton / Pluecker in [-1, 1], masks in {0, 1}) so the sampler, pipeline and CLI can be run
and benchmarked without data: cameras on a ring looking at the origin, Pluecker maps [d, o x d].
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch


def relative_poses(poses: torch.Tensor) -> torch.Tensor:
    """Camera-to-world poses expressed in the frame of the sample's first camera (spatem_dataset.py:169-173)."""
    return torch.inverse(poses[0]) @ poses


def plucker_maps(height: int, width: int, Ks: torch.Tensor, poses: torch.Tensor) -> torch.Tensor:
    """[N, 6, H, W] fp32 maps with the reference's convention (ray_utils.py:101-112): unit ray through every pixel centre in
    the frame the poses are given in, and its moment about the origin, [d | o x d]."""
    ext = torch.inverse(poses)
    R, T = ext[:, :3, :3], ext[:, :3, 3]
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32) + 0.5, torch.arange(width, dtype=torch.float32) + 0.5,
                            indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], dim=0).reshape(3, -1)            # [3, HW]
    cam = torch.inverse(Ks) @ pix                                                      # [N, 3, HW]
    origin = -(R.mT @ T[:, :, None])                                                   # [N, 3, 1]
    d = R.mT @ (cam - T[:, :, None]) - origin
    d = d / (d.norm(dim=1, keepdim=True) + 1e-8)
    m = torch.linalg.cross(origin.expand_as(d), d, dim=1)
    return torch.cat([d, m], dim=1).reshape(-1, 6, height, width)


class SyntheticSpaTemDataset:
    """plucker="host" (default, what SpaTemDataset does): ``get_item`` returns the full-resolution fp32 Pluecker maps.
    plucker="cameras": it returns ``plucker_embeds=None`` and the sampler hands the cameras (``Ks``, ``poses``, present in
    both modes as in the reference's sample dict, spatem_dataset.py:178-189) to the pipeline, which evaluates the rays at
    latent resolution on the device (``SlidingIterativeSampler(plucker_on_device=True)``)."""

    def __init__(self, scene_label: str = "synthetic", height: int = 576, width: int = 320, num_cameras: int = 48,
                 data_dir: str = "", seed: int = 1234, plucker: str = "host", **_ignored):
        if plucker not in ("host", "cameras"):
            raise ValueError("plucker must be 'host' or 'cameras'")
        self.scene_label, self.height, self.width = scene_label, height, width
        self.num_cameras, self.data_dir, self.seed, self.plucker = num_cameras, data_dir, seed, plucker

    # -- geometry ---------------------------------------------------------------------------------
    def _pose(self, cam: int) -> torch.Tensor:
        """Camera-to-world 4x4 of camera `cam` on a ring of radius 0.35 looking at the origin (x right, y down, z forward)."""
        a = 2 * math.pi * cam / self.num_cameras
        o = torch.tensor([0.35 * math.cos(a), 0.0, 0.35 * math.sin(a)])
        fwd = -o / o.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]))
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        pose = torch.eye(4)
        pose[:3, :3] = torch.stack([right, down, fwd], dim=1)
        pose[:3, 3] = o
        return pose

    def _intrinsics(self) -> torch.Tensor:
        f = 1.2 * self.width
        return torch.tensor([[f, 0.0, self.width / 2], [0.0, f, self.height / 2], [0.0, 0.0, 1.0]])

    def nearest_input_camera(self, cam: int, input_cams: Sequence[int]) -> int:
        o = self._pose(cam)[:3, 3]
        return min(input_cams, key=lambda c: float((self._pose(c)[:3, 3] - o).norm()))

    # -- the get_item contract (spatem_dataset.py:77-229) -------------------------------------------
    def get_item(self, scene_label: str, spa_labels: List[str], tem_labels: List[str], input_spa_labels: List[str]) -> Dict:
        H, W = self.height, self.width
        if len(tem_labels) == 1:  # spatial sample: all cameras of one frame
            labels = [(i, s, tem_labels[0]) for i, s in enumerate(spa_labels)]
        else:  # temporal sample: T frames of the nearest input camera, then T frames of the target camera
            tgt = spa_labels[0]
            near = f"{self.nearest_input_camera(int(tgt), [int(c) for c in input_spa_labels]):02d}"
            labels = [(i, near, t) for i, t in enumerate(tem_labels)]
            labels += [(len(tem_labels) + i, tgt, t) for i, t in enumerate(tem_labels)]
        n = len(labels)
        pix = torch.empty(n, 3, H, W)
        skel = -torch.ones(n, 3, H, W)
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
        inside = ((xs / 0.55) ** 2 + (ys / 0.85) ** 2) < 1.0
        for k, (_, s, t) in enumerate(labels):
            g = torch.Generator().manual_seed(self.seed + 1000 * int(s) + int(t))
            img = torch.rand(3, H, W, generator=g) * 2 - 1
            pix[k] = torch.where(inside, img, torch.ones_like(img))  # white background (:166)
            r0 = int(H * 0.2) + (int(t) % 7)
            skel[k, :, r0:r0 + 4, W // 4: 3 * W // 4] = torch.rand(3, 1, 1, generator=g) * 2 - 1
        cams = sorted({int(s) for _, s, _ in labels}, key=[int(s) for _, s, _ in labels].index)
        pose_of = {c: self._pose(c) for c in cams}
        poses = relative_poses(torch.stack([pose_of[int(s)] for _, s, _ in labels]))  # relative to the sample's first camera
        Ks = self._intrinsics()[None].repeat(n, 1, 1)
        pl = None
        if self.plucker == "host":
            per_cam = {c: plucker_maps(H, W, Ks[:1], poses[[i for i, (_, s, _) in enumerate(labels) if int(s) == c][:1]])[0]
                       for c in cams}  # a camera's map is the same for every frame of the sample
            pl = torch.stack([per_cam[int(s)] for _, s, _ in labels]).clamp(-1, 1)
        return {"pixel_values": pix, "skeletons": skel, "plucker_embeds": pl, "cond_masks": torch.ones(n, 1, H, W),
                "labels": labels, "crops": [None] * n, "Ks": Ks, "poses": poses, "hws": [(H, W)] * n}
