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


class SyntheticSpaTemDataset:
    def __init__(self, scene_label: str = "synthetic", height: int = 576, width: int = 320, num_cameras: int = 48,
                 data_dir: str = "", seed: int = 1234, **_ignored):
        self.scene_label, self.height, self.width = scene_label, height, width
        self.num_cameras, self.data_dir, self.seed = num_cameras, data_dir, seed
        self._plucker_cache: Dict[int, torch.Tensor] = {}

    # -- geometry ---------------------------------------------------------------------------------
    def _camera(self, cam: int):
        a = 2 * math.pi * cam / self.num_cameras
        o = torch.tensor([0.35 * math.cos(a), 0.0, 0.35 * math.sin(a)])
        fwd = -o / o.norm()
        up = torch.tensor([0.0, 1.0, 0.0])
        right = torch.linalg.cross(fwd, up)
        right = right / right.norm()
        up = torch.linalg.cross(right, fwd)
        return o, torch.stack([right, up, fwd], dim=1)  # columns: camera axes in world coords

    def _plucker(self, cam: int) -> torch.Tensor:
        """[6, H, W]: unit ray direction d and moment o x d (ray_utils.py:101-112 convention)."""
        if cam in self._plucker_cache:  # a camera's map is the same for every frame
            return self._plucker_cache[cam]
        H, W = self.height, self.width
        o, R = self._camera(cam)
        f = 1.2 * W
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32) + 0.5, torch.arange(W, dtype=torch.float32) + 0.5,
                                indexing="ij")
        d_cam = torch.stack([(xs - W / 2) / f, -(ys - H / 2) / f, torch.ones_like(xs)], dim=0)
        d = torch.einsum("ij,jhw->ihw", R, d_cam)
        d = d / torch.sqrt((d * d).sum(0, keepdim=True))
        m = torch.linalg.cross(o[:, None, None].expand_as(d), d, dim=0)
        out = self._plucker_cache[cam] = torch.cat([d, m], dim=0)
        return out

    def nearest_input_camera(self, cam: int, input_cams: Sequence[int]) -> int:
        o = self._camera(cam)[0]
        return min(input_cams, key=lambda c: float((self._camera(c)[0] - o).norm()))

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
        pl = torch.empty(n, 6, H, W)
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
        for k, (_, s, t) in enumerate(labels):
            g = torch.Generator().manual_seed(self.seed + 1000 * int(s) + int(t))
            img = torch.rand(3, H, W, generator=g) * 2 - 1
            inside = ((xs / 0.55) ** 2 + (ys / 0.85) ** 2) < 1.0
            pix[k] = torch.where(inside, img, torch.ones_like(img))  # white background (:166)
            r0 = int(H * 0.2) + (int(t) % 7)
            skel[k, :, r0:r0 + 4, W // 4: 3 * W // 4] = torch.rand(3, 1, 1, generator=g) * 2 - 1
            pl[k] = self._plucker(int(s))
        return {"pixel_values": pix, "skeletons": skel, "plucker_embeds": pl.clamp(-1, 1),
                "cond_masks": torch.ones(n, 1, H, W), "labels": labels, "crops": [None] * n}
