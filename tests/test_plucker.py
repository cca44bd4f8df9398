"""CPU: oracle/plucker.py against the reference's own ray_utils (build container only -- the module needs nothing but
torch, so it is imported from /root/reference directly) and the host-side camera constants handed to the HIP kernel."""
import importlib.util
import math
from pathlib import Path

import pytest
import torch

from oracle import plucker as op

REF = Path("/root/reference/src/data/utils/ray_utils.py")


def cameras(n=5, H=64, W=40, seed=0):
    """Cameras on a ring looking at the origin (DNA-Rendering-like), camera-to-world poses, pinhole K."""
    g = torch.Generator().manual_seed(seed)
    Ks, poses = [], []
    for k in range(n):
        a = 2 * math.pi * k / n + float(torch.rand((), generator=g)) * 0.1
        o = torch.tensor([2.5 * math.cos(a), 0.3 * float(torch.randn((), generator=g)), 2.5 * math.sin(a)])
        fwd = -o / o.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]))
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        pose = torch.eye(4)
        pose[:3, :3] = torch.stack([right, down, fwd], dim=1)
        pose[:3, 3] = o
        poses.append(pose)
        f = 1.2 * W * (1 + 0.05 * k)
        Ks.append(torch.tensor([[f, 0.0, W / 2 + k], [0.0, f, H / 2 - k], [0.0, 0.0, 1.0]]))
    return torch.stack(Ks), torch.stack(poses)


@pytest.mark.skipif(not REF.exists(), reason="the reference checkout exists only in the build container")
def test_oracle_plucker_is_pinned_to_the_reference():
    spec = importlib.util.spec_from_file_location("ref_ray_utils", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    Ks, poses = cameras()
    assert torch.equal(ref.calc_relative_poses(poses), op.calc_relative_poses(poses))
    rel = op.calc_relative_poses(poses)
    a, b = ref.calc_plucker_embeds(64, 40, Ks, rel), op.calc_plucker_embeds(64, 40, Ks, rel)
    assert a.shape == b.shape == (5, 6, 64, 40)
    assert float((a - b).abs().max()) <= 1e-6
    assert float(a[:, :3].norm(dim=1).sub(1).abs().max()) < 1e-5  # unit directions


def test_camera_rows_layout():
    from diffuman4d_amd.host.ops import camera_rows
    Ks, poses = cameras(3)
    rows = camera_rows(Ks, poses)
    assert rows.shape == (3, 24) and rows.dtype == torch.float32
    ext = torch.inverse(poses)
    assert torch.allclose(rows[:, :9].view(3, 3, 3) @ Ks, torch.eye(3).expand(3, 3, 3), atol=1e-5)
    assert torch.equal(rows[:, 9:18].view(3, 3, 3), ext[:, :3, :3]) and torch.equal(rows[:, 18:21], ext[:, :3, 3])
    assert torch.allclose(rows[:, 21:24], poses[:, :3, 3], atol=1e-5)  # -R^T T is the camera centre
