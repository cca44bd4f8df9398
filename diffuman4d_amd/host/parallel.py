"""In-window frame sharding over RCCL (BASELINE.json config 4).

Rank r of P holds F/P consecutive frames of every window call.  Everything in the UNet is per-frame
(convs, GroupNorm statistics are per sample, LayerNorm, feed-forward, 2-D attention) except the 3-D
attention layers, whose K/V must cover all F frames: one all-gather of the local [cfg, F/P*HW, 2C] K|V
block per 3-D layer (11 per UNet call).  Heads (5/10/20) do not divide 8, so head-parallel all-to-all
is not an option; the all-gather moves each rank's 1/P share over its own xGMI link.  After the DDIM
update the F/P updated latent rows are all-gathered so every rank's copy of the task tensor stays whole.

Query rows see their keys in the same global order and tile boundaries as on one GPU, so results are
bitwise identical to the unsharded path (tests/opcheck.py::attn_kv_split).
Works with backend "nccl" (= RCCL) and "gloo" (CPU tests of the layout).
"""
from __future__ import annotations

from typing import Optional

import torch


class FrameShard:
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def local_frames(self, num_frames: int) -> slice:
        if num_frames % self.world != 0:
            raise ValueError(f"window of {num_frames} frames cannot be sharded over {self.world} ranks")
        fl = num_frames // self.world
        return slice(self.rank * fl, (self.rank + 1) * fl)

    def gather_rows(self, x_local: torch.Tensor) -> torch.Tensor:
        """[n, ...] on every rank -> [world*n, ...] in rank order."""
        x_local = x_local.contiguous()
        out = torch.empty((self.world * x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        # one contiguous output buffer: RCCL writes every rank's block in place (the list form of all_gather goes through a
        # flatten + per-rank copy-out)
        self.dist.all_gather_into_tensor(out, x_local, group=self.group)
        return out

    def gather_kv_start(self, kv_local: torch.Tensor):
        """Start the all-gather of kv_local [cfg, Ls, 2C] (this rank's frames) and return a handle for gather_kv_finish.
        With RCCL the collective runs on the communicator's own stream, so whatever is launched on the compute
        stream between start and finish (the Q projection of the same layer) overlaps with the xGMI transfer."""
        cfg, ls, c2 = kv_local.shape
        out = torch.empty((cfg, self.world, ls, c2), dtype=kv_local.dtype, device=kv_local.device)
        works = []
        for b in range(cfg):  # one collective per CFG half: each output block is contiguous and in frame order
            works.append(self.dist.all_gather_into_tensor(out[b].view(self.world * ls, c2), kv_local[b].contiguous(),
                                                          group=self.group, async_op=True))
        return out, works

    def gather_kv_finish(self, handle) -> torch.Tensor:
        """-> [cfg, world*Ls, 2C] with ranks (= frames) in order; the compute stream waits for the collectives."""
        out, works = handle
        for w in works:
            w.wait()
        cfg, world, ls, c2 = out.shape
        return out.view(cfg, world * ls, c2)

    def gather_kv(self, kv_local: torch.Tensor) -> torch.Tensor:
        """kv_local [cfg, Ls, 2C] (this rank's frames) -> [cfg, world*Ls, 2C] with ranks (= frames) in order."""
        return self.gather_kv_finish(self.gather_kv_start(kv_local))
