"""CPU, gloo, world_size 2: layout produced by the frame-shard collectives (K/V all-gather per CFG half in frame
order; latent-row all-gather).  The device-side use (UNet 3-D attention with Lq != Lk) is checked bitwise on the GPU
by tests/opcheck.py::attn_kv_split and tests/modelcheck.py::unet_frame_shard_world1."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffuman4d_amd.host.parallel import FrameShard
        sh = FrameShard()
        cfg, ls, c2 = 2, 3, 4
        full = torch.arange(cfg * world * ls * c2, dtype=torch.float32).view(cfg, world * ls, c2)
        local = full[:, rank * ls:(rank + 1) * ls].contiguous()
        got = sh.gather_kv(local)
        pending = sh.gather_kv_start(local)  # split form used by the UNet (Q projection between start and finish)
        got2 = sh.gather_kv_finish(pending)
        assert torch.equal(got, got2)
        rows = sh.gather_rows(torch.full((2, 5), float(rank)))
        assert sh.local_frames(16) == slice(rank * 8, rank * 8 + 8)
        with pytest.raises(ValueError):
            sh.local_frames(15)
        torch.save({"kv_ok": bool(torch.equal(got, full)), "rows": rows}, f"{outdir}/r{rank}.pt")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_layouts(tmp_path):
    port = 29600 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        blob = torch.load(tmp_path / f"r{r}.pt")
        assert blob["kv_ok"]
        assert blob["rows"].tolist() == [[0.0] * 5] * 2 + [[1.0] * 5] * 2
