#!/usr/bin/env python
"""GPU: how far one UNet call of the fast / fp16 precision is from the fp32 result AS A FUNCTION OF ITS INPUT -- the bench's task latents
after n units of the (random-weight) sweep.  Reference = the same call in the parity precision (1.3e-5 of the fp32 CPU oracle, bench
`parity.modes.parity`), so no CPU forward is needed.  Prints one row per n; bench.py compares on the state after PARITY_STATE_UNITS."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    from diffuman4d_amd.host import ops
    from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
    from diffuman4d_amd.host.scheduler import DDIMScheduler
    from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel as HU
    from diffuman4d_amd.host.weights import random_state_dict, unet_param_shapes
    dev = torch.device("cuda", 0)
    cfg = UNetConfig()
    sd = random_state_dict(unet_param_shapes(cfg), 0, dev)
    unets = {p: HU(cfg, sd, dev, p) for p in ("fast", "fp16", "parity")}
    pipe = Diffuman4DPipeline(None, unets["fast"], DDIMScheduler(), dev)
    one = bench.build_tasks(pipe, dev)
    task, frames, done = one["spatial"], 16, 0
    tb = task["tables"]
    print("units  lat_rms   fast_rel_l2  fp16_rel_l2   (vs the parity precision, first spatial window, F = 16, CFG batch 32)")
    with torch.no_grad():
        for n in (0, 1, 2, 4, 8, 16, 24, 37, 48, 64):
            while done < n:
                bench.run_unit(pipe, one, done)
                done += 1
            widx, cond = tb["win"][0][:frames], tb["cond"][0][:frames]
            x = ops.pack_model_input(task["lat"].clone(), task["pv"], task["pl"], task["sk"], task["cm"], cond.contiguous(),
                                     pipe.unet.IN_PAD, True, frame_idx=widx.contiguous())
            t_in = torch.cat([tb["t"][0][:frames]] * 2)
            xb = x.view(2 * frames, bench.LAT_H, bench.LAT_W, pipe.unet.IN_PAD)
            kw = dict(domains=["spatial"] * 2, num_frames=frames)
            out = {"fast": unets["fast"](xb, t_in, **kw).float()}
            for p in ("fp16", "parity"):
                out[p] = unets[p](ops.split(xb.float(), h16=p == "fp16"), t_in, **kw).float()
            ref = out["parity"]
            rel = {p: float((out[p] - ref).norm() / ref.norm()) for p in ("fast", "fp16")}
            print(f"{n:5d}  {float(task['lat'].float().pow(2).mean().sqrt()):8.3f}  {rel['fast']:.4e}   {rel['fp16']:.4e}", flush=True)


if __name__ == "__main__":
    main()
