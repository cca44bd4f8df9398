"""How long does the CPU-oracle sample of bench.py take for a few (threads, frames) settings on this box?"""
import os, sys, time, torch
sys.path.insert(0, '.')
from oracle.unet import UNetConfig, UNetMultiviewConditionModel
cfg = UNetConfig()
with torch.no_grad():
    m = UNetMultiviewConditionModel(cfg).eval()
    for threads, frames in ((32, 2), (64, 2), (128, 2), (64, 4), (128, 4)):
        torch.set_num_threads(threads)
        x = torch.randn(2 * frames, cfg.in_channels, 72, 40); t = torch.randint(0, 1000, (2 * frames,))
        t0 = time.time(); m(x, t, domains=["spatial"] * 2, num_frames=frames); dt = time.time() - t0
        print(f"threads {threads:4d} frames {frames}: {dt:6.1f} s", flush=True)
print("cpu_count", os.cpu_count())
