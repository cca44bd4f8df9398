"""Probe for the next round: can one window call (pack -> UNet -> CFG + DDIM, ~330 launches from libdm4d.so through ctypes)
be captured into a HIP graph via torch.cuda.graph and replayed bitwise-identically, and what does a replay cost against
the eager launch sequence?  (A window call has static shapes; only the small index / timestep / coefficient rows change
from call to call, and those can live in static device buffers.)"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench
from diffuman4d_amd.host.pipeline import Diffuman4DPipeline
from diffuman4d_amd.host.scheduler import DDIMScheduler
from diffuman4d_amd.host.unet import UNetConfig, UNetMultiviewConditionModel
from diffuman4d_amd.host.weights import random_state_dict, unet_param_shapes

dev = torch.device("cuda", 0)
cfg = UNetConfig()
unet = UNetMultiviewConditionModel(cfg, random_state_dict(unet_param_shapes(cfg), 0, dev), dev)
pipe = Diffuman4DPipeline(None, unet, DDIMScheduler(), dev)
task = bench.build_tasks(pipe, dev)["spatial"]
lat0 = task["lat"].clone()
K = 6
try:
    with torch.no_grad():
        bench.run_call(pipe, task, 0)
        ref = task["lat"].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            bench.run_call(pipe, task, 0)
        torch.cuda.synchronize()
        t_eager = (time.perf_counter() - t0) / K
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            task["lat"].copy_(lat0)
            bench.run_call(pipe, task, 0)  # warm-up on the capture stream
            task["lat"].copy_(lat0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            bench.run_call(pipe, task, 0)
        torch.cuda.synchronize()
        task["lat"].copy_(lat0)
        g.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(task["lat"], ref))
        t0 = time.perf_counter()
        for _ in range(K):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / K
    print(f"graph capture ok; replay bitwise equal to eager: {same}; eager {t_eager*1e3:.2f} ms/call, graph replay {t_graph*1e3:.2f} ms/call", flush=True)
except BaseException as e:  # noqa: BLE001
    import traceback
    traceback.print_exc()
    print("graph capture FAILED:", type(e).__name__, str(e)[:400], flush=True)
