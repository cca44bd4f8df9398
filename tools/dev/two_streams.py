"""Experiment: N independent task streams on one GPU (tasks of a round do not interact): units/s with 1, 2, 3 worker
threads, each on its own HIP stream, sharing one set of weights."""
import sys, threading, time
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
NMAX, K = 3, 4
task_sets = [bench.build_tasks(pipe, dev) for _ in range(NMAX)]

def worker(i, k, stream):
    torch.cuda.set_device(0)
    with torch.no_grad(), torch.cuda.stream(stream):
        for u in range(k):
            bench.run_unit(pipe, task_sets[i], u)

with torch.no_grad():
    bench.run_unit(pipe, task_sets[0], 0)
torch.cuda.synchronize()
for n in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i, K, streams[i])) for i in range(n)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} stream(s): {n*K} units in {dt*1e3:.1f} ms -> {dt*1e3/(n*K):.2f} ms/unit, {2*n*K/dt:.2f} latents/s", flush=True)
