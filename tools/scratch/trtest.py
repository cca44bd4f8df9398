import ctypes, torch, numpy as np
hip = ctypes.CDLL("libamdhip64.so")
lib = ctypes.CDLL("tools/scratch/libtr.so")
# launch via hipModule API is clumsy; use hipLaunchKernel through a tiny C shim? simpler: use torch + hipModuleLoad
import subprocess
