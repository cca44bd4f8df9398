"""Build libdm4d.so (HIP, gfx950) in-tree.  `python -m diffuman4d_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
INCLUDE = ROOT.parent / "include"
LIB = ROOT / "libdm4d.so"
SOURCES = ["api.hip", "gemm.hip", "conv_direct.hip", "attention.hip", "norm.hip", "elementwise.hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{INCLUDE}", f"-I{CSRC}",
           "-o", str(LIB)] + [str(CSRC / s) for s in SOURCES]
    if verbose:
        print("[dm4d build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
