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
SOURCES = ["api.hip", "gemm.hip", "ff_fused.hip", "conv_direct.hip", "attention.hip", "norm.hip", "elementwise.hip", "parity.hip"]
# attention.hip: the 4-wave x 64-row kernel form needs more than 256 registers per lane; without this flag hipcc puts every MFMA
# result of such a kernel into AGPRs and copies the score accumulators to VGPRs and back on every step (0.67x, measured);
# kernels that fit in 256 registers are unaffected
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


STAMP = ROOT / "libdm4d.so.srchash"  # sha256 of the sources the shipped libdm4d.so was built from (travels with it; git-ignored)


def source_hash() -> str:
    """sha256 over every source, header and compile flag that goes into libdm4d.so."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted([CSRC / s for s in SOURCES] + list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))):
        h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def needs_build() -> bool:
    """True unless libdm4d.so exists AND was built from exactly the sources in the tree: the content hash recorded at build time
    must match (file times do not survive a copy to another machine, and a stale library must never be blessed)."""
    if not LIB.exists() or not STAMP.exists():
        return True
    return STAMP.read_text().strip() != source_hash()


def _stale(obj: Path, src: Path, headers) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    return src.stat().st_mtime > t or any(h.stat().st_mtime > t for h in headers)


def build(force: bool = False, verbose: bool = True, defines=()) -> Path:
    """One object per .hip source (compiled in parallel, rebuilt only when the source or a header changed), then one
    link.  `defines`: extra -D flags (tuning builds); they force a full rebuild."""
    if not force and not defines and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objdir = ROOT / "build"
    objdir.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}"] + [f"-D{d}" for d in defines]
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src, obj = CSRC / s, objdir / (Path(s).stem + ".o")
        if force or defines or _stale(obj, src, headers):
            jobs.append([hipcc, *flags, *EXTRA_FLAGS.get(s, []), "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print("[dm4d build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(objdir / (Path(s).stem + ".o")) for s in SOURCES])
    if not defines:
        STAMP.write_text(source_hash() + "\n")
    elif STAMP.exists():
        STAMP.unlink()  # a tuning build: never mistaken for the shipped library
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
