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
SOURCES = ["api.hip", "gemm.hip", "gemm_h16.hip", "ff_fused.hip", "conv_direct.hip", "attention.hip", "norm.hip", "elementwise.hip", "parity.hip"]
# attention.hip: the 4-wave x 64-row kernel form needs more than 256 registers per lane; without this flag hipcc puts every MFMA
# result of such a kernel into AGPRs and copies the score accumulators to VGPRs and back on every step (0.67x, measured);
# kernels that fit in 256 registers are unaffected
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
# sources that #include another .hip source (gemm_h16.hip = the PAR = 2 instantiations of gemm.hip's kernels)
EXTRA_DEPS = {"gemm_h16.hip": ["gemm.hip"]}


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
    for f in sorted([CSRC / s for s in SOURCES] + list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h"))):  # EXTRA_DEPS are SOURCES too
        h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def needs_build() -> bool:
    """True unless libdm4d.so exists AND was built from exactly the sources in the tree: the content hash recorded at build time
    must match (file times do not survive a copy to another machine, and a stale library must never be blessed)."""
    if not LIB.exists() or not STAMP.exists():
        return True
    return STAMP.read_text().strip() != source_hash()


def _object_key(src: Path, deps, headers, cmd_flags) -> str:
    """What an object file was made from: its source, the sources it includes, every header, and the full flag list (so that a
    tuning build's -D objects are never reused by a plain build, and the other way round).  File times are not consulted."""
    import hashlib
    h = hashlib.sha256()
    for f in [src] + sorted(deps) + sorted(headers):
        h.update(f.name.encode() + b"\0" + f.read_bytes() + b"\0")
    h.update(repr(list(cmd_flags)).encode())
    return h.hexdigest()


def _stale(obj: Path, key: str) -> bool:
    stamp = obj.with_suffix(".o.key")
    return not obj.exists() or not stamp.exists() or stamp.read_text().strip() != key


def build(force: bool = False, verbose: bool = True, defines=()) -> Path:
    """One object per .hip source (compiled in parallel; an object is reused only if the content hash recorded beside it -- source,
    included sources, headers, flags -- matches), then one link.  `defines`: extra -D flags (tuning builds)."""
    if not force and not defines and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objdir = ROOT / "build"
    objdir.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list(INCLUDE.glob("*.h"))  # attn64_asm.inc: tools/attn64/gen.py --write
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}"] + [f"-D{d}" for d in defines]
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src, obj = CSRC / s, objdir / (Path(s).stem + ".o")
        cmd_flags = [*flags, *EXTRA_FLAGS.get(s, [])]
        key = _object_key(src, [CSRC / d for d in EXTRA_DEPS.get(s, [])], headers, cmd_flags)
        if force or _stale(obj, key):
            jobs.append(([hipcc, *cmd_flags, "-c", str(src), "-o", str(obj)], obj, key))

    def run(cmd):
        if verbose:
            print("[dm4d build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    def compile_one(job):
        cmd, obj, key = job
        stamp = obj.with_suffix(".o.key")
        if stamp.exists():
            stamp.unlink()  # a failed or interrupted compile leaves no key behind
        run(cmd)
        stamp.write_text(key + "\n")
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as ex:
        list(ex.map(compile_one, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB)] + [str(objdir / (Path(s).stem + ".o")) for s in SOURCES])
    if not defines:
        STAMP.write_text(source_hash() + "\n")
    elif STAMP.exists():
        STAMP.unlink()  # a tuning build: never mistaken for the shipped library
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
