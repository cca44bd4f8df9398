"""CPU: the C-ABI shared library loads (no GPU needed) and exports every symbol include/dm4d.h declares;
the ctypes prototype table in diffuman4d_amd/host/lib.py covers exactly that set with matching arity."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "dm4d.h").read_text()


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\s*\*)\s+(dm4d_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_header_declares_the_expected_surface():
    fns = header_functions()
    for name in ("dm4d_gemm_bf16", "dm4d_conv3x3_nhwc_bf16", "dm4d_attention_bf16", "dm4d_groupnorm_nhwc_bf16",
                 "dm4d_layernorm_bf16", "dm4d_pack_model_input_bf16", "dm4d_cfg_ddim_step_bf16"):
        assert name in fns


def test_library_builds_loads_and_exports_every_symbol():
    from diffuman4d_amd import build
    lib_path = build.build(force=False, verbose=False)
    lib = ctypes.CDLL(str(lib_path))
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in dm4d.h but not exported by {lib_path.name}"
    lib.dm4d_version.restype = ctypes.c_int
    assert lib.dm4d_version() >= 100


def test_ctypes_table_matches_header():
    from diffuman4d_amd.host import lib as L
    fns = header_functions()
    assert set(L.SIGNATURES) == set(fns)
    for name, (_, argtypes) in L.SIGNATURES.items():
        assert len(argtypes) == fns[name], f"{name}: header has {fns[name]} args, ctypes table {len(argtypes)}"
    L.load()


def test_ops_refuse_cpu_tensors():
    """There is no CPU fallback: a host tensor is an error, never a silent slow path."""
    import torch
    from diffuman4d_amd.host import lib as L, ops
    a = torch.zeros(32, 32, dtype=torch.bfloat16)
    with pytest.raises(L.Dm4dError, match="HIP device"):
        ops.gemm(a, a)
    with pytest.raises(L.Dm4dError, match="HIP device"):
        ops.layernorm(a, a[0], a[0])


def test_conv_split_decision_depends_on_the_image_not_on_the_batch():
    """dm4d_conv3x3_ws_bytes (host-only query): the split over the kernel rows applies to small images with a deep K
    (the 9x5 level) and to stride-1 'same' convolutions only, and whether it applies never depends on B -- a
    frame-sharded rank convolves fewer images per launch and must sum in the same order as the unsharded run."""
    from diffuman4d_amd.host import lib as L
    lib = L.load()

    def ws(B, H, W, Cin, Cout, stride=1, pad=1, up=0, Ho=None, Wo=None):
        Ho, Wo = (H if Ho is None else Ho), (W if Wo is None else Wo)
        return lib.dm4d_conv3x3_ws_bytes(B, H, W, Cin, Ho, Wo, Cout, stride, pad, up)

    for B in (1, 2, 3, 32, 48):
        assert ws(B, 9, 5, 1280, 1280) == 3 * B * 45 * 1280 * 4      # three fp32 partial planes
        assert ws(B, 9, 5, 2560, 1280) == 3 * B * 45 * 1280 * 4
        assert ws(B, 8, 8, 512, 128) == 3 * B * 64 * 128 * 4
        assert ws(B, 18, 10, 1280, 1280) == 0                        # 180 pixels: enough rows, never split
        assert ws(B, 9, 5, 256, 1280) == 0                           # shallow K
        assert ws(B, 9, 5, 1280, 1280, stride=2, Ho=5, Wo=3) == 0    # stride 2 is not a strip convolution
        assert ws(B, 9, 5, 1280, 1280, up=1, Ho=18, Wo=10) == 0      # neither is the fused up-sampling
        assert ws(B, 9, 5, 1280, 1284) == 0                          # N not a multiple of 8: no vector epilogue


def test_a_stale_library_is_not_blessed(monkeypatch, tmp_path):
    """`needs_build()` compares the content hash recorded when libdm4d.so was linked with the hash of the sources in the tree (file times do
    not survive the copy to a GPU box): one changed byte in a kernel source, a header or a compile flag makes the library stale, and
    `__graft_entry__.build()` asserts on exactly this predicate after building."""
    from diffuman4d_amd import build
    build.build(force=False, verbose=False)
    assert not build.needs_build()
    h0 = build.source_hash()
    # a different recorded hash (= the library was linked from other sources)
    stamp = tmp_path / "stamp"
    stamp.write_text("0" * 64)
    monkeypatch.setattr(build, "STAMP", stamp)
    assert build.needs_build()
    stamp.write_text(h0)
    assert not build.needs_build()
    # one more byte in a source: the tree's hash moves, the recorded one does not
    fake = tmp_path / "csrc"
    fake.mkdir()
    for f in list(build.CSRC.glob("*.hip")) + list(build.CSRC.glob("*.h")):
        (fake / f.name).write_bytes(f.read_bytes())
    (fake / build.SOURCES[0]).write_bytes((fake / build.SOURCES[0]).read_bytes() + b"\n")
    monkeypatch.setattr(build, "CSRC", fake)
    assert build.source_hash() != h0 and build.needs_build()
    # a missing stamp is stale too
    monkeypatch.setattr(build, "CSRC", build.ROOT / "csrc")
    monkeypatch.setattr(build, "STAMP", tmp_path / "absent")
    assert build.needs_build()
