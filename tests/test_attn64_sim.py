"""The hand-placed attention stream (tools/attn64/gen.py -> diffuman4d_amd/csrc/attn64_asm.inc) on the numpy model of a workgroup
(tools/attn64/sim.py): the committed .inc is what the generator emits, the static hazard check is clean, and the stream computes
soft-max attention (reference attention.py:68-83 = F.scaled_dot_product_attention) under both DMA landing models and two wave orders,
with every load covered by a counted wait and no LDS read racing a DMA.  CPU only: no compute call into the library."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools" / "attn64"))
import gen  # noqa: E402
import sim  # noqa: E402


def test_inc_is_current():
    assert gen.OUT.read_text() == gen.emit_file(), "run python tools/attn64/gen.py --write"


@pytest.mark.parametrize("h16", [False, True])
def test_hazards_and_counts(h16):
    prog = gen.Program(gen.Variant(h16)).build()
    assert gen.check_hazards(prog.ins) == []
    st = gen.stats(prog)  # one loop trip = two 64-key tiles
    assert st["mfma"] == 64 and st["mfma16"] == 16 and st["trans"] == 128 and st["valu"] == 64 and st["ds"] == 48 and st["dma"] == 8
    assert st["barrier"] == 2


@pytest.mark.parametrize("h16,land,order,Lk", [(False, "issue", (0, 1, 2, 3), 192), (False, "wait", (3, 2, 1, 0), 256), (False, "wait", (0, 1, 2, 3), 448),
                                                (True, "issue", (3, 2, 1, 0), 192), (True, "wait", (0, 1, 2, 3), 320)])
def test_stream_computes_attention(h16, land, order, Lk):
    err, wg = sim.run_case(Lk=Lk, h16=h16, land=land, order=order, seed=Lk)
    assert err < (5e-4 if h16 else 4e-3), err
    assert wg.count["mfma"] == 4 * (16 + 32 * (Lk // 64))  # first tile's QK^T + every step's 32 (the last QK^T runs on a clamped tile)


def test_row_strides_and_query_tail():
    err, _ = sim.run_case(Lq=200, Lk=256, ldq=192, ldk=384, ldv=320, seed=7)
    assert err < 4e-3, err


def test_missing_wait_is_caught():
    """The model must notice a consumer in front of its wait: drop one counted wait of the stream."""
    prog = gen.Program(gen.Variant(False)).build()
    k = next(i for i, x in enumerate(prog.ins) if x.kind == "wait" and x.sim[2] not in (None, 0) and i > 400)
    del prog.ins[k]
    import numpy as np
    rng = np.random.default_rng(0)
    gmem = rng.integers(0, 255, 1 << 20, dtype=np.uint8)
    inputs = [sim.wave_inputs(w, 4096, 1 << 18, 1 << 19, 64, 64, 64, 0, 256, 4, False) for w in range(4)]
    with pytest.raises(AssertionError, match="before the wait"):
        sim.Workgroup(prog, gmem, inputs).run()
