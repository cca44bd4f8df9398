"""The round-6 strip-convolution experiment (tools/conv64: measured, not shipped -- its README): the generated stream is hazard-clean and
computes the convolution on the numpy workgroup model, for both tiles, odd slab counts, a second row tile and both DMA landing models."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools" / "conv64"))
import cgen  # noqa: E402
import csim  # noqa: E402


def test_inc_is_current_and_hazard_clean():
    assert cgen.OUT.read_text() == cgen.emit_file()
    for name, geo in cgen.CONFIGS.items():
        prog = cgen.ConvProgram(cgen.Cfg(*geo)).build()
        assert cgen.check_hazards(prog.ins) == []


@pytest.mark.parametrize("cfg,cout,kw", [("256X128", 128, dict(B=1, H=6, W=20, Cin=128)), ("256X128", 128, dict(B=2, H=9, W=20, Cin=64, tile=(1, 0), land="wait")),
                                         ("256X160", 320, dict(B=1, H=13, W=20, Cin=192, tile=(0, 1), land="wait", order=(3, 2, 1, 0))),
                                         ("256X160", 160, dict(B=1, H=6, W=20, Cin=64, h16=True, bias=False))])
def test_stream_computes_the_convolution(cfg, cout, kw):
    err, _ = csim.run_case(cfg, Cout=cout, **kw)
    assert err < 2e-6, err
