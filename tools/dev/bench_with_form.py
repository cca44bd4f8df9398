"""bench.py with the strip-convolution form forced (tuning hook): python tools/dev/bench_with_form.py <1|2> [bench.py args]"""
import runpy
import sys

sys.path.insert(0, '.')
from diffuman4d_amd.host import lib as L  # noqa: E402

L.load().dm4d_tune_set_strip_form(int(sys.argv[1]))
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
