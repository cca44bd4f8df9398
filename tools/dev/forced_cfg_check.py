"""Parity of forced GEMM / conv kernel configurations (tuning hook): python tools/dev/forced_cfg_check.py 41 42 ..."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from diffuman4d_amd.host import lib as L
import opcheck
lib = L.load()
GEMM = ("gemm_256x128_plain", "gemm_big_tiles", "gemm_n64_tiles", "gemm_mtail_ntail", "gemm_nobias", "gemm_geglu", "gemm_split_a", "gemm_rowbias")
CONV = ("conv_s2", "conv_up", "conv_big", "conv_s1_res")
for cfg in [int(a) for a in sys.argv[1:]]:
    lib.dm4d_tune_set_gemm_config(cfg)
    for n in GEMM + CONV:
        try:
            e, m, t = opcheck.run_case(n)
            print(f"cfg {cfg} {n:22s} {'PASS' if e <= t else 'FAIL'} rel_l2={e:.3e}", flush=True)
        except Exception as ex:  # unsupported shape for this configuration, or a real failure
            print(f"cfg {cfg} {n:22s} ERR {str(ex)[:90]}", flush=True)
lib.dm4d_tune_set_gemm_config(0)
