"""ctypes binding of libdm4d.so (C ABI declared in include/dm4d.h).

The library is the product's only compute path: there is no CPU or PyTorch fallback.  Loading
fails loudly if the shared object has not been built (``python -m diffuman4d_amd.build``).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent.parent / "libdm4d.so"
_lib = None

_vp, _i, _i64, _f, _u = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint

# name -> (restype, argtypes); mirrors include/dm4d.h exactly (tests/test_abi.py checks the header)
SIGNATURES = {
    "dm4d_version": (_i, []),
    "dm4d_last_error": (C.c_char_p, []),
    "dm4d_gemm_bf16": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _i64, _i, _vp,
                            _i64, _u, _f]),
    "dm4d_conv3x3_nhwc_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp,
                                    _i64, _f]),
    "dm4d_conv3x3_nhwc_bf16_ws": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp,
                                    _i64, _f, _vp, C.c_size_t]),
    "dm4d_conv3x3_nhwc_bf16_flags": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp,
                                          _i64, _f, _u]),
    "dm4d_conv3x3_ws_bytes": (C.c_size_t, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "dm4d_conv2d_direct_nhwc_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "dm4d_conv2d_direct_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "dm4d_groupnorm_ws_bytes": (C.c_size_t, [_i, _i, _i]),
    "dm4d_groupnorm_nhwc_bf16": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "dm4d_layernorm_bf16": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _f]),
    "dm4d_attention_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _f]),
    "dm4d_attention_kv_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _f]),
    "dm4d_attention_qscaled_kv_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i]),
    "dm4d_softmax_rows_bf16": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _f]),
    "dm4d_softmax_rows_f32in_bf16": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _f]),
    "dm4d_timestep_embedding_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _f]),
    "dm4d_silu_bf16": (_i, [_vp, _vp, _vp, _i64]),
    "dm4d_pack_model_input_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "dm4d_cfg_ddim_step_bf16": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _f, _i]),
    "dm4d_cfg_linear_step_bf16": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _f]),
    "dm4d_cfg_multistep_step_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _f]),
    "dm4d_vae_sample_bf16": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _i, _f]),
    "dm4d_scale_pad_bf16": (_i, [_vp, _vp, _i64, _vp, _i, _i64, _i, _f]),
    "dm4d_resize_nchw_f32_to_nhwc_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "dm4d_plucker_latent_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "dm4d_postprocess_images_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "dm4d_resize_aa_nchw_f32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i]),
    "dm4d_conv_up2x_prepare_bf16": (_i, [_vp, _vp, _vp, _i, _i]),
    "dm4d_conv_up2x_nhwc_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "dm4d_ff_geglu_prepare_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    "dm4d_ff_geglu_supported": (_i, [_i, _i]),
    "dm4d_ff_geglu_fused_bf16": (_i, [_vp, _vp, _i64, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i]),
    "dm4d_attn_out_ff_geglu_fused_bf16": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i,
                                                _i]),
    # parity precision (fp32 tensors between kernels, two-term bf16 operands)
    "dm4d_split_f32": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _i64, _i, _vp, _i64, _i64, _i, _i, _f, _i]),
    "dm4d_groupnorm_f32_ws_bytes": (C.c_size_t, [_i, _i, _i]),
    "dm4d_groupnorm_nhwc_f32_split": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "dm4d_layernorm_f32_split": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _f]),
    "dm4d_softmax_rows_f32_split": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _f]),
    "dm4d_attention_split_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _f]),
    "dm4d_timestep_embedding_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _f]),
    "dm4d_pack_model_input_f32_split": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "dm4d_cfg_ddim_step_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _f, _i]),
    "dm4d_cfg_linear_step_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _f]),
    "dm4d_cfg_multistep_step_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _f]),
    "dm4d_vae_sample_f32": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _i, _f]),
    "dm4d_resize_nchw_f32_to_nhwc_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "dm4d_plucker_latent_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "dm4d_postprocess_images_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "dm4d_nhwc_to_nchw_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    # fp16 precision (fp32 tensors between kernels, single-term fp16 MFMA operands)
    "dm4d_gemm_f16": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _i64, _i, _vp, _i64, _u, _f, _i, _f]),
    "dm4d_conv3x3_nhwc_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp, _i64, _f, _u, _vp,
                                   C.c_size_t]),
    "dm4d_conv_up2x_prepare_f16": (_i, [_vp, _vp, _vp, _i, _i]),
    "dm4d_conv_up2x_nhwc_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _u]),
    "dm4d_to_f16_f32": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _i64, _i, _vp, _i64, _i64, _i, _i, _f]),
    "dm4d_groupnorm_nhwc_f32_f16": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "dm4d_groupnorm_nhwc_f16_f16": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "dm4d_groupnorm_nhwc_f32_f16_raw": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _vp]),
    "dm4d_layernorm_f32_f16": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _f]),
    "dm4d_groupnorm_f32_f16_general": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "dm4d_layernorm_f32_f16_general": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _f]),
    "dm4d_softmax_rows_f32_f16": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _f]),
    "dm4d_attention_qscaled_kv_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i]),
    "dm4d_attn_out_ff_geglu_fused_f16": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i,
                                               _i]),
    "dm4d_pack_model_input_f32_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "dm4d_tune_set_gemm_config": (_i, [_i]),
    "dm4d_tune_set_groupnorm_resident": (_i, [_i]),
    "dm4d_nchw_to_nhwc_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "dm4d_nhwc_to_nchw_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
}

EPI_GEGLU = 1
EPI_SILU = 2
EPI_F32OUT = 4
EPI_F32SIDE = 8
EPI_SPLITOUT = 16


class Dm4dError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load libdm4d.so and attach prototypes.  Raises if it is missing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise Dm4dError(
            f"{_LIB_PATH} not found: build the HIP extension first (python -m diffuman4d_amd.build). "
            "diffuman4d_amd has no CPU/PyTorch fallback path."
        )
    # PyTorch ships its own libamdhip64; load it FIRST so that libdm4d.so binds to the same HIP runtime instance as the
    # tensors and streams it is handed (loading libdm4d.so first pulls in /opt/rocm's copy and every launch then
    # fails with "no ROCm-capable device is detected")
    import torch  # noqa: F401
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().dm4d_last_error()
        raise Dm4dError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
