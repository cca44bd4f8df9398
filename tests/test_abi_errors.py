"""CPU: argument validation of the C ABI.  Every entry point returns DM4D_ERR_ARG (-1) and leaves a message in the
thread-local dm4d_last_error() BEFORE it touches the device, so these calls are safe without a GPU (the pointers are
fake, suitably aligned addresses that are never dereferenced)."""
import ctypes
import threading

import pytest

P = 0x10000  # a non-null, 16-byte aligned "device pointer"
ERR_ARG = -1


@pytest.fixture(scope="module")
def lib():
    from diffuman4d_amd.host import lib as L
    return L.load()


def last(lib):
    return lib.dm4d_last_error().decode()


def test_gemm_rejects_bad_shapes(lib):
    g = lib.dm4d_gemm_bf16
    # A, lda, A2, lda2, K1, W, ldw, C, ldc, M, N, K, bias, rowbias, ld_rb, rows_per_rb, residual, ld_res, flags, scale
    assert g(None, None, 320, None, 0, 0, P, 320, P, 320, 64, 320, 320, None, None, 0, 1, None, 0, 0, 1.0) == ERR_ARG
    assert "null pointer" in last(lib)
    assert g(None, P, 320, None, 0, 0, P, 320, P, 320, 64, 320, 33, None, None, 0, 1, None, 0, 0, 1.0) == ERR_ARG
    assert "multiple of 32" in last(lib)
    assert g(None, P, 324, None, 0, 0, P, 320, P, 320, 64, 320, 320, None, None, 0, 1, None, 0, 0, 1.0) == ERR_ARG
    assert "multiples of 8" in last(lib)
    assert g(None, P, 320, P, 320, 100, P, 640, P, 320, 64, 320, 640, None, None, 0, 1, None, 0, 0, 1.0) == ERR_ARG
    assert "split-A" in last(lib)
    assert g(None, P, 320, None, 0, 0, P, 320, P, 320, 64, 320, 320, None, P, 320, 0, None, 0, 0, 1.0) == ERR_ARG
    assert "rows_per_rowbias" in last(lib)


def test_conv_rejects_bad_geometry(lib):
    c = lib.dm4d_conv3x3_nhwc_bf16
    # X, B, H, W, Cin, Wt, Y, Ho, Wo, Cout, stride, pad, upsample, bias, rowbias, ld_rb, residual, ld_res, scale
    assert c(None, P, 2, 9, 5, 30, P, P, 9, 5, 64, 1, 1, 0, None, None, 0, None, 0, 1.0) == ERR_ARG
    assert "multiple of 32" in last(lib)
    assert c(None, P, 2, 9, 5, 32, P, P, 9, 5, 64, 3, 1, 0, None, None, 0, None, 0, 1.0) == ERR_ARG
    assert "stride" in last(lib)
    assert c(None, P, 2, 9, 5, 32, P, P, 18, 10, 64, 2, 1, 1, None, None, 0, None, 0, 1.0) == ERR_ARG
    assert "upsample" in last(lib)
    assert c(None, P, 1 << 12, 1 << 10, 1 << 10, 32, P, P, 1 << 10, 1 << 10, 64, 1, 1, 0, None, None, 0, None, 0, 1.0) == ERR_ARG
    assert "2^31" in last(lib)
    assert c(None, None, 2, 9, 5, 32, P, P, 9, 5, 64, 1, 1, 0, None, None, 0, None, 0, 1.0) == ERR_ARG


def test_attention_rejects_bad_layouts(lib):
    a = lib.dm4d_attention_kv_bf16
    # Q, K, V, O, ldq, ldk, ldv, ldo, batch, heads, Lq, Lk, scale
    assert a(None, P, P, P, P, 192, 192, 192, 64, 1, 1, 0, 64, 0.125) == ERR_ARG
    assert "empty shape" in last(lib)
    assert a(None, P, P, P, P, 190, 192, 192, 64, 1, 1, 64, 64, 0.125) == ERR_ARG
    assert "multiples of 8" in last(lib)
    assert a(None, P + 2, P, P, P, 192, 192, 192, 64, 1, 1, 64, 64, 0.125) == ERR_ARG
    assert "16-byte aligned" in last(lib)
    assert a(None, P, P, P, P, 192, 1 << 24, 192, 64, 1, 1, 64, 64, 0.125) == ERR_ARG
    assert "out of range" in last(lib)
    q = lib.dm4d_attention_qscaled_kv_bf16
    assert q(None, P, None, P, P, 192, 192, 192, 64, 1, 1, 64, 64) == ERR_ARG


def test_norms_reject_bad_channel_counts(lib):
    gn = lib.dm4d_groupnorm_nhwc_bf16
    # X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, Y, silu, ws
    assert gn(None, P, 12, None, 0, 2, 45, 4, 1e-5, P, P, P, 0, P) == ERR_ARG
    assert "multiples of 8" in last(lib)
    assert gn(None, P, 64, None, 0, 2, 45, 5, 1e-5, P, P, P, 0, P) == ERR_ARG  # 64 % 5 != 0
    assert gn(None, P, 64, None, 0, 2, 45, 8, 1e-5, P, P, P, 0, None) == ERR_ARG  # no workspace
    ln = lib.dm4d_layernorm_bf16
    # X, ldx, gamma, beta, Y, ldy, M, C, eps
    assert ln(None, P, 12, P, P, P, 12, 4, 12, 1e-5) == ERR_ARG
    assert "multiple of 8" in last(lib)


def test_last_error_is_thread_local(lib):
    assert lib.dm4d_layernorm_bf16(None, P, 12, P, P, P, 12, 4, 12, 1e-5) == ERR_ARG
    mine = last(lib)
    seen = []

    def other():
        lib.dm4d_silu_bf16(None, None, None, 0)
        seen.append(last(lib))

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert "silu" in seen[0]
    assert last(lib) == mine  # the other thread's failure did not overwrite this thread's message


def test_scheduler_steps_reject_bad_arguments(lib):
    lin = lib.dm4d_cfg_linear_step_bf16
    # latents, x0_prev, noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale
    assert lin(None, P, None, P, 4, P, P, None, 4, 45, 1, 2.0) == ERR_ARG  # a multistep scheduler has state: x0_prev is required
    assert "cfg_linear_step" in last(lib)
    assert lin(None, P, P, P, 3, P, P, None, 4, 45, 1, 2.0) == ERR_ARG     # fewer than four prediction channels per row
    assert lin(None, P, P, P, 4, P, P, None, 0, 45, 1, 2.0) == ERR_ARG
    ddim = lib.dm4d_cfg_ddim_step_bf16
    assert ddim(None, P, P, 4, None, P, None, 4, 45, 1, 2.0, 0) == ERR_ARG
    assert "cfg_ddim_step" in last(lib)
