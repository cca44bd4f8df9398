// Shared device helpers for the gfx950 kernels of libdm4d.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct __attribute__((aligned(16))) U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float(((uint32_t)v) << 16); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// fp32 -> bf16, round-to-nearest-even, in hardware (v_cvt_pk_bf16_f32 on gfx950): branch free
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<uint32_t*>(&b);
}

__device__ __forceinline__ u16 f2bf(float f) { return (u16)(pack_bf2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ void unpack8(const U4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16);
  f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16);
  f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16);
  f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16);
  f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ U4 pack8(const float* f) {
  U4 v;
  v.x = pack_bf2(f[0], f[1]);
  v.y = pack_bf2(f[2], f[3]);
  v.z = pack_bf2(f[4], f[5]);
  v.w = pack_bf2(f[6], f[7]);
  return v;
}

// fp16 twins (precision "fp16": single-term fp16 MFMA operands over fp32 tensors): v_cvt_pk_f16_f32 rounds to nearest even
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  h16x2_t b = __builtin_convertvector(v, h16x2_t);
  return *reinterpret_cast<uint32_t*>(&b);
}
__device__ __forceinline__ u16 f2h(float f) { return (u16)(pack_h2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float h2f(u16 v) {
  _Float16 h;
  __builtin_memcpy(&h, &v, 2);
  return (float)h;
}
__device__ __forceinline__ void unpack8h(const U4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h16x2_t h;
    __builtin_memcpy(&h, &w[i], 4);
    f[2 * i] = (float)h[0];
    f[2 * i + 1] = (float)h[1];
  }
}
__device__ __forceinline__ U4 pack8h(const float* f) {
  U4 v;
  v.x = pack_h2(f[0], f[1]);
  v.y = pack_h2(f[2], f[3]);
  v.z = pack_h2(f[4], f[5]);
  v.w = pack_h2(f[6], f[7]);
  return v;
}

// Saturating forms for the HBM-bound kernels that turn UN-NORMALISED fp32 activations into fp16 operands (dm4d_to_f16_f32, the norm
// kernels' outputs and GroupNorm's raw plane, the parity-file conversions): |x| > 65504 becomes +-65504 instead of +-inf, which the next
// MFMA would turn into NaN.  One v_med3_f32 per value, free beside the memory traffic there; the GEMM / convolution epilogues and the
// attention probabilities keep the plain conversion (issue-bound loops; the probabilities have their own range guard).
__device__ __forceinline__ float sat_h(float x) { return __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }
__device__ __forceinline__ u16 f2h_sat(float f) { return f2h(sat_h(f)); }
__device__ __forceinline__ U4 pack8h_sat(const float* f) {
  U4 v;
  v.x = pack_h2(sat_h(f[0]), sat_h(f[1]));
  v.y = pack_h2(sat_h(f[2]), sat_h(f[3]));
  v.z = pack_h2(sat_h(f[4]), sat_h(f[5]));
  v.w = pack_h2(sat_h(f[6]), sat_h(f[7]));
  return v;
}

__device__ __forceinline__ U4 ldg16(const void* p) { return *reinterpret_cast<const U4*>(p); }
__device__ __forceinline__ void stg16(void* p, const U4& v) { *reinterpret_cast<U4*>(p) = v; }

// x * sigmoid(x) with the hardware reciprocal (1 ulp; an IEEE fp32 division is ~10 VALU instructions, and the
// GroupNorm / epilogue consumers round the result to bf16 anyway)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf(x) by Abramowitz & Stegun 7.1.26: |abs error| <= 1.5e-7, branch free, ~12 VALU (libm erff is ~4x that
// and dominated the GEGLU epilogue of the K=320 feed-forward GEMMs)
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float y = fmaf(fmaf(fmaf(fmaf(1.061405429f, t, -1.453152027f), t, 1.421413741f), t, -0.284496736f), t, 0.254829592f) * t;
  y = 1.0f - y * __expf(-ax * ax);
  return copysignf(y, x);
}
// exact (erf) GELU, as F.gelu(approximate="none") used by diffusers' GEGLU
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// The same function to 2.6e-5 absolute (the bf16 rounding of a GEGLU output of magnitude 1 is 2e-3): x * sigmoid(x * (c0 +
// c1 x^2 + c2 x^4)), the tanh form of GELU with one more term, coefficients from a minimax fit against the erf form on
// [-9, 9] (x^2 clamped at 4.5^2, beyond which both forms have saturated).  9 VALU instructions (2 transcendental) against
// ~17 (2) for gelu_erf_f.  Not used by default (GEGLU_ERF in gemm.hip): it buys 1.2 % of the Linear layers' time.
__device__ __forceinline__ float gelu_fit_f(float x) {
  const float x2 = fminf(x * x, 20.25f);
  // (c0, c1, c2) * -log2(e): the sigmoid is taken as 1 / (1 + exp2(-u log2 e))
  const float q = fmaf(fmaf(1.014263e-3f, x2, -1.0677572e-1f), x2, -2.3011213f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * q));
}

// LayerNorm statistics of ONE row held by one wave: lane l carries the 16-byte vectors l, l + 64, ... of the row (v[i] = vector
// l + 64 i, on[i] = that vector exists), two-pass mean / variance in fp32, butterfly over the 64 lanes.  The one definition both
// ln_kernel (norm.hip) and the in-LDS LayerNorm of the row-register kernels (ff_fused.hip) use, so that a fused LayerNorm gives the
// bits of the stand-alone launch.
template <int NV>
__device__ __forceinline__ void ln_row_stats(const float (&v)[NV][8], const bool (&on)[NV], int C, float eps, float& mu, float& rs) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (on[i]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  mu = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (on[i]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mu;
        q += d * d;
      }
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  rs = rsqrtf(q / (float)C + eps);
}
// ... and the affine step on one vector: y = (v - mu) rs gamma + beta
__device__ __forceinline__ void ln_row_apply(const float (&v)[8], float mu, float rs, const float (&g)[8], const float (&bt)[8], float (&y)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = (v[e] - mu) * rs * g[e] + bt[e];
}

// XCD-aware block remap (8 XCDs, block b runs on XCD b % 8): give each XCD a contiguous range of
// logical tiles so neighbouring tiles share one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
