// Parity-precision kernels for gfx950 (host/ops.py, precision = "parity"): the same path with fp32 tensors between kernels and
// TWO-TERM bf16 operands on the matrix unit.
//
// Why: BASELINE.json's north_star asks for decoded RGB within 1e-3 rel-L2 of the reference's fp32 CPU path.  The fast path cannot
// get there -- rounding the MFMA operands to bf16 alone costs 7e-3 on the judged UNet call (profiles/r03_error_budget.log) -- and
// gfx950 has neither an xf32 mode nor a fast fp32 MFMA.  A bf16/fp16 checkpoint's WEIGHTS are exact in bf16, so only the
// activations need more bits: x = hi + lo with hi = bf16(x), lo = bf16(x - hi) carries 16 mantissa bits (|x - hi - lo| <= 2^-17 |x|),
// and  x W^T = [hi | lo] [W | W]^T  is one more pass of the SAME kernels over a K axis twice as long (dm4d_gemm_bf16 /
// dm4d_conv3x3_nhwc_bf16_flags with DM4D_EPI_F32OUT / F32SIDE / SPLITOUT).  This file holds what sits between those launches:
//   * split_kernel            fp32 [M, C] (one or two sources = the up-block channel concat) -> operand [M, hi(Cp) | lo(Cp)]
//   * gn32_*                  GroupNorm(+SiLU) fp32 in, statistics in fp64, operand out
//   * ln32_kernel             LayerNorm fp32 in, operand out
//   * softmax32_split_kernel  row softmax of fp32 logits -> three-plane operand [p_hi | p_lo | p_hi] (VAE mid block: both factors
//                             of q k^T and of p v are activations, so each product takes three terms hi hi + lo hi + hi lo)
//   * attn_split_kernel       the d = 64 flash attention with three MFMAs per product (Kh Qh + Kh Ql + Kl Qh; Vh Ph + Vl Ph + Vh Pl),
//                             exact running-max softmax in fp32, operand out
// None of these is tuned: the parity mode exists to PROVE the arithmetic of the path (tests/modelcheck.py --precision parity), its
// cost is reported beside the fast mode's (bench.py `parity`).
#include "common.h"
#include "dm4d.h"
#include "errors.h"

namespace {

__device__ __forceinline__ void split2(float x, u16& hi, u16& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

inline dim3 grid1d(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

// ------------------------------------------------------------------------------------------------
// fp32 -> two-term operand
// ------------------------------------------------------------------------------------------------
struct SplitParams {
  const float* X1;
  int64_t rs1, cs1;  // row / column stride of X1 in floats (cs1 != 1: a transposed read, V^T of the VAE mid block)
  const float* X2;
  int64_t rs2;
  int C1, C2, Cp;
  int64_t M;
  u16* Y;
  int64_t ldy;
  int act;      // 1 = SiLU
  float scale;  // applied after the activation
  int pattern;  // 0: [hi | lo]   1: [hi | lo | hi]   2: [hi | hi | lo]   3: one fp16 plane (precision "fp16")
};

__global__ __launch_bounds__(256) void split_kernel(SplitParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M * p.Cp) return;
  const int64_t m = i / p.Cp;
  const int c = (int)(i - m * p.Cp);
  float x = 0.f;
  if (c < p.C1) x = p.X1[m * p.rs1 + (int64_t)c * p.cs1];
  else if (c < p.C1 + p.C2) x = p.X2[m * p.rs2 + (c - p.C1)];
  if (p.act == 1) x = silu_f(x);
  x *= p.scale;
  u16* y = p.Y + m * p.ldy + c;
  if (p.pattern == 3) {
    y[0] = f2h_sat(x);
    return;
  }
  u16 hi, lo;
  split2(x, hi, lo);
  y[0] = hi;
  if (p.pattern == 0) {
    y[p.Cp] = lo;
  } else if (p.pattern == 1) {
    y[p.Cp] = lo;
    y[2 * p.Cp] = hi;
  } else {
    y[p.Cp] = hi;
    y[2 * p.Cp] = lo;
  }
}

// precision "fp16", the plain case (round 6): contiguous channels, every count a multiple of 8, 16-byte aligned rows -- eight channels per
// thread, 2 x 16 bytes in, 16 bytes out (the per-element kernel above moved 1.6 TB/s on the 72 x 40 level's tensors; same values:
// act, scale, one saturating rounding)
__global__ __launch_bounds__(256) void to_f16_vec8_kernel(SplitParams p) {
  const int CV = p.Cp >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.M * CV) return;
  const int64_t m = i / CV;
  const int c = (int)(i - m * CV) << 3;
  float v[8];
  if (c < p.C1 + p.C2) {
    const float* src = c < p.C1 ? p.X1 + m * p.rs1 + c : p.X2 + m * p.rs2 + (c - p.C1);
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src), b = *reinterpret_cast<const f32x4_t*>(src + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float x = v[e];
    if (p.act == 1) x = silu_f(x);
    v[e] = x * p.scale;
  }
  stg16(p.Y + m * p.ldy + c, pack8h_sat(v));
}

// ------------------------------------------------------------------------------------------------
// GroupNorm (+SiLU), fp32 NHWC in (two sources = channel concat), operand out.  Sums in fp64: no shift trick needed.
// ------------------------------------------------------------------------------------------------
constexpr int GN_THREADS = 256;
constexpr int GN_MAXC = 4096;

__host__ __device__ inline int gn32_nchunk(int B, int HW) {
  int n = (2048 + B - 1) / B;
  int cap = (HW + 15) / 16;
  if (n > cap) n = cap;
  if (n < 1) n = 1;
  return n;
}

struct GN32Params {
  const float* X1;
  const float* X2;
  int C1, C2, B, HW, groups, nchunk;
  float eps;
  const u16* gamma;
  const u16* beta;
  u16* Y;  // [B*HW, 2 C]
  int silu;
  double* ws;  // [B][nchunk][groups][2]
};

__global__ __launch_bounds__(GN_THREADS) void gn32_stats_kernel(GN32Params p) {
  extern __shared__ __attribute__((aligned(16))) double smd[];  // [PPB][C][2]
  const int C = p.C1 + p.C2;
  const int b = blockIdx.x / p.nchunk, chunk = blockIdx.x % p.nchunk;
  const int per = (p.HW + p.nchunk - 1) / p.nchunk;
  const int p0 = chunk * per, p1 = min(p0 + per, p.HW);
  const int PPB = C >= GN_THREADS ? 1 : GN_THREADS / C;
  const int tid = threadIdx.x;
  for (int slot = tid; slot < C * PPB; slot += GN_THREADS) {
    const int c = slot % C, prow = slot / C;
    const float* src;
    int ld;
    if (c < p.C1) {
      src = p.X1 + (int64_t)b * p.HW * p.C1 + c;
      ld = p.C1;
    } else {
      src = p.X2 + (int64_t)b * p.HW * p.C2 + (c - p.C1);
      ld = p.C2;
    }
    double s = 0.0, q = 0.0;
    for (int px = p0 + prow; px < p1; px += PPB) {
      const double v = (double)src[(int64_t)px * ld];
      s += v;
      q += v * v;
    }
    smd[(prow * C + c) * 2 + 0] = s;
    smd[(prow * C + c) * 2 + 1] = q;
  }
  __syncthreads();
  const int gs = C / p.groups;
  for (int g = tid; g < p.groups; g += GN_THREADS) {
    double s = 0.0, q = 0.0;
    for (int r = 0; r < PPB; ++r)
      for (int c = g * gs; c < (g + 1) * gs; ++c) {
        s += smd[(r * C + c) * 2 + 0];
        q += smd[(r * C + c) * 2 + 1];
      }
    double* w = p.ws + (((int64_t)b * p.nchunk + chunk) * p.groups + g) * 2;
    w[0] = s;
    w[1] = q;
  }
}

// H16 (precision "fp16"): gamma / beta are fp16 and the result is ONE fp16 plane, Y [B*HW, C].
template <bool H16>
__global__ __launch_bounds__(GN_THREADS) void gn32_apply_kernel(GN32Params p) {
  extern __shared__ __attribute__((aligned(16))) float smf[];  // mean[groups], rstd[groups]
  const int C = p.C1 + p.C2;
  float* mean = smf;
  float* rstd = smf + p.groups;
  const int b = blockIdx.x / p.nchunk, chunk = blockIdx.x % p.nchunk;
  const int per = (p.HW + p.nchunk - 1) / p.nchunk;
  const int p0 = chunk * per, p1 = min(p0 + per, p.HW);
  const int tid = threadIdx.x;
  const int gs = C / p.groups;
  for (int g = tid; g < p.groups; g += GN_THREADS) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < p.nchunk; ++k) {
      const double* w = p.ws + (((int64_t)b * p.nchunk + k) * p.groups + g) * 2;
      s += w[0];
      q += w[1];
    }
    const double n = (double)gs * (double)p.HW;
    const double mu = s / n;
    double var = q / n - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    mean[g] = (float)mu;
    rstd[g] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  const int PPB = C >= GN_THREADS ? 1 : GN_THREADS / C;
  for (int slot = tid; slot < C * PPB; slot += GN_THREADS) {
    const int c = slot % C, prow = slot / C;
    const int g = c / gs;
    const float mu = mean[g], a = rstd[g] * (H16 ? h2f(p.gamma[c]) : bf2f(p.gamma[c])), bt = H16 ? h2f(p.beta[c]) : bf2f(p.beta[c]);
    const float* src;
    int ld;
    if (c < p.C1) {
      src = p.X1 + (int64_t)b * p.HW * p.C1 + c;
      ld = p.C1;
    } else {
      src = p.X2 + (int64_t)b * p.HW * p.C2 + (c - p.C1);
      ld = p.C2;
    }
    constexpr int PL = H16 ? 1 : 2;
    u16* dst = p.Y + (int64_t)b * p.HW * (PL * C) + c;
    for (int px = p0 + prow; px < p1; px += PPB) {
      float y = (src[(int64_t)px * ld] - mu) * a + bt;
      if (p.silu) y = silu_f(y);
      if constexpr (H16) {
        dst[(int64_t)px * C] = f2h_sat(y);
      } else {
        u16 hi, lo;
        split2(y, hi, lo);
        dst[(int64_t)px * (2 * C)] = hi;
        dst[(int64_t)px * (2 * C) + C] = lo;
      }
    }
  }
}

// The same two passes with FOUR channels per thread (16-byte loads, 8-byte stores of each plane): what every layer of the UNet and the VAE
// runs (C1, C2 multiples of 4).  With one channel per thread a C = 320 layer kept a quarter of the block idle in its second sweep and moved
// 2 bytes per store instruction: 1.15 TB/s over a step (round 4 bench); the sums are still fp64, so the statistics differ from the
// one-channel form by fp64 rounding only.
__global__ __launch_bounds__(GN_THREADS) void gn32_stats4_kernel(GN32Params p) {
  extern __shared__ __attribute__((aligned(16))) double smd[];  // [PPB][C][2]
  const int C = p.C1 + p.C2, Q = C / 4;
  const int b = blockIdx.x / p.nchunk, chunk = blockIdx.x % p.nchunk;
  const int per = (p.HW + p.nchunk - 1) / p.nchunk;
  const int p0 = chunk * per, p1 = min(p0 + per, p.HW);
  const int PPB = Q >= GN_THREADS ? 1 : GN_THREADS / Q;
  const int tid = threadIdx.x;
  for (int slot = tid; slot < Q * PPB; slot += GN_THREADS) {
    const int c = (slot % Q) * 4, prow = slot / Q;
    const float* src;
    int ld;
    if (c < p.C1) {
      src = p.X1 + (int64_t)b * p.HW * p.C1 + c;
      ld = p.C1;
    } else {
      src = p.X2 + (int64_t)b * p.HW * p.C2 + (c - p.C1);
      ld = p.C2;
    }
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    for (int px = p0 + prow; px < p1; px += PPB) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + (int64_t)px * ld);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double d = (double)v[e];
        s[e] += d;
        q[e] += d * d;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      smd[(prow * C + c + e) * 2 + 0] = s[e];
      smd[(prow * C + c + e) * 2 + 1] = q[e];
    }
  }
  __syncthreads();
  const int gs = C / p.groups;
  for (int g = tid; g < p.groups; g += GN_THREADS) {
    double s = 0.0, q = 0.0;
    for (int r = 0; r < PPB; ++r)
      for (int c = g * gs; c < (g + 1) * gs; ++c) {
        s += smd[(r * C + c) * 2 + 0];
        q += smd[(r * C + c) * 2 + 1];
      }
    double* w = p.ws + (((int64_t)b * p.nchunk + chunk) * p.groups + g) * 2;
    w[0] = s;
    w[1] = q;
  }
}

template <bool H16>
__global__ __launch_bounds__(GN_THREADS) void gn32_apply4_kernel(GN32Params p) {
  extern __shared__ __attribute__((aligned(16))) float smf[];  // mean[groups], rstd[groups]
  const int C = p.C1 + p.C2, Q = C / 4;
  float* mean = smf;
  float* rstd = smf + p.groups;
  const int b = blockIdx.x / p.nchunk, chunk = blockIdx.x % p.nchunk;
  const int per = (p.HW + p.nchunk - 1) / p.nchunk;
  const int p0 = chunk * per, p1 = min(p0 + per, p.HW);
  const int tid = threadIdx.x;
  const int gs = C / p.groups;
  for (int g = tid; g < p.groups; g += GN_THREADS) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < p.nchunk; ++k) {
      const double* w = p.ws + (((int64_t)b * p.nchunk + k) * p.groups + g) * 2;
      s += w[0];
      q += w[1];
    }
    const double n = (double)gs * (double)p.HW;
    const double mu = s / n;
    double var = q / n - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    mean[g] = (float)mu;
    rstd[g] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  const int PPB = Q >= GN_THREADS ? 1 : GN_THREADS / Q;
  for (int slot = tid; slot < Q * PPB; slot += GN_THREADS) {
    const int c = (slot % Q) * 4, prow = slot / Q;
    float mu[4], a[4], bt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = (c + e) / gs;
      mu[e] = mean[g];
      a[e] = rstd[g] * (H16 ? h2f(p.gamma[c + e]) : bf2f(p.gamma[c + e]));
      bt[e] = H16 ? h2f(p.beta[c + e]) : bf2f(p.beta[c + e]);
    }
    const float* src;
    int ld;
    if (c < p.C1) {
      src = p.X1 + (int64_t)b * p.HW * p.C1 + c;
      ld = p.C1;
    } else {
      src = p.X2 + (int64_t)b * p.HW * p.C2 + (c - p.C1);
      ld = p.C2;
    }
    constexpr int PL = H16 ? 1 : 2;
    u16* dst = p.Y + (int64_t)b * p.HW * (PL * C) + c;
    for (int px = p0 + prow; px < p1; px += PPB) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + (int64_t)px * ld);
      if constexpr (H16) {
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = (v[e] - mu[e]) * a[e] + bt[e];
          if (p.silu) y[e] = silu_f(y[e]);
        }
        uint2 ph;
        ph.x = pack_h2(sat_h(y[0]), sat_h(y[1])), ph.y = pack_h2(sat_h(y[2]), sat_h(y[3]));
        *reinterpret_cast<uint2*>(dst + (int64_t)px * C) = ph;
        continue;
      }
      u16 hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = (v[e] - mu[e]) * a[e] + bt[e];
        if (p.silu) y = silu_f(y);
        split2(y, hi[e], lo[e]);
      }
      uint2 ph, pl;
      ph.x = (uint32_t)hi[0] | ((uint32_t)hi[1] << 16), ph.y = (uint32_t)hi[2] | ((uint32_t)hi[3] << 16);
      pl.x = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16), pl.y = (uint32_t)lo[2] | ((uint32_t)lo[3] << 16);
      *reinterpret_cast<uint2*>(dst + (int64_t)px * (2 * C)) = ph;
      *reinterpret_cast<uint2*>(dst + (int64_t)px * (2 * C) + C) = pl;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm, fp32 in, operand out: one wave per row, two-pass statistics in fp32
// ------------------------------------------------------------------------------------------------
template <bool H16>
__global__ __launch_bounds__(256) void ln32_kernel(const float* X, int64_t ldx, const u16* gamma, const u16* beta, u16* Y, int64_t ldy,
                                                   int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* x = X + (int64_t)row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mu = s / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = x[c] - mu;
    q += d * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rs = rsqrtf(q / (float)C + eps);
  u16* y = Y + (int64_t)row * ldy;
  for (int c = lane; c < C; c += 64) {
    if constexpr (H16) {
      y[c] = f2h_sat((x[c] - mu) * rs * h2f(gamma[c]) + h2f(beta[c]));
    } else {
      const float v = (x[c] - mu) * rs * bf2f(gamma[c]) + bf2f(beta[c]);
      u16 hi, lo;
      split2(v, hi, lo);
      y[c] = hi;
      y[C + c] = lo;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// P = softmax(S * scale) per row of fp32 logits -> [p_hi | p_lo | p_hi], each plane Np wide (columns N..Np-1 zero)
// ------------------------------------------------------------------------------------------------
template <bool H16>
__global__ __launch_bounds__(256) void softmax32_split_kernel(const float* S, int64_t lds, u16* P, int64_t ldp, int M, int N, int Np,
                                                              float scale) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* s = S + (int64_t)row * lds;
  float mx = -3.0e38f;
  for (int c = tid; c < N; c += 256) mx = fmaxf(mx, s[c]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = tid; c < N; c += 256) sum += expf((s[c] - mx) * scale);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.0f / sum;
  u16* y = P + (int64_t)row * ldp;
  for (int c = tid; c < Np; c += 256) {
    if constexpr (H16) {  // one fp16 plane [M, Np]
      y[c] = c < N ? f2h(expf((s[c] - mx) * scale) * inv) : (u16)0;
    } else {
      u16 hi = 0, lo = 0;
      if (c < N) split2(expf((s[c] - mx) * scale) * inv, hi, lo);
      y[c] = hi;
      y[Np + c] = lo;
      y[2 * Np + c] = hi;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Self-attention, head_dim 64, two-term Q / K / V (hi plane at the pointer, lo plane `*_lo` elements behind it), exact
// running-max softmax in fp32, two-term P, operand out.  Loop structure, LDS layouts and the register <-> key mapping are those
// of attention.hip's exact loop (kv_loop<SAFE>): S^T = K Q^T with lane (q = lane & 31) holding 16 of a 32-key block's scores,
// O^T = V^T P^T with V delivered by transposing LDS reads in the accumulator's key order.
// ------------------------------------------------------------------------------------------------
struct AttnSplitParams {
  const u16 *Q, *K, *V;
  u16* O;
  int64_t ldq, ldk, ldv, ldo;
  int64_t q_lo, k_lo, v_lo, o_lo;
  int L, Lk, heads, nqt;
  float c;  // scale * log2(e)
};

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int KV = 64;
constexpr int LDK = 72;  // K rows: 64 + 8 pad bf16
constexpr int LDV = 96;  // V rows: 64 + 32 pad bf16
constexpr int LDO = 72;
constexpr float RESCALE_THR = 8.0f;

__device__ __forceinline__ bf16x8_t pack_frag(const float* v) {
  U4 w = pack8(v);
  return *reinterpret_cast<bf16x8_t*>(&w);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_split_kernel(AttnSplitParams p) {
  constexpr int KS = 2 * KV * LDK, VS = 2 * KV * LDV;  // elements of one double-buffered plane
  __shared__ __attribute__((aligned(16))) u16 smem[2 * KS + 2 * VS];
  u16* Ks[2] = {smem, smem + KS};                    // [plane][buf][64][LDK]
  u16* Vs[2] = {smem + 2 * KS, smem + 2 * KS + VS};  // [plane][buf][64][LDV]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int L = p.L, Lk = p.Lk;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = lid % p.nqt, bh = lid / p.nqt;
  const int head = bh % p.heads, batch = bh / p.heads;
  const int q_tile0 = qt * (NW * 32) + wave * 32;
  const u16* Qb = p.Q + (int64_t)batch * L * p.ldq + head * 64;
  const u16* Kb = p.K + (int64_t)batch * Lk * p.ldk + head * 64;
  const u16* Vb = p.V + (int64_t)batch * Lk * p.ldv + head * 64;
  u16* Ob = p.O + (int64_t)batch * L * p.ldo + head * 64;

  bf16x8_t qf[2][4];
  {
    int q = q_tile0 + l31;
    if (q > L - 1) q = L - 1;
    const u16* qp = Qb + (int64_t)q * p.ldq + lh * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      U4 v = ldg16(qp + j * 16);
      qf[0][j] = *reinterpret_cast<bf16x8_t*>(&v);
      U4 w = ldg16(qp + p.q_lo + j * 16);
      qf[1][j] = *reinterpret_cast<bf16x8_t*>(&w);
    }
  }
  f32x16_t o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float cs = p.c;

  constexpr int RPT = 8 / NW;  // key rows per thread and tile
  static_assert(RPT >= 1, "at most 8 waves");
  U4 rk[2][RPT], rv[2][RPT];
  const int s_key = tid >> 3, s_c = tid & 7;
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      int key = t * KV + s_key + i * (NW * 8);
      if (key > Lk - 1) key = Lk - 1;
      const u16* kp = Kb + (int64_t)key * p.ldk + s_c * 8;
      const u16* vp = Vb + (int64_t)key * p.ldv + s_c * 8;
      rk[0][i] = ldg16(kp);
      rk[1][i] = ldg16(kp + p.k_lo);
      rv[0][i] = ldg16(vp);
      rv[1][i] = ldg16(vp + p.v_lo);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        *reinterpret_cast<U4*>(Ks[pl] + (buf * KV + s_key + i * (NW * 8)) * LDK + s_c * 8) = rk[pl][i];
        *reinterpret_cast<U4*>(Vs[pl] + (buf * KV + s_key + i * (NW * 8)) * LDV + s_c * 8) = rv[pl][i];
      }
  };
  // V fragment base (elements inside a plane): row 4 lh + ((lane & 15) >> 2), column 16 ((lane >> 4) & 1) + 4 (lane & 3)
  const int v_lane = (4 * lh + ((lane & 15) >> 2)) * LDV + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  const int nt = (Lk + KV - 1) / KV;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) load_tile(t + 1);
    f32x16_t s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int off = (buf * KV + kb * 32 + l31) * LDK + j * 16 + lh * 8;
        const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(Ks[0] + off);
        const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(Ks[1] + off);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][j], s[kb], 0, 0, 0);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[1][j], s[kb], 0, 0, 0);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[0][j], s[kb], 0, 0, 0);
      }
    if ((t == nt - 1) && (Lk % KV) != 0) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int key0 = t * KV + kb * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + (r & 3) + 8 * (r >> 2) >= Lk) s[kb][r] = -1e30f;
      }
    }
    {
      float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (__any((mx - m_run) * cs > RESCALE_THR)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cs);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      }
    }
    const float mc = m_run * cs;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float ph[16], pl[16];
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = s[kb][r] <= -1e29f ? 0.f : __builtin_amdgcn_exp2f(s[kb][r] * cs - mc);
        sum += pv;
        ph[r] = bf2f(f2bf(pv));
        pl[r] = pv - ph[r];
      }
      l_run += sum;
      bf16x8_t pfh[2], pfl[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        pfh[jj] = pack_frag(ph + jj * 8);
        pfl[jj] = pack_frag(pl + jj * 8);
      }
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int off = v_lane + (buf * KV + kb * 32 + jj * 16) * LDV + db * 32;
          bf16x8_t vf[2];
#pragma unroll
          for (int pln = 0; pln < 2; ++pln) {
            const u16* vp = Vs[pln] + off;
            s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)vp);
            s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vp + 8 * LDV));
            s16x8_t v01 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
            vf[pln] = *reinterpret_cast<bf16x8_t*>(&v01);
          }
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pfh[jj], o[db], 0, 0, 0);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pfl[jj], o[db], 0, 0, 0);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pfh[jj], o[db], 0, 0, 0);
        }
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  // O through two wave-private LDS tiles (hi, lo), then whole 128-byte rows out; the loop ended with a workgroup barrier
  u16* Oh = smem + wave * (32 * LDO);
  u16* Ol = smem + (NW + wave) * (32 * LDO);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4], h[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = o[db][4 * g + e] * inv;
        h[e] = bf2f(f2bf(v[e]));
      }
      uint2 wh, wl;
      wh.x = pack_bf2(h[0], h[1]);
      wh.y = pack_bf2(h[2], h[3]);
      wl.x = pack_bf2(v[0] - h[0], v[1] - h[1]);
      wl.y = pack_bf2(v[2] - h[2], v[3] - h[3]);
      *reinterpret_cast<uint2*>(Oh + l31 * LDO + db * 32 + 8 * g + 4 * lh) = wh;
      *reinterpret_cast<uint2*>(Ol + l31 * LDO + db * 32 + 8 * g + 4 * lh) = wl;
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = 8 * k + (lane >> 3), ch = lane & 7;
    const u32x4_t vh = *reinterpret_cast<const u32x4_t*>(Oh + row * LDO + ch * 8);
    const u32x4_t vl = *reinterpret_cast<const u32x4_t*>(Ol + row * LDO + ch * 8);
    const int q = q_tile0 + row;
    if (q < L) {
      *reinterpret_cast<u32x4_t*>(Ob + (int64_t)q * p.ldo + ch * 8) = vh;
      *reinterpret_cast<u32x4_t*>(Ob + (int64_t)q * p.ldo + p.o_lo + ch * 8) = vl;
    }
  }
}

}  // namespace

extern "C" int dm4d_split_f32(void* stream, const float* X1, int64_t row_stride1, int64_t col_stride1, int C1, const float* X2,
                              int64_t row_stride2, int C2, void* Y, int64_t ldy, int64_t M, int Cp, int act_silu, float scale,
                              int pattern) {
  if (!X1 || !Y || M <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && !X2) || Cp < C1 + C2 || pattern < 0 || pattern > 2)
    return dm4d_set_error(DM4D_ERR_ARG, "split_f32: bad arguments");
  if (ldy < (int64_t)(pattern == 0 ? 2 : 3) * Cp) return dm4d_set_error(DM4D_ERR_ARG, "split_f32: ldy too small for the planes");
  SplitParams p{X1, row_stride1, col_stride1, X2, row_stride2, C1, C2, Cp, M, (u16*)Y, ldy, act_silu, scale, pattern};
  hipLaunchKernelGGL(split_kernel, grid1d(M * Cp, 256), dim3(256), 0, (hipStream_t)stream, p);
  return dm4d_check_launch("split_kernel");
}

extern "C" int dm4d_to_f16_f32(void* stream, const float* X1, int64_t row_stride1, int64_t col_stride1, int C1, const float* X2,
                               int64_t row_stride2, int C2, void* Y, int64_t ldy, int64_t M, int Cp, int act_silu, float scale) {
  if (!X1 || !Y || M <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && !X2) || Cp < C1 + C2 || ldy < Cp)
    return dm4d_set_error(DM4D_ERR_ARG, "to_f16_f32: bad arguments");
  SplitParams p{X1, row_stride1, col_stride1, X2, row_stride2, C1, C2, Cp, M, (u16*)Y, ldy, act_silu, scale, 3};
  const bool vec8 = col_stride1 == 1 && ((C1 | C2 | Cp) & 7) == 0 && (row_stride1 & 3) == 0 && (C2 == 0 || (row_stride2 & 3) == 0) && (ldy & 7) == 0 &&
                    (((uintptr_t)X1) & 15) == 0 && (C2 == 0 || (((uintptr_t)X2) & 15) == 0) && (((uintptr_t)Y) & 15) == 0;
  if (vec8) {
    hipLaunchKernelGGL(to_f16_vec8_kernel, grid1d(M * (Cp >> 3), 256), dim3(256), 0, (hipStream_t)stream, p);
    return dm4d_check_launch("to_f16_vec8_kernel");
  }
  hipLaunchKernelGGL(split_kernel, grid1d(M * Cp, 256), dim3(256), 0, (hipStream_t)stream, p);
  return dm4d_check_launch("split_kernel");
}

extern "C" size_t dm4d_groupnorm_f32_ws_bytes(int B, int HW, int groups) {
  return (size_t)B * gn32_nchunk(B, HW) * groups * 2 * sizeof(double);
}

template <bool H16>
static int groupnorm_f32_impl(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW, int groups, float eps,
                              const void* gamma, const void* beta, void* Y, int apply_silu, void* ws) {
  if (!X1 || !gamma || !beta || !Y || !ws || B <= 0 || HW <= 0 || groups <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && !X2))
    return dm4d_set_error(DM4D_ERR_ARG, "groupnorm_f32: bad arguments");
  const int C = C1 + C2;
  if (C % groups != 0 || C > GN_MAXC) return dm4d_set_error(DM4D_ERR_ARG, "groupnorm_f32: channels must divide into groups and be <= 4096");
  GN32Params p{X1, X2, C1, C2, B, HW, groups, gn32_nchunk(B, HW), eps, (const u16*)gamma, (const u16*)beta, (u16*)Y, apply_silu,
               (double*)ws};
  const bool vec4 = C1 % 4 == 0 && C2 % 4 == 0 && ((uintptr_t)X1 & 15) == 0 && (C2 == 0 || ((uintptr_t)X2 & 15) == 0) && ((uintptr_t)Y & 7) == 0;
  if (vec4) {
    const int Q = C / 4, PPB = Q >= GN_THREADS ? 1 : GN_THREADS / Q;
    const size_t sm1 = (size_t)PPB * C * 2 * sizeof(double);
    if (sm1 <= 64 * 1024) {
      hipLaunchKernelGGL(gn32_stats4_kernel, dim3(B * p.nchunk), dim3(GN_THREADS), sm1, (hipStream_t)stream, p);
      int rc = dm4d_check_launch("gn32_stats4_kernel");
      if (rc) return rc;
      hipLaunchKernelGGL((gn32_apply4_kernel<H16>), dim3(B * p.nchunk), dim3(GN_THREADS), 2 * groups * sizeof(float), (hipStream_t)stream, p);
      return dm4d_check_launch("gn32_apply4_kernel");
    }
  }
  const int PPB = C >= GN_THREADS ? 1 : GN_THREADS / C;
  const size_t sm1 = (size_t)PPB * C * 2 * sizeof(double);
  hipLaunchKernelGGL(gn32_stats_kernel, dim3(B * p.nchunk), dim3(GN_THREADS), sm1, (hipStream_t)stream, p);
  int rc = dm4d_check_launch("gn32_stats_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL((gn32_apply_kernel<H16>), dim3(B * p.nchunk), dim3(GN_THREADS), 2 * groups * sizeof(float), (hipStream_t)stream, p);
  return dm4d_check_launch("gn32_apply_kernel");
}

extern "C" int dm4d_groupnorm_nhwc_f32_split(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW,
                                             int groups, float eps, const void* gamma, const void* beta, void* Y, int apply_silu,
                                             void* ws) {
  return groupnorm_f32_impl<false>(stream, X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, Y, apply_silu, ws);
}

extern "C" int dm4d_groupnorm_f32_f16_general(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW, int groups,
                                           float eps, const void* gamma, const void* beta, void* Y, int apply_silu, void* ws) {
  return groupnorm_f32_impl<true>(stream, X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, Y, apply_silu, ws);
}

extern "C" int dm4d_layernorm_f32_split(void* stream, const float* X, int64_t ldx, const void* gamma, const void* beta, void* Y,
                                        int64_t ldy, int M, int C, float eps) {
  if (!X || !gamma || !beta || !Y || M <= 0 || C <= 0 || ldx < C || ldy < 2 * (int64_t)C)
    return dm4d_set_error(DM4D_ERR_ARG, "layernorm_f32: bad arguments");
  hipLaunchKernelGGL(ln32_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, (const u16*)gamma, (const u16*)beta,
                     (u16*)Y, ldy, M, C, eps);
  return dm4d_check_launch("ln32_kernel");
}

extern "C" int dm4d_layernorm_f32_f16_general(void* stream, const float* X, int64_t ldx, const void* gamma, const void* beta, void* Y,
                                      int64_t ldy, int M, int C, float eps) {
  if (!X || !gamma || !beta || !Y || M <= 0 || C <= 0 || ldx < C || ldy < (int64_t)C)
    return dm4d_set_error(DM4D_ERR_ARG, "layernorm_f32_f16: bad arguments");
  hipLaunchKernelGGL(ln32_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, (const u16*)gamma, (const u16*)beta,
                     (u16*)Y, ldy, M, C, eps);
  return dm4d_check_launch("ln32_kernel");
}

extern "C" int dm4d_softmax_rows_f32_f16(void* stream, const float* S, int64_t lds, void* P, int64_t ldp, int M, int N, int Np,
                                         float scale) {
  if (!S || !P || M <= 0 || N <= 0 || Np < N || lds < N || ldp < (int64_t)Np)
    return dm4d_set_error(DM4D_ERR_ARG, "softmax_rows_f32_f16: bad arguments");
  hipLaunchKernelGGL(softmax32_split_kernel<true>, dim3(M), dim3(256), 0, (hipStream_t)stream, S, lds, (u16*)P, ldp, M, N, Np, scale);
  return dm4d_check_launch("softmax32_split_kernel");
}

extern "C" int dm4d_softmax_rows_f32_split(void* stream, const float* S, int64_t lds, void* P, int64_t ldp, int M, int N, int Np,
                                           float scale) {
  if (!S || !P || M <= 0 || N <= 0 || Np < N || lds < N || ldp < 3 * (int64_t)Np)
    return dm4d_set_error(DM4D_ERR_ARG, "softmax_rows_f32_split: bad arguments");
  hipLaunchKernelGGL(softmax32_split_kernel<false>, dim3(M), dim3(256), 0, (hipStream_t)stream, S, lds, (u16*)P, ldp, M, N, Np, scale);
  return dm4d_check_launch("softmax32_split_kernel");
}

extern "C" int dm4d_attention_split_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq, int64_t ldk,
                                         int64_t ldv, int64_t ldo, int64_t q_lo, int64_t k_lo, int64_t v_lo, int64_t o_lo, int batch,
                                         int heads, int Lq, int Lk, float scale) {
  if (!Q || !K || !V || !O || batch <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "attention_split: null pointer or empty shape");
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7) || (q_lo & 7) || (k_lo & 7) || (v_lo & 7) || (o_lo & 7))
    return dm4d_set_error(DM4D_ERR_ARG, "attention_split: strides and plane offsets must be multiples of 8 elements");
  if ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V) | ((uintptr_t)O)) & 15)
    return dm4d_set_error(DM4D_ERR_ARG, "attention_split: Q, K, V and O must be 16-byte aligned");
  AttnSplitParams p{(const u16*)Q, (const u16*)K, (const u16*)V, (u16*)O, ldq, ldk, ldv, ldo, q_lo, k_lo, v_lo, o_lo,
                    Lq, Lk, heads, 0, scale * 1.4426950408889634f};
  constexpr int nw = 8;
  p.nqt = (Lq + nw * 32 - 1) / (nw * 32);
  const long nwg = (long)p.nqt * heads * batch;
  if (nwg > 0x7fffffffL) return dm4d_set_error(DM4D_ERR_ARG, "attention_split: grid too large");
  hipLaunchKernelGGL((attn_split_kernel<nw>), dim3((unsigned)nwg), dim3(nw * 64), 0, (hipStream_t)stream, p);
  return dm4d_check_launch("attn_split_kernel");
}
