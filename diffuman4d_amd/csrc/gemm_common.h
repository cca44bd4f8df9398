// Shared pieces of the MFMA kernels (gemm.hip: Linear layers and convolutions; ff_fused.hip: the fused feed-forward): launch
// parameters, the swapped-operand MFMA wrapper, the bias-as-first-k-step initialisation, the LDS-staged epilogue (bias / SiLU /
// GEGLU / row bias / residual / scale, 16-byte coalesced stores) and the LDS-DMA statement.  Everything lives in an anonymous
// namespace: each translation unit gets its own copy.
#pragma once
#include "common.h"
#include "dm4d.h"
#include "errors.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct GemmParams {
  const u16* A;
  int64_t lda;
  const u16* A2;
  int64_t lda2;
  int K1;
  // conv geometry (CONV only)
  int H, W, Cin, Ho, Wo, stride, pad, upsample;
  const u16* Wt;
  int64_t ldw;
  u16* C;
  int64_t ldc;
  int M, N, K;
  const u16* bias;
  const u16* rowbias;
  int64_t ld_rb;
  int rows_per_rb;
  const u16* res;
  int64_t ld_res;
  unsigned flags;
  float out_scale;
  int tiles_n;
  // split-K of the strip convolution over the 3 kernel rows (small images): partial sums go to ws[split][M][N] fp32
  int splits;
  float* ws;
  // phase-decomposed x2 upsampling convolution (conv_strip2_kernel<.., KT = 2>): rows m of the GEMM are LOW-resolution
  // pixels (b, y, x) of width up_w, output row = 2 m + 2 up_w (m / up_w) from a C pointer moved to the phase's first pixel
  int up_w;
  // precision "fp16" (PAR = 2 kernels): output columns [0, scale_cols) are multiplied by col_scale in fp32 before the one rounding
  // (the to_q third of a fused QKV projection takes scale * log2 e for dm4d_attention_qscaled_kv_f16); 0 = none
  int scale_cols;
  float col_scale;
};

__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};

// Weight row (B-tile row r) -> global row of W.  GEGLU tiles pair hidden rows with their gate rows
// inside each wave's column range so the product is formed in registers.
template <int TN>
__device__ __forceinline__ int weight_row(const GemmParams& p, int n0, int r, bool geglu) {
  int n;
  if (geglu) {
    int wr = r / TN, rr = r % TN;
    n = n0 + wr * (TN / 2) + (rr % (TN / 2));
    if (n > p.N - 1) n = p.N - 1;
    if (rr >= TN / 2) n += p.N;
  } else {
    n = n0 + r;
    if (n > p.N - 1) n = p.N - 1;
  }
  return n;
}

// Every kernel of this file multiplies with the operands SWAPPED: the weight fragment goes in as the MFMA's A operand and
// the activation fragment as its B operand (both fragment formats are "lane & 31 = row / column, lane >> 5 = k half", so the
// same registers serve either role).  The unit then delivers the TRANSPOSED 32x32 block: lane holds output ROW m = lane & 31
// and the 16 COLUMNS n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), i.e. four runs of four consecutive columns.  Every output
// element is the same sum of the same products in the same order as with the operands the other way round (bit-identical
// results), but the epilogue can now stage four columns per LDS instruction (ds_write_b128 of fp32, or ds_write_b64 of four
// packed bf16) instead of one element per ds_write_b32 -- a quarter of the LDS store instructions -- and layers without a
// residual / row bias round to bf16 BEFORE staging (same rounding point: nothing else is applied to them afterwards), which
// halves the staged bytes and leaves a read-back loop of LDS reads and global stores only.
// PAR (template parameter of every kernel): 0 = fast precision, 1 = parity precision (two-term bf16 operands), 2 = precision "fp16":
// the same registers hold fp16 values and go through v_mfma_f32_32x32x16_f16 (same rate, same fragment layout).
template <int PAR = 0>
__device__ __forceinline__ f32x16_t mfma_t(const bf16x8_t& a_rows, const bf16x8_t& w_rows, const f32x16_t& c) {
  if constexpr (PAR == 2) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8_t, w_rows), __builtin_bit_cast(h16x8_t, a_rows), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_rows, a_rows, c, 0, 0, 0);
  }
}

// Stage one 32-row block of a wave (JN 32-column blocks of the transposed accumulators `a`, first one J0; the bias is
// already in them, see gemm_epilogue_impl): GEGLU / SiLU applied, four columns per LDS store -- PLAIN: rounded to bf16 and
// packed (srow = this lane's halfword row + 4 lh), otherwise fp32 (srow = this lane's float row + 4 lh).
template <int NI, int MODE, int J0, int JN, bool PLAIN>
__device__ __forceinline__ void stage_block(const f32x16_t (&a)[NI], void* srow) {
  constexpr bool geglu = MODE == 2;
#pragma unroll
  for (int j = 0; j < JN; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e;
        v[e] = a[J0 + j][r];
        if constexpr (geglu) {
          v[e] = v[e] * gelu_erf_f(a[J0 + j + NI / 2][r]);
        }
        if constexpr (MODE == 1) v[e] = silu_f(v[e]);
      }
      if constexpr (PLAIN) {
        uint2 pk;
        pk.x = pack_bf2(v[0], v[1]);
        pk.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(srow) + j * 32 + 8 * q) = pk;
      } else {
        const f32x4_t v4 = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(srow) + j * 32 + 8 * q) = v4;
      }
    }
  }
}

// The bias as one more k step of the product -- the FIRST one: the accumulators do not start from zero but from an MFMA
// whose activation fragment is e_0 (1.0 in k slot 0 of every row) and whose weight fragment holds the bias in k slot 0 of its
// column's row: fifteen exact zeros and one exact product.  Each lane needs the bias of ONE column per 32-column block
// (lane & 31, the row of the weight tile it would read), whatever the accumulator layout; the 16 v_mov per block of a zero
// initialisation and the 16 v_add_f32 per block of a register-side bias add (VALU time is not hidden behind other waves'
// MFMAs on this chip) become one MFMA.  Without a bias (and for split-K partial sums, whose bias the
// reduce kernel adds) the fragment is zero and the accumulators start at +0.
__device__ __forceinline__ bf16x8_t k0_fragment(u16 bits, int lh) {
  union {
    uint32_t u[4];
    bf16x8_t v;
  } f;
  f.u[0] = lh ? 0u : (uint32_t)bits;
  f.u[1] = f.u[2] = f.u[3] = 0u;
  return f.v;
}

template <int MI, int NI, int TN, int PAR = 0>
__device__ __forceinline__ void acc_init(const GemmParams& p, f32x16_t (&acc)[MI][NI], int n0, int wn, int lane, bool geglu) {
  const int l31 = lane & 31, lh = lane >> 5;
  const bf16x8_t one0 = k0_fragment(PAR == 2 ? 0x3c00 : 0x3f80, lh);  // 1.0 in the operand type (the bias vector has that type too)
  f32x16_t zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  const bool use = p.bias != nullptr && p.splits <= 1;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    u16 bits = 0;
    if (use) bits = p.bias[weight_row<TN>(p, n0, wn * TN + j * 32 + l31, geglu)];  // clamped columns are never stored
    const bf16x8_t bf = k0_fragment(bits, lh);
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[i][j] = mfma_t<PAR>(one0, bf, zero);
  }
}

// The epilogue is specialised at compile time on what it has to apply (MODE: 0 = bias only, 1 = + SiLU, 2 = GEGLU): the
// flags are wave-uniform, but with a runtime `if (do_silu)` inside the 16-element loops hipcc if-converts the branch --
// every output element of every Linear layer and convolution then pays v_exp + v_rcp + a select for a SiLU only the
// two time-embedding GEMMs use, plus one scalar branch per element for GEGLU and per-element staging address arithmetic
// (runtime row stride).  Ablation (profiles/r02_gemm_ablation.log): the epilogue was 39 % of the K = 320 Linear layers'
// time, the stores only 8 %.  With MODE a template parameter the staging stride, the chunk geometry and the trip counts
// are constants: the staging stores take immediate offsets and the read-back loop unrolls.
template <int MI, int NI, int TM, int TN, int MODE, int EPW, int J0, int JN, bool SYNC, bool STRAIGHT, int PAR = 0>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmParams& p, f32x16_t (&acc)[MI][NI], float* smem_f, int m0, int n0,
                                                   int wm, int wn, int wave, int lane) {
  // This instance handles output blocks [J0, J0 + JN) of the wave's NJ 32-column blocks (a "column group"): wide per-wave
  // tiles (TN = 160) are staged in groups that fit the LDS (EPW = columns of staging space per wave).
  const int l31 = lane & 31, lh = lane >> 5;
  constexpr bool geglu = MODE == 2;
  constexpr int NJ = geglu ? NI / 2 : NI;
  static_assert(J0 + JN <= NJ && JN * 32 <= EPW, "column group outside the wave's tile or wider than its staging area");
  constexpr int TNO = JN * 32;   // output columns of this group
  constexpr int SLD = TNO + 4;   // fp32 staging row stride (floats): rows stay 16-byte aligned, b128 stores of 16 lanes tile the banks
  constexpr int SBH = TNO + 4;   // bf16 staging row stride (halfwords) = TNO / 2 + 2 dwords: b64 stores of 32 lanes tile the banks
  constexpr int CPR = TNO / 8;   // 8-column chunks per row
  constexpr int TASKS = 32 * CPR;
  float* stage = smem_f + wave * 32 * (EPW + 4);
  const int ncol0 = n0 + wn * (NJ * 32) + J0 * 32;

  const bool vec_ok = ((p.ldc & 7) == 0) && ((p.N & 7) == 0) && (!p.res || (p.ld_res & 7) == 0) &&
                      (!p.rowbias || (p.ld_rb & 7) == 0);
  const bool f32out = (p.flags & DM4D_EPI_F32OUT) != 0;  // C is float* (fp32 logits of the VAE mid-block attention)
  // the fast read-back loops take row / chunk of a task from shifts: they need a power-of-two number of chunks per row
  constexpr bool CPR_POW2 = (CPR & (CPR - 1)) == 0;
  constexpr int cshift = CPR >= 16 ? 4 : (CPR >= 8 ? 3 : (CPR >= 4 ? 2 : (CPR >= 2 ? 1 : 0)));
  constexpr int cmask = CPR - 1;
  // PAR instantiations (parity-precision launches, host/ops.py precision "parity") also take fp32 row bias / residual
  // (DM4D_EPI_F32SIDE) and split the result into two bf16 terms (DM4D_EPI_SPLITOUT).  They are separate kernels: the code below
  // would otherwise raise the register count of every fast kernel (the 74 KB two-workgroups-per-CU tiles live on 128 registers).
  // PAR == 2 (precision "fp16"): bias / 16-bit side inputs / 16-bit output are fp16, no two-term output; always the general loop.
  constexpr bool H16 = PAR == 2;
  const bool f32side = PAR && (p.flags & DM4D_EPI_F32SIDE) != 0, splitout = PAR == 1 && (p.flags & DM4D_EPI_SPLITOUT) != 0;
  const bool fast_ok = CPR_POW2 && vec_ok && p.ldc < (1 << 24) && (!p.res || p.ld_res < (1 << 24)) &&
                       (!p.rowbias || p.rows_per_rb > 0) && !f32out && !(PAR && (f32side || splitout)) && p.up_w == 0 && !H16;
  // nothing is applied after the staging: round to bf16 first, stage packed halfwords, copy rows out
  const bool plain = fast_ok && !p.res && !p.rowbias && p.out_scale == 1.0f;
  // The staging area is private to a wave and LDS operations of one wave complete in program order, so only one
  // workgroup barrier is needed: the one that retires every wave's main-loop fragment reads before the area is reused
  // (SYNC: the first column group of the tile issues it).
  if constexpr (SYNC) __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m_base = m0 + wm * TM + i * 32;
    if (plain) {
      u16* sb = reinterpret_cast<u16*>(stage);
      u16* srow = sb + l31 * SBH + 4 * lh;  // this lane's staging row; column runs are compile-time offsets away
      stage_block<NI, MODE, J0, JN, true>(acc[i], srow);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // compiler: keep the staging stores ahead of the row reads
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      u16* c_base = p.C + (int64_t)m_base * p.ldc + ncol0;
      if (TASKS % 64 == 0 && m_base + 32 <= p.M && ncol0 + TNO <= p.N) {
        // block entirely inside the matrix (wave-uniform): straight-line code, every LDS read issued before the first store
        U4 o[TASKS / 64 > 0 ? TASKS / 64 : 1];
#pragma unroll
        for (int it = 0; it < TASKS / 64; ++it) {
          const int id = lane + it * 64;
          const int row = id >> cshift, cc = id & cmask;
          const uint2* s = reinterpret_cast<const uint2*>(sb + row * SBH + cc * 8);  // rows are 8-byte aligned
          const uint2 lo = s[0], hi = s[1];
          o[it].x = lo.x; o[it].y = lo.y; o[it].z = hi.x; o[it].w = hi.y;
        }
#pragma unroll
        for (int it = 0; it < TASKS / 64; ++it) {
          const int id = lane + it * 64;
          const int row = id >> cshift, cc = id & cmask;
          stg16(c_base + (uint32_t)(row * (int)p.ldc + cc * 8), o[it]);
        }
      } else {
#pragma unroll
        for (int it = 0; it < (TASKS + 63) / 64; ++it) {
          const int id = lane + it * 64;
          if (TASKS % 64 != 0 && id >= TASKS) continue;
          const int row = id >> cshift, cc = id & cmask;
          if (m_base + row >= p.M || ncol0 + cc * 8 >= p.N) continue;
          const uint2* s = reinterpret_cast<const uint2*>(sb + row * SBH + cc * 8);  // rows are 8-byte aligned
          const uint2 lo = s[0], hi = s[1];
          U4 o;
          o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
          stg16(c_base + (uint32_t)(row * (int)p.ldc + cc * 8), o);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // ... and the row reads ahead of the next block's stores
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      continue;  // next 32-row block of this wave
    }
    float* srow = stage + l31 * SLD + 4 * lh;  // this lane's staging row; column runs are compile-time offsets away
    stage_block<NI, MODE, J0, JN, false>(acc[i], srow);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // compiler: keep the staging stores ahead of the row reads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (fast_ok) {
      // 32-bit offsets from wave-uniform row-block bases, and the rowbias row (m / rows_per_rb) from one division per
      // 32-row block: rows of a block are consecutive, so row r lies in image q0 + (r0 + r >= rows_per_rb).
      const u16* res_base = p.res ? p.res + (int64_t)m_base * p.ld_res + ncol0 : nullptr;
      u16* c_base = p.C + (int64_t)m_base * p.ldc + ncol0;
      int q0 = 0, r0 = 0;
      if (p.rowbias) {
        q0 = m_base / p.rows_per_rb;
        r0 = m_base - q0 * p.rows_per_rb;
      }
      if (STRAIGHT && TASKS % 64 == 0 && m_base + 32 <= p.M && ncol0 + TNO <= p.N &&
          (!p.rowbias || p.rows_per_rb >= 32)) {
        // block entirely inside the matrix (wave-uniform): straight-line code -- the residual / row-bias loads of all of a
        // lane's tasks are issued together and ahead of the staging read-back instead of one load-wait-store chain per task
        auto run = [&](auto has_rb, auto has_res) {
          constexpr bool RB = decltype(has_rb)::value, RS = decltype(has_res)::value;
          constexpr int T = TASKS / 64;      // tasks per lane
          constexpr int CH = T >= 2 ? 2 : 1;  // issued together (more would cost the 74 KB geometries their 128-register budget)
#pragma unroll
          for (int c0 = 0; c0 < T; c0 += CH) {
            U4 trb[RB ? CH : 1], trs[RS ? CH : 1];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
              const int id = lane + (c0 + u) * 64;
              const int row = id >> cshift, cc = id & cmask;
              if constexpr (RB) {
                const int q = q0 + ((r0 + row >= p.rows_per_rb) ? 1 : 0);
                trb[u] = ldg16(p.rowbias + (int64_t)q * p.ld_rb + ncol0 + cc * 8);
              }
              if constexpr (RS) trs[u] = ldg16(res_base + (uint32_t)(row * (int)p.ld_res + cc * 8));
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
              const int id = lane + (c0 + u) * 64;
              const int row = id >> cshift, cc = id & cmask;
              float v[8];
              const float* s = stage + row * SLD + cc * 8;
              f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(s);
              f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(s + 4);
              v[0] = s0[0]; v[1] = s0[1]; v[2] = s0[2]; v[3] = s0[3];
              v[4] = s1[0]; v[5] = s1[1]; v[6] = s1[2]; v[7] = s1[3];
              if constexpr (RB) {
                float t[8];
                unpack8(trb[u], t);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += t[e];
              }
              if constexpr (RS) {
                float t[8];
                unpack8(trs[u], t);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += t[e];
              }
              if (p.out_scale != 1.0f) {  // wave-uniform; x * 1.0f is exact, so skipping it changes nothing
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
              }
              stg16(c_base + (uint32_t)(row * (int)p.ldc + cc * 8), pack8(v));
            }
          }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (p.rowbias && p.res) run(T_{}, T_{});
        else if (p.rowbias) run(T_{}, F_{});
        else if (p.res) run(F_{}, T_{});
        else run(F_{}, F_{});
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // ... and the row reads ahead of the next block's stores
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        continue;  // next 32-row block of this wave
      }
#pragma unroll
      for (int it = 0; it < (TASKS + 63) / 64; ++it) {
        const int id = lane + it * 64;
        if (TASKS % 64 != 0 && id >= TASKS) continue;
        const int row = id >> cshift, cc = id & cmask;
        if (m_base + row >= p.M || ncol0 + cc * 8 >= p.N) continue;
        float v[8];
        const float* s = stage + row * SLD + cc * 8;
        f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(s);
        f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(s + 4);
        v[0] = s0[0]; v[1] = s0[1]; v[2] = s0[2]; v[3] = s0[3];
        v[4] = s1[0]; v[5] = s1[1]; v[6] = s1[2]; v[7] = s1[3];
        if (p.rowbias) {
          int q;
          if (p.rows_per_rb >= 32) {
            q = q0 + ((r0 + row >= p.rows_per_rb) ? 1 : 0);
          } else {
            q = (m_base + row) / p.rows_per_rb;
          }
          float t[8];
          unpack8(ldg16(p.rowbias + (int64_t)q * p.ld_rb + ncol0 + cc * 8), t);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        if (p.res) {
          float t[8];
          unpack8(ldg16(res_base + (uint32_t)(row * (int)p.ld_res + cc * 8)), t);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        if (p.out_scale != 1.0f) {  // wave-uniform; x * 1.0f is exact, so skipping it changes nothing
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
        }
        stg16(c_base + (uint32_t)(row * (int)p.ldc + cc * 8), pack8(v));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // ... and the row reads ahead of the next block's stores
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      continue;  // next 32-row block of this wave
    }
    if constexpr (H16) {
      // precision "fp16": the read-back loop of the fast precision (row / chunk from shifts, 32-bit offsets from wave-uniform
      // row-block bases, the row-bias row from one division per 32-row block) with fp32 or fp16 side inputs and outputs
      if (CPR_POW2 && vec_ok && p.up_w == 0 && p.ldc < (1 << 23) && (!p.res || p.ld_res < (1 << 23)) && (!p.rowbias || p.rows_per_rb > 0)) {
        const int sb = f32side ? 4 : 2, ob = f32out ? 4 : 2;  // bytes per side-input / output element
        const char* res_base = p.res ? reinterpret_cast<const char*>(p.res) + ((int64_t)m_base * p.ld_res + ncol0) * sb : nullptr;
        char* c_base = reinterpret_cast<char*>(p.C) + ((int64_t)m_base * p.ldc + ncol0) * ob;
        int q0 = 0, r0 = 0;
        if (p.rowbias) {
          q0 = m_base / p.rows_per_rb;
          r0 = m_base - q0 * p.rows_per_rb;
        }
        auto side8 = [&](const char* ptr, float* t) {
          if (f32side) {
            const f32x4_t t0 = *reinterpret_cast<const f32x4_t*>(ptr), t1 = *reinterpret_cast<const f32x4_t*>(ptr + 16);
            t[0] = t0[0]; t[1] = t0[1]; t[2] = t0[2]; t[3] = t0[3];
            t[4] = t1[0]; t[5] = t1[1]; t[6] = t1[2]; t[7] = t1[3];
          } else {
            unpack8h(ldg16(ptr), t);
          }
        };
#pragma unroll
        for (int it = 0; it < (TASKS + 63) / 64; ++it) {
          const int id = lane + it * 64;
          if (TASKS % 64 != 0 && id >= TASKS) continue;
          const int row = id >> cshift, cc = id & cmask;
          if (m_base + row >= p.M || ncol0 + cc * 8 >= p.N) continue;
          float v[8];
          const float* s = stage + row * SLD + cc * 8;
          const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(s), s1 = *reinterpret_cast<const f32x4_t*>(s + 4);
          v[0] = s0[0]; v[1] = s0[1]; v[2] = s0[2]; v[3] = s0[3];
          v[4] = s1[0]; v[5] = s1[1]; v[6] = s1[2]; v[7] = s1[3];
          if (p.rowbias) {
            const int q = p.rows_per_rb >= 32 ? q0 + ((r0 + row >= p.rows_per_rb) ? 1 : 0) : (m_base + row) / p.rows_per_rb;
            float t[8];
            side8(reinterpret_cast<const char*>(p.rowbias) + ((int64_t)q * p.ld_rb + ncol0 + cc * 8) * sb, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
          }
          if (p.res) {
            float t[8];
            side8(res_base + (uint32_t)((row * (int)p.ld_res + cc * 8) * sb), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
          }
          const float osc = (ncol0 + cc * 8 < p.scale_cols) ? p.out_scale * p.col_scale : p.out_scale;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= osc;
          char* cp = c_base + (uint32_t)((row * (int)p.ldc + cc * 8) * ob);
          if (f32out) {
            const f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f32x4_t*>(cp) = o0;
            *reinterpret_cast<f32x4_t*>(cp + 16) = o1;
          } else {
            stg16(cp, pack8h(v));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // ... and the row reads ahead of the next block's stores
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        continue;  // next 32-row block of this wave
      }
    }
    for (int id = lane; id < TASKS; id += 64) {
      int row = id / CPR, cc = id % CPR;
      int m = m_base + row;
      int n = ncol0 + cc * 8;
      if (m >= p.M || n >= p.N) continue;
      const int64_t mo = p.up_w ? 2 * (int64_t)m + 2 * (int64_t)p.up_w * (m / p.up_w) : (int64_t)m;  // output row
      float v[8];
      const float* s = stage + row * SLD + cc * 8;
      f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(s);
      f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(s + 4);
      v[0] = s0[0]; v[1] = s0[1]; v[2] = s0[2]; v[3] = s0[3];
      v[4] = s1[0]; v[5] = s1[1]; v[6] = s1[2]; v[7] = s1[3];
      if constexpr (PAR) {
        if (vec_ok) {
          if (p.rowbias) {
            float t[8];
            if (f32side) {
              const float* rb = reinterpret_cast<const float*>(p.rowbias) + (int64_t)(m / p.rows_per_rb) * p.ld_rb + n;
              const f32x4_t t0 = *reinterpret_cast<const f32x4_t*>(rb), t1 = *reinterpret_cast<const f32x4_t*>(rb + 4);
              t[0] = t0[0]; t[1] = t0[1]; t[2] = t0[2]; t[3] = t0[3];
              t[4] = t1[0]; t[5] = t1[1]; t[6] = t1[2]; t[7] = t1[3];
            } else if constexpr (H16) {
              unpack8h(ldg16(p.rowbias + (int64_t)(m / p.rows_per_rb) * p.ld_rb + n), t);
            } else {
              unpack8(ldg16(p.rowbias + (int64_t)(m / p.rows_per_rb) * p.ld_rb + n), t);
            }
  #pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
          }
          if (p.res) {
            float t[8];
            if (f32side) {
              const float* rs = reinterpret_cast<const float*>(p.res) + (int64_t)m * p.ld_res + n;
              const f32x4_t t0 = *reinterpret_cast<const f32x4_t*>(rs), t1 = *reinterpret_cast<const f32x4_t*>(rs + 4);
              t[0] = t0[0]; t[1] = t0[1]; t[2] = t0[2]; t[3] = t0[3];
              t[4] = t1[0]; t[5] = t1[1]; t[6] = t1[2]; t[7] = t1[3];
            } else if constexpr (H16) {
              unpack8h(ldg16(p.res + (int64_t)m * p.ld_res + n), t);
            } else {
              unpack8(ldg16(p.res + (int64_t)m * p.ld_res + n), t);
            }
  #pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
          }
          const float osc = (H16 && n < p.scale_cols) ? p.out_scale * p.col_scale : p.out_scale;  // chunks never straddle scale_cols (a multiple of 8)
  #pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= osc;
          if (f32out) {
            float* cf = reinterpret_cast<float*>(p.C) + mo * p.ldc + n;
            f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f32x4_t*>(cf) = o0;
            *reinterpret_cast<f32x4_t*>(cf + 4) = o1;
          } else if constexpr (H16) {
            stg16(p.C + mo * p.ldc + n, pack8h(v));
          } else if (splitout) {  // x = hi + lo + O(2^-17 x): two bf16 planes, columns [0, N) and [N, 2N) of C
            const U4 hi = pack8(v);
            float h[8];
            unpack8(hi, h);
  #pragma unroll
            for (int e = 0; e < 8; ++e) v[e] -= h[e];
            stg16(p.C + mo * p.ldc + n, hi);
            stg16(p.C + mo * p.ldc + p.N + n, pack8(v));
          } else {
            stg16(p.C + mo * p.ldc + n, pack8(v));
          }
        } else {
          for (int e = 0; e < 8 && n + e < p.N; ++e) {
            float x = v[e];
            if (p.rowbias) {
              const int64_t o = (int64_t)(m / p.rows_per_rb) * p.ld_rb + n + e;
              x += f32side ? reinterpret_cast<const float*>(p.rowbias)[o] : (H16 ? h2f(p.rowbias[o]) : bf2f(p.rowbias[o]));
            }
            if (p.res) {
              const int64_t o = (int64_t)m * p.ld_res + n + e;
              x += f32side ? reinterpret_cast<const float*>(p.res)[o] : (H16 ? h2f(p.res[o]) : bf2f(p.res[o]));
            }
            x *= p.out_scale;
            if (H16 && n + e < p.scale_cols) x *= p.col_scale;
            if (f32out) {
              reinterpret_cast<float*>(p.C)[mo * p.ldc + n + e] = x;
            } else if constexpr (H16) {
              p.C[mo * p.ldc + n + e] = f2h(x);
            } else if (splitout) {
              const u16 hi = f2bf(x);
              p.C[mo * p.ldc + n + e] = hi;
              p.C[mo * p.ldc + p.N + n + e] = f2bf(x - bf2f(hi));
            } else {
              p.C[mo * p.ldc + n + e] = f2bf(x);
            }
          }
        }
      } else {
        if (vec_ok) {
          if (p.rowbias) {
            float t[8];
            unpack8(ldg16(p.rowbias + (int64_t)(m / p.rows_per_rb) * p.ld_rb + n), t);
  #pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
          }
          if (p.res) {
            float t[8];
            unpack8(ldg16(p.res + (int64_t)m * p.ld_res + n), t);
  #pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
          }
  #pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
          if (f32out) {
            float* cf = reinterpret_cast<float*>(p.C) + mo * p.ldc + n;
            f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f32x4_t*>(cf) = o0;
            *reinterpret_cast<f32x4_t*>(cf + 4) = o1;
          } else {
            stg16(p.C + mo * p.ldc + n, pack8(v));
          }
        } else {
          for (int e = 0; e < 8 && n + e < p.N; ++e) {
            float x = v[e];
            if (p.rowbias) x += bf2f(p.rowbias[(int64_t)(m / p.rows_per_rb) * p.ld_rb + n + e]);
            if (p.res) x += bf2f(p.res[(int64_t)m * p.ld_res + n + e]);
            if (f32out) reinterpret_cast<float*>(p.C)[mo * p.ldc + n + e] = x * p.out_scale;
            else p.C[mo * p.ldc + n + e] = f2bf(x * p.out_scale);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // ... and the row reads ahead of the next block's stores
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Staging columns per wave: the whole per-wave tile up to 128 columns; wider tiles (TN = 160) go in column groups.
template <int TN>
struct EpiGeom {
  static constexpr int EPW = TN <= 128 ? TN : 128;
};

template <int MI, int NI, int TM, int TN, int MODE, bool STRAIGHT, int PAR = 0>
__device__ __forceinline__ void gemm_epilogue_mode(const GemmParams& p, f32x16_t (&acc)[MI][NI], float* smem_f, int m0, int n0,
                                                   int wm, int wn, int wave, int lane) {
  constexpr int NJ = MODE == 2 ? NI / 2 : NI;
  constexpr int EPW = EpiGeom<TN>::EPW;
  constexpr int G = EPW / 32;  // blocks per full column group
  if constexpr (NJ <= G) {
    gemm_epilogue_impl<MI, NI, TM, TN, MODE, EPW, 0, NJ, true, STRAIGHT, PAR>(p, acc, smem_f, m0, n0, wm, wn, wave, lane);
  } else {
    static_assert(NJ <= 2 * G, "at most two column groups");
    gemm_epilogue_impl<MI, NI, TM, TN, MODE, EPW, 0, G, true, STRAIGHT, PAR>(p, acc, smem_f, m0, n0, wm, wn, wave, lane);
    gemm_epilogue_impl<MI, NI, TM, TN, MODE, EPW, G, NJ - G, false, STRAIGHT, PAR>(p, acc, smem_f, m0, n0, wm, wn, wave, lane);
  }
}

// STRAIGHT (EpiBudget): whether the kernel can afford the ~20 extra VGPRs of the branch-free residual read-back
template <int NW, int MI, int NI, int SMEM>
struct EpiBudget {
  // not for 16-wave workgroups (128 registers per lane), for the wide 64x160 wave tiles (their 160 accumulators fill the
  // register file), or for the 8-wave tiles of <= 80 KB that rely on a second resident workgroup (128 registers again)
  static constexpr bool STRAIGHT = NW <= 8 && MI * NI <= 8 && !(NW == 8 && SMEM <= 80 * 1024);
};

template <int MI, int NI, int TM, int TN, bool STRAIGHT = false, int PAR = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16_t (&acc)[MI][NI], float* smem_f, int m0, int n0,
                                              int wm, int wn, int wave, int lane) {
  // wave-uniform dispatch; GEGLU pairs hidden block j with gate block j + NI/2 inside the wave, so it needs an even NI
  if constexpr (NI >= 2 && NI % 2 == 0) {
    if (p.flags & DM4D_EPI_GEGLU) {
      gemm_epilogue_mode<MI, NI, TM, TN, 2, STRAIGHT, PAR>(p, acc, smem_f, m0, n0, wm, wn, wave, lane);
      return;
    }
  }
  if (p.flags & DM4D_EPI_SILU) gemm_epilogue_mode<MI, NI, TM, TN, 1, STRAIGHT, PAR>(p, acc, smem_f, m0, n0, wm, wn, wave, lane);
  else gemm_epilogue_mode<MI, NI, TM, TN, 0, STRAIGHT, PAR>(p, acc, smem_f, m0, n0, wm, wn, wave, lane);
}


__device__ __forceinline__ void dma16_sv(const void* uniform_base, uint32_t lane_byte_off, uint32_t lds_dst) {
  // global_load_lds_dwordx4 vaddr(32-bit offset), saddr: lane i's 16 bytes land at lds_dst + 16 i.  Volatile asm: no VGPR
  // destination, completion awaited by hand (s_waitcnt vmcnt(0) before the step's barrier).
  // M0 (the LDS destination) is a reserved register the compiler sets right before each of its own uses, so it is written
  // here without being saved; the statement is atomic as far as the compiler is concerned.
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(lane_byte_off), "s"(uniform_base), "s"(lds_dst)
               : "memory");
}


}  // namespace
