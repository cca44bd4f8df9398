// Fused feed-forward of a transformer block (level 0 of the UNet, C = 320), with the LayerNorm in front of it (norm3) folded in and,
// in the form the UNet calls (ff_proj_fused_kernel), the attention output projection + residual in front of that: a kernel that keeps
// its ACTIVATION ROWS IN REGISTERS for the whole launch.  Shares the MFMA / epilogue helpers of gemm.hip through gemm_common.h.
// (The transformer's proj_out as a post-projection of the same launch was built as well: bit-identical, no gain in a bench step,
// removed -- profiles/r03_ff_proj_fused.log.  The same row-register structure was built for the K = 320 Linear layers -- QKV with norm1 folded in, output
// projection, proj_in / proj_out -- bit-identical and SLOWER than layernorm + gemm on every shape, 217 vs 118 us for QKV: one
// workgroup per CU has nothing to overlap its tile fetch, LayerNorm and epilogue with.  profiles/r03_lin320_rowreg_ab.log; removed.)
#include "gemm_common.h"

namespace {

constexpr int FF_BM = 128, FF_STEP = 32;
#ifdef FF_TILE_LN
constexpr bool FF_REG_LN = false;  // round 3's form of the block tail: h through the LDS tile and the output buffer (A/B builds)
#else
constexpr bool FF_REG_LN = true;
#endif

// ------------------------------------------------------------------------------------------------
// A tile of FF_BM rows x CK channels: HBM -> LDS by DMA as CK / 64 slabs of [FF_BM rows][64 k] (128-byte row pieces, the 16-byte
// chunks of a row XOR-permuted on the source side like every A tile of gemm.hip), optionally LayerNorm'ed IN PLACE, then read
// into registers as the MFMA fragments of the wave's 32 rows (CK / 16 k steps x 4 VGPRs).
// ------------------------------------------------------------------------------------------------
struct LnArgs {
  const u16* gamma;  // nullptr: the rows are used as they are
  const u16* beta;
  float eps;
};

// The attention output projection in front of it all (ff_proj_fused_kernel): h = A0 Wo^T + bo + X is formed by the same workgroup,
// written once to the output buffer (the block's new residual stream) and left in the LDS tile norm3 reads.
struct ProjArgs {
  const u16* A0;  // attention output rows [M, CK]
  int64_t lda0;
  const u16* Wo;  // [CK, CK] row-major (attn1.to_out.0.weight)
  const u16* bo;  // [CK] or nullptr
  const u16* X;   // residual of the projection [M, CK] (the block's input)
  int64_t ldx;
};

template <int CK>
__device__ __forceinline__ void rows_issue(const u16* X, int64_t ldx, int M, int m0, uint32_t lds_tile, int wave, int lane) {
#pragma unroll
  for (int h = 0; h < FF_BM / 64; ++h) {
    const int row = (wave + 8 * h) * 8 + (lane >> 3);
    int m = m0 + row;
    if (m > M - 1) m = M - 1;
    const uint32_t voff = (uint32_t)m * (uint32_t)(ldx * 2) + (uint32_t)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
#pragma unroll
    for (int t = 0; t < CK / 64; ++t) dma16_sv(X + t * 64, voff, lds_tile + t * (FF_BM * 128) + (wave + 8 * h) * 1024);
  }
}

// LayerNorm of the tile in the LDS, in place: wave w normalises rows 16 w .. 16 w + 15 one after the other with lane l on the
// row's 16-byte vector l -- the arrangement and the arithmetic of ln_kernel<1> (ln_row_stats / ln_row_apply), so the result is the
// stand-alone LayerNorm launch bit for bit.  The caller puts a barrier on either side.
// raw (ff_proj_fused_kernel): the un-normalised rows are also written to `raw` (row stride ldraw) on the way -- rows_valid of them.
template <int CK, bool RAW = false>
__device__ __forceinline__ void rows_layernorm(char* tile, const LnArgs& ln, int wave, int lane, u16* raw = nullptr, int64_t ldraw = 0,
                                               int rows_valid = FF_BM) {
  static_assert(CK / 8 <= 64, "one vector per lane");
  const bool on[1] = {lane < CK / 8};
  float g[8], bt[8];
  if (on[0]) {
    unpack8(ldg16(ln.gamma + lane * 8), g);
    unpack8(ldg16(ln.beta + lane * 8), bt);
  }
  const int t = lane >> 3, c = lane & 7;
  for (int i = 0; i < FF_BM / 8; ++i) {
    const int r = wave * (FF_BM / 8) + i;
    char* a = tile + t * (FF_BM * 128) + r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
    float v[1][8];
    if (on[0]) {
      const U4 rawv = *reinterpret_cast<const U4*>(a);
      if constexpr (RAW) {
        if (r < rows_valid) stg16(raw + (int64_t)r * ldraw + lane * 8, rawv);
      }
      unpack8(rawv, v[0]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[0][e] = 0.f;
    }
    float mu, rs;
    ln_row_stats<1>(v, on, CK, ln.eps, mu, rs);
    if (on[0]) {
      float y[8];
      ln_row_apply(v[0], mu, rs, g, bt, y);
      *reinterpret_cast<U4*>(a) = pack8(y);
    }
  }
}

template <int CK>
__device__ __forceinline__ void rows_fragments(const char* tile, int wm, int lane, bf16x8_t (&xf)[CK / 16]) {
  const int l31 = lane & 31, lh = lane >> 5, sw = (l31 >> 1) & 7;
#pragma unroll
  for (int t = 0; t < CK / 64; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      xf[4 * t + ks] = *reinterpret_cast<const bf16x8_t*>(tile + t * (FF_BM * 128) + (wm * 32 + l31) * 128 + (((ks * 2 + lh) ^ sw) * 16));
}

// ------------------------------------------------------------------------------------------------
// Fused feed-forward of a transformer block at C = 320 (level 0 of the UNet; attention.py:129-149: ff(norm3(x)) + x, the
// GEGLU projection 320 -> 2 x 1280 followed by 1280 -> 320):   out = x + W2 (h * gelu(g)) + b2,  [h | g] = W1 y + b1
// in ONE launch: the [M, 1280] hidden tensor (236 MB per layer call at CFG batch 32, written and read back by the two-GEMM
// form) never leaves the chip.
//
// A workgroup = 8 waves owns 128 rows: wave (wm, wn) = (wave >> 1, wave & 1) holds rows 32 wm .. 32 wm + 31.
//   * its y rows live in REGISTERS for the whole launch as MFMA fragments (20 k steps x 4 VGPRs): fetched through the LDS in
//     whole 128-byte row pieces (DMA, source-side swizzle) like every A tile of this file;
//   * the output accumulators O[32 rows x 160 columns] (columns 160 wn ..) stay in registers: 5 blocks x 16 VGPRs;
//   * the hidden axis goes in steps of 32 units.  Step s: (1) S^T[32 x 32] = W1c y^T over K = 320 (20 MFMAs) where W1c holds, for
//     THIS wave's 16 units 32 s + 16 wn .., the 16 hidden rows and then the 16 gate rows -- with the transposed accumulators
//     (mfma_t) a lane then has hidden unit u in register e and its gate in register e + 8; (2) GEGLU in registers, 8 values per
//     lane, rounded to bf16 (the rounding point of the two-GEMM form) and stored to a [128 x 32] tile in LDS, through which
//     the two waves of a row group exchange their halves; (3) O^T += W2c H^T (10 MFMAs: 5 column blocks x 2 k steps).
//   * weights stream L2 -> LDS by DMA, double buffered, from a per-step packed copy (ff_prepare_kernel): W1p[s] = 64 rows x
//     320 (40 KB: rows of wn = 0, then wn = 1), W2p[s] = 320 rows x 32 units (20 KB); every workgroup reads the same 2.4 MB.
// K order of both products = ascending 16-wide k steps into each accumulator with the bias as the first step, exactly as in
// the Linear kernels above, and the same gelu: results are BIT-IDENTICAL to gemm(GEGLU) followed by gemm(residual)
// (tests/opcheck.py ff_fused_*).
// ------------------------------------------------------------------------------------------------
template <int CK, bool PROJ, int PAR = 0>
__device__ __forceinline__ void ff_fused_body(const GemmParams& p, const LnArgs& ln, const ProjArgs& proj, const u16* __restrict__ W1p,
                                              const u16* __restrict__ b1p, const u16* __restrict__ W2p, int nsteps) {
  static_assert(CK % 64 == 0 && (CK / 2) % 32 == 0, "channel count");
  static_assert(PAR == 0 || (PAR == 2 && PROJ), "precision \"fp16\" is built for the whole block tail only");
  constexpr int NSLAB = CK / 64, KS1 = CK / 16, NJ = CK / 2 / 32;  // 64-wide K slabs of y / W1, k steps of product 1, column blocks per wave
  constexpr int W1_BYTES = 64 * CK * 2, W2_BYTES = CK * FF_STEP * 2, WBUF = W1_BYTES + W2_BYTES;
  constexpr int H_LD = 80, H_OFF = 2 * WBUF, H_BYTES = FF_BM * H_LD;  // H rows: 32 units = 64 B + 16 B pad (conflict-free b128 reads); 2 tiles
  constexpr int Y_OFF = WBUF, Y_BYTES = FF_BM * CK * 2;                // prologue only: y tile over buffer 1, H and the tail
  constexpr int SMEM_MAIN = (H_OFF + 2 * H_BYTES) > (Y_OFF + Y_BYTES) ? (H_OFF + 2 * H_BYTES) : (Y_OFF + Y_BYTES);
  constexpr int SMEM_EPI = 8 * 32 * (EpiGeom<CK / 2>::EPW + 4) * 4;
  // projection prologue (PROJ): two Wo slabs of [CK rows][64 k] in front, the attention rows tile behind them -- all of the LDS
  constexpr int WO_SLAB = CK * 128, A0_OFF = 2 * WO_SLAB, SMEM_PROJ = PROJ ? A0_OFF + Y_BYTES : 0;
  constexpr int SMEM_1 = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  constexpr int SMEM_BYTES = SMEM_1 > SMEM_PROJ ? SMEM_1 : SMEM_PROJ;
  static_assert(SMEM_BYTES <= 160 * 1024, "does not fit the LDS");
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * FF_BM;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

  // ---- weight DMA: lane offsets are loop invariant, the step enters through the uniform base ----
  // W1p[s]: NSLAB slabs of [64 rows][64 k]; this wave moves rows 8 wave .. 8 wave + 7 of every slab (one 1-KiB instruction each)
  const int w1_row = wave * 8 + (lane >> 3);
  const uint32_t w1_voff = (uint32_t)w1_row * (CK * 2) + (uint32_t)((lane & 7) ^ ((w1_row >> 1) & 7)) * 16u;
  // W2p[s]: [CK rows][32 units] = CK / 16 instructions of 16 rows; this wave moves instructions wave, wave + 8, ...
  const uint32_t w2_voff = (uint32_t)(lane >> 2) * 64u + (uint32_t)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;
  auto issue_w1 = [&](int s, int b) {
    const u16* w1b = W1p + (int64_t)s * (64 * CK);
    const uint32_t d1 = lds0 + b * WBUF + wave * 1024;
#pragma unroll
    for (int t = 0; t < NSLAB; ++t) dma16_sv(w1b + t * 64, w1_voff, d1 + t * 8192);
  };
  auto issue_w2 = [&](int s, int b) {
    const u16* w2b = W2p + (int64_t)s * (CK * FF_STEP);
    const uint32_t d2 = lds0 + b * WBUF + W1_BYTES;
#pragma unroll
    for (int q = 0; q < (CK / 16 + 7) / 8; ++q) {
      const int inst = wave + 8 * q;
      if (inst < CK / 16) dma16_sv(w2b + inst * 512, w2_voff, d2 + inst * 1024);
    }
  };

  bf16x8_t xf[KS1];
  f32x16_t acc[1][NJ];
  if constexpr (PROJ) {
    // ---- projection prologue: h = A0 Wo^T + bo + X for this workgroup's 128 rows, into the LDS tile at Y_OFF (and, from
    // rows_layernorm, out to p.C: the epilogue's residual).  The products and their order are those of gemm(A0, Wo, bias, residual):
    // bias as the first k step, ascending 16-wide k steps, the residual added to the fp32 sum, one rounding to bf16 -- bit-identical
    // to that launch.  A0 rows -> LDS behind the two Wo buffers -> registers; Wo streams as NSLAB slabs of [CK rows][64 k] = 40 KB
    // straight from its row-major layout (128-byte row pieces, source-side swizzle), double buffered.
    uint32_t wo_voff[CK / 64];
#pragma unroll
    for (int i = 0; i < CK / 64; ++i) {
      const int row = (wave + 8 * i) * 8 + (lane >> 3);
      wo_voff[i] = (uint32_t)row * (CK * 2) + (uint32_t)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
    }
    auto issue_wo = [&](int t, int b) {
#pragma unroll
      for (int i = 0; i < CK / 64; ++i) dma16_sv(proj.Wo + t * 64, wo_voff[i], lds0 + b * WO_SLAB + (wave + 8 * i) * 1024);
    };
    rows_issue<CK>(proj.A0, proj.lda0, p.M, m0, lds0 + A0_OFF, wave, lane);
    issue_wo(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rows_fragments<CK>(smem + A0_OFF, wm, lane, xf);
    f32x16_t hacc[1][NJ];
    {
      GemmParams pb = p;
      pb.bias = proj.bo;
      acc_init<1, NJ, CK / 2, PAR>(pb, hacc, 0, wn, lane, false);
    }
    const int wo_rd = (wn * (CK / 2) + l31) * 128, wo_sw = (l31 >> 1) & 7;
#pragma unroll
    for (int t = 0; t < NSLAB; ++t) {
      if (t + 1 < NSLAB) issue_wo(t + 1, (t + 1) & 1);  // its buffer was last read in slab t - 1, a barrier ago
      const char* wb = smem + (t & 1) * WO_SLAB;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(wb + wo_rd + j * (32 * 128) + (((ks * 2 + lh) ^ wo_sw) * 16));
          hacc[0][j] = mfma_t<PAR>(xf[4 * t + ks], wf, hacc[0][j]);
        }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // residual in the accumulators' own layout (lane: row l31 of its 32, runs of four columns): 8-byte loads, all issued first
    issue_w1(0, 0);  // buffer 0 (Wo slab NSLAB - 1 was its last reader, behind the barrier above); lands while h is formed
    if constexpr (PAR == 2 || FF_REG_LN) {
      // Round 6, fast precision too (FF_REG_LN; -DFF_TILE_LN restores round 3's form for the A/B): the same structure on bf16 tensors --
      // v = A0 Wo^T + bo + X is rounded to bf16 IN THE REGISTERS (the rounding point of gemm(A0, Wo, bias, residual): h as the residual
      // stream holds it), norm3 is taken from those values, and they start the accumulators of the second product, so h is neither stored
      // nor re-read and the serial row-by-row LayerNorm over the LDS tile (36 us of a 348 us launch at M = 92 160, profiles/r06_ffabl2.log)
      // is gone.  Same products and rounding points as the four launches; the ORDER of fp32 additions differs (row statistics, h first in
      // the output sum), so the two forms agree except for isolated one-ulp differences of a bf16 rounding (tests/opcheck.py ff_proj_fused_*).
      // Precision "fp16": the residual stream is fp32 and h = A0 Wo^T + bo + X is never rounded -- and never stored.  It stays in
      // the accumulators it was formed in: norm3 is taken from them (row sums over the lane pair l, l ^ 32 and, through 2 KB of LDS,
      // over the two waves that share a row: two-pass mean / variance in fp32 like ln_row_stats, y rounded once to fp16 with the
      // saturating conversion of ln_kernel<., 2>), the fp16 y rows go to the LDS tile the fragments are read from, and the very same
      // registers then become the accumulators of the second product: O = h + b2 + W2 H.  Against the separate launches of this
      // precision (gemm -> fp32 h, layernorm, gemm GEGLU, gemm + fp32 residual) the 118 MB of h per launch are neither written nor
      // read twice; the sums differ from theirs only in the order of fp32 additions (h enters the output sum first instead of last,
      // the row statistics are added up in a different order): tests/opcheck.py h16_ff_proj_fused_*.
      // (lane-derived addresses of this block are formed HERE, from a copy of the lane id the compiler cannot see through, so that they do
      // not live through the projection loop, whose hacc + xf hold 160 registers; the fast-precision kernel still spills 7 dwords of
      // loop-invariant addresses around this block -- none inside the steady loop)
      int lane_b = lane;
      asm volatile("" : "+v"(lane_b));
      const int l31 = lane_b & 31, lh = lane_b >> 5;
      const int r = wm * 32 + l31;
      {
        int m = m0 + r;
        if (m > p.M - 1) m = p.M - 1;
        if constexpr (PAR == 2) {
          const float* xr = reinterpret_cast<const float*>(proj.X) + (int64_t)m * proj.ldx + wn * (CK / 2) + 4 * lh;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            f32x4_t rx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) rx[q] = *reinterpret_cast<const f32x4_t*>(xr + 32 * j + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) hacc[0][j][4 * q + e] += rx[q][e];
          }
        } else {
          const u16* xr = proj.X + (int64_t)m * proj.ldx + wn * (CK / 2) + 4 * lh;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            uint2 rx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) rx[q] = *reinterpret_cast<const uint2*>(xr + 32 * j + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x4[4] = {bf2f((u16)(rx[q].x & 0xffffu)), bf2f((u16)(rx[q].x >> 16)), bf2f((u16)(rx[q].y & 0xffffu)), bf2f((u16)(rx[q].y >> 16))};
              // h as the residual stream holds it: one rounding to bf16, kept as an fp32 value
              const uint32_t p01 = pack_bf2(hacc[0][j][4 * q + 0] + x4[0], hacc[0][j][4 * q + 1] + x4[1]);
              const uint32_t p23 = pack_bf2(hacc[0][j][4 * q + 2] + x4[2], hacc[0][j][4 * q + 3] + x4[3]);
              hacc[0][j][4 * q + 0] = bf2f((u16)(p01 & 0xffffu));
              hacc[0][j][4 * q + 1] = bf2f((u16)(p01 >> 16));
              hacc[0][j][4 * q + 2] = bf2f((u16)(p23 & 0xffffu));
              hacc[0][j][4 * q + 3] = bf2f((u16)(p23 >> 16));
            }
          }
        }
      }
      float* st = reinterpret_cast<float*>(smem + W1_BYTES);  // [pass][wn][FF_BM]: buffer 0's W2 area, first written in body 0
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += hacc[0][j][e];
      sum += __shfl_xor(sum, 32);
      if (lh == 0) st[wn * FF_BM + r] = sum;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const float mu = (st[r] + st[FF_BM + r]) / (float)CK;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float d = hacc[0][j][e] - mu;
          sq += d * d;
        }
      sq += __shfl_xor(sq, 32);
      if (lh == 0) st[(2 + wn) * FF_BM + r] = sq;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const float rs = rsqrtf((st[2 * FF_BM + r] + st[3 * FF_BM + r]) / (float)CK + ln.eps);
      const u16* gp = ln.gamma + wn * (CK / 2) + 4 * lh;
      const u16* bp = ln.beta + wn * (CK / 2) + 4 * lh;
      const int key = (r >> 1) & 7;
      char* trow = smem + Y_OFF + r * 128 + 8 * lh;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint2 g2 = *reinterpret_cast<const uint2*>(gp + 32 * j + 8 * q), b2 = *reinterpret_cast<const uint2*>(bp + 32 * j + 8 * q);
          auto par = [](u16 v) { return PAR == 2 ? h2f(v) : bf2f(v); };
          const float g[4] = {par((u16)(g2.x & 0xffffu)), par((u16)(g2.x >> 16)), par((u16)(g2.y & 0xffffu)), par((u16)(g2.y >> 16))};
          const float bt[4] = {par((u16)(b2.x & 0xffffu)), par((u16)(b2.x >> 16)), par((u16)(b2.y & 0xffffu)), par((u16)(b2.y >> 16))};
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = (hacc[0][j][4 * q + e] - mu) * rs * g[e] + bt[e];
            if constexpr (PAR == 2) y[e] = sat_h(y[e]);
          }
          uint2 pk;
          if constexpr (PAR == 2) {
            pk.x = pack_h2(y[0], y[1]);
            pk.y = pack_h2(y[2], y[3]);
          } else {
            pk.x = pack_bf2(y[0], y[1]);
            pk.y = pack_bf2(y[2], y[3]);
          }
          const int n = wn * (CK / 2) + 32 * j + 8 * q;  // first of the four columns, before the + 4 lh
          *reinterpret_cast<uint2*>(trow + (n >> 6) * (FF_BM * 128) + ((((n & 63) >> 3) ^ key) << 4)) = pk;
        }
      // b2 as one more k step on top of h: the accumulators of the second product
      const bf16x8_t one_h = k0_fragment(PAR == 2 ? 0x3c00 : 0x3f80, lh);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const u16 bits = p.bias ? p.bias[wn * (CK / 2) + j * 32 + l31] : (u16)0;
        acc[0][j] = mfma_t<PAR>(one_h, k0_fragment(bits, lh), hacc[0][j]);
      }
    } else {
      int m = m0 + wm * 32 + l31;
      if (m > p.M - 1) m = p.M - 1;
      const u16* xr = proj.X + (int64_t)m * proj.ldx + wn * (CK / 2) + 4 * lh;
      uint2 rx[NJ][4];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) rx[j][q] = *reinterpret_cast<const uint2*>(xr + 32 * j + 8 * q);
      const int r = wm * 32 + l31, key = (r >> 1) & 7;
      char* trow = smem + Y_OFF + r * 128 + 8 * lh;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v0 = hacc[0][j][4 * q + 0] + bf2f((u16)(rx[j][q].x & 0xffffu));
          const float v1 = hacc[0][j][4 * q + 1] + bf2f((u16)(rx[j][q].x >> 16));
          const float v2 = hacc[0][j][4 * q + 2] + bf2f((u16)(rx[j][q].y & 0xffffu));
          const float v3 = hacc[0][j][4 * q + 3] + bf2f((u16)(rx[j][q].y >> 16));
          uint2 pk;
          pk.x = pack_bf2(v0, v1);
          pk.y = pack_bf2(v2, v3);
          const int n = wn * (CK / 2) + 32 * j + 8 * q;  // first of the four columns, before the + 4 lh
          *reinterpret_cast<uint2*>(trow + (n >> 6) * (FF_BM * 128) + ((((n & 63) >> 3) ^ key) << 4)) = pk;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // W1(0) and the residual loads landed, the h tile is stored
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    // ---- prologue: y tile -> LDS (5 slabs of [128 rows][64 k]) beside the weights of step 0, then -> registers ----
    rows_issue<CK>(p.A, p.lda, p.M, m0, lds0 + Y_OFF, wave, lane);
    issue_w1(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (!(PROJ && (PAR == 2 || FF_REG_LN)) && ln.gamma) {  // norm3 folded in: the tile holds x, not LayerNorm(x) (bit-identical to the stand-alone launch, rows_layernorm)
    if constexpr (PROJ) {
      rows_layernorm<CK, true>(smem + Y_OFF, ln, wave, lane, p.C + (int64_t)m0 * p.ldc, p.ldc, p.M - m0);
    } else {
      rows_layernorm<CK>(smem + Y_OFF, ln, wave, lane);
    }
    // PROJ: rows_layernorm<RAW> has just stored h to Out with plain global stores, and the epilogue re-reads those rows as the
    // residual (p.res == Out) from OTHER waves of this workgroup.  The stores are drained here (vmcnt(0) covers stores on gfx9-class
    // parts) before the barrier, so that the dependency does not rest on a later interval's wait happening to come first.
    if constexpr (PROJ) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  rows_fragments<CK>(smem + Y_OFF, wm, lane, xf);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave has its y rows: buffer 1 and the H tile may be written
  asm volatile("" ::: "memory");

  if constexpr (!(PROJ && (PAR == 2 || FF_REG_LN))) acc_init<1, NJ, CK / 2>(p, acc, 0, wn, lane, false);  // b2 as the first k step of product 2
  f32x16_t zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  const bf16x8_t one0 = k0_fragment(PAR == 2 ? 0x3c00 : 0x3f80, lh);

  // fragment read addresses (bytes from a buffer's start)
  const int w1_rd = (wn * 32 + l31) * 128, w1_sw = (l31 >> 1) & 7;
  const int w2_rd = W1_BYTES + (wn * (CK / 2) + l31) * 64, w2_sw = (l31 >> 2) & 3;
  const int h_row = H_OFF + (wm * 32 + l31) * H_LD;

  // Schedule: software pipelined by one step, ONE barrier per step.  Body s holds three pieces of work,
  //     A: GEGLU of S(s) -> H(s)                  (vector ALU, 2 LDS stores)
  //     B: O^T += W2c(s-1) H(s-1)^T               (10 MFMAs; H(s-1) was stored before the previous barrier)
  //     C: S(s+1)^T = W1c(s+1) y^T                (21 MFMAs, into the registers A has just read)
  // and receives W1(s+2) -> the buffer of W1(s) and W2(s) -> the buffer of W2(s-2), both last read in body s-1.
  // Two other schedules were built and measured on MI355X (profiles/r03_ff_fused_schedules.log), both bit-identical:
  //  * product 1, GEGLU, barrier, product 2, barrier (two barriers per step): 272-305 us at M = 92 160 against 330-347 us for the
  //    two-GEMM form; its first version also stalled on the bias load behind the DMA issue (see `settle`);
  //  * the two halves of the workgroup half a step apart, so that every SIMD pairs one wave in product 1 with one wave in GEGLU:
  //    305-336 us, SLOWER -- a wave that streams MFMAs starves the vector ALU of the wave it shares the SIMD with (DESIGN.md
  //    section 4, issue probe): vector work hides only behind a wave's OWN MFMAs, which is what A + B in one block provide.
  // Round 6 (profiles/r06_ffp1.log): product 1's weight fragments read 3 / 4 / 6 k steps ahead of their MFMA with hand-counted waits
  // (the compiler keeps one read in flight): bit-identical, +-1 % per launch, null over a bench step -- the partner wave already
  // covers the LDS latency; removed.  Timing-only ablations of the same round (profiles/r06_ffabl.log, r06_ffabl2.log; M = 92 160, 348 us):
  // per step the kernel pays product 1 + product 2 + GEGLU + DMA issue + barrier IN SERIES (1690 + 835 + 780 + 500 + 335 cycles for 1984 cycles
  // of matrix pipe per SIMD); the launch without the steady loop is 121 us, of which the row-by-row LayerNorm over the LDS tile was 36 us
  // and the epilogue 20 us -- hence FF_REG_LN above.
  auto product1 = [&](int s, u16 bias_bits) {
    const char* wb = smem + (s & 1) * WBUF;
    f32x16_t sa = mfma_t<PAR>(one0, k0_fragment(bias_bits, lh), zero);
#pragma unroll
    for (int t = 0; t < NSLAB; ++t)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(wb + t * 8192 + w1_rd + (((ks * 2 + lh) ^ w1_sw) * 16));
        sa = mfma_t<PAR>(xf[4 * t + ks], wf, sa);
      }
    return sa;
  };
  // GEGLU in registers: register e = hidden unit 4 lh + (e & 3) + 8 (e >> 2) of this wave's 16, register e + 8 = its gate.
  // The store of the packed result is a separate step: fragment reads cannot be moved above an LDS store they might alias, so the
  // store comes LAST in a body and the products' reads (and with them their MFMAs) are free to run beside the vector work.
  struct HPack { uint2 lo, hi; };
  auto geglu = [&](const f32x16_t& sa) {
    float hv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) hv[e] = sa[e] * gelu_erf_f(sa[e + 8]);
    HPack h;
    if constexpr (PAR == 2) {
      h.lo.x = pack_h2(hv[0], hv[1]);
      h.lo.y = pack_h2(hv[2], hv[3]);
      h.hi.x = pack_h2(hv[4], hv[5]);
      h.hi.y = pack_h2(hv[6], hv[7]);
    } else {
      h.lo.x = pack_bf2(hv[0], hv[1]);
      h.lo.y = pack_bf2(hv[2], hv[3]);
      h.hi.x = pack_bf2(hv[4], hv[5]);
      h.hi.y = pack_bf2(hv[6], hv[7]);
    }
    return h;
  };
  auto h_store = [&](int s, const HPack& h) {
    char* hp = smem + h_row + (s & 1) * H_BYTES + (16 * wn + 4 * lh) * 2;
    *reinterpret_cast<uint2*>(hp) = h.lo;        // units 16 wn + 4 lh + 0..3
    *reinterpret_cast<uint2*>(hp + 16) = h.hi;   // units 16 wn + 8 + 4 lh + 0..3
  };
  auto product2 = [&](int s) {
    const char* wb = smem + (s & 1) * WBUF;
    const char* hb = smem + h_row + (s & 1) * H_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t hf = *reinterpret_cast<const bf16x8_t*>(hb + (16 * ks + 8 * lh) * 2);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(wb + w2_rd + j * (32 * 64) + (((ks * 2 + lh) ^ w2_sw) * 16));
        acc[0][j] = mfma_t<PAR>(hf, wf, acc[0][j]);
      }
    }
  };
  f32x16_t s_cur;
  auto close_interval = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed, its LDS reads and stores are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // Bias values (one bf16 per lane and step) are ordinary loads the compiler counts; the DMA statements are not.  A value
  // loaded in one body is settled at the top of the body that uses it, BEFORE that body's DMA goes out: the wait the compiler
  // places there finds nothing else in flight (close_interval drained the queue), whereas a wait placed behind the DMA issue
  // stalls on the DMA (the first version of this kernel paid that in every step).
  auto settle = [&](uint32_t v) {
    asm volatile("" : "+v"(v));
    return (u16)v;
  };
  const u16* b1l = b1p + wn * 32 + l31;
  uint32_t b1v = b1l[0];
  // prologue of the pipeline: W1(1) (over the y tile's first 40 KB: every wave holds its rows by now), then S(0)
  {
    const u16 b0 = settle(b1v);
    issue_w1(1, 1);
    b1v = b1l[64];
    s_cur = product1(0, b0);
    close_interval();
  }
  {  // body 0: no product 2 yet
    const u16 bb = settle(b1v);
    const int s2 = 2 < nsteps ? 2 : nsteps - 1;  // past the end: the last slab once more, into a buffer nobody reads again
    issue_w1(s2, 0);                             // (unconditional issue keeps the body one basic block)
    issue_w2(0, 0);
    b1v = b1l[s2 * 64];
    const HPack h = geglu(s_cur);
    s_cur = product1(1, bb);
    h_store(0, h);
    close_interval();
  }
  for (int s = 1; s + 1 < nsteps; ++s) {
    const u16 bb = settle(b1v);
    const int s2 = s + 2 < nsteps ? s + 2 : nsteps - 1;
    issue_w1(s2, s & 1);
    issue_w2(s, s & 1);
    b1v = b1l[s2 * 64];
    __builtin_amdgcn_sched_barrier(0);  // the bias load stays up here (the scheduler otherwise sinks it to the end of the body, in
                                        // front of close_interval's vmcnt(0): one L2 round trip exposed per step)
    const HPack h = geglu(s_cur);
    product2(s - 1);
    s_cur = product1(s + 1, bb);
    h_store(s, h);
    asm volatile("" ::"v"(s_cur), "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]), "v"(acc[0][3]), "v"(acc[0][4]));
    close_interval();
  }
  {  // last body, then the product 2 that is still owed
    const int s = nsteps - 1;
    issue_w2(s, s & 1);
    const HPack h = geglu(s_cur);
    product2(s - 1);
    h_store(s, h);
    close_interval();
    product2(s);
    close_interval();
  }
  // everybody's fragment reads are behind a barrier: the epilogue may stage through the same LDS
  gemm_epilogue<1, NJ, 32, CK / 2, false, PAR>(p, acc, reinterpret_cast<float*>(smem), m0, 0, wm, wn, wave, lane);
}

// (the body is a device function: the host pass instantiates only the stub of a __global__ template, and the register
// constraints of the asm statements above are device-only)
template <int CK>
__global__ __launch_bounds__(512) void ff_fused_kernel(GemmParams p, LnArgs ln, const u16* __restrict__ W1p,
                                                       const u16* __restrict__ b1p, const u16* __restrict__ W2p, int nsteps) {
  ff_fused_body<CK, false>(p, ln, ProjArgs{}, W1p, b1p, W2p, nsteps);
}

// The tail of a transformer block in one launch: attention output projection + residual, norm3, feed-forward + residual
// (attention.py:88-90 and :129-149).  Round 6 (FF_REG_LN): h stays in registers from the projection to the end of the launch.
// (-DFF_TILE_LN, rounds 3-5: p.res == p.C, the projection's result goes out once through rows_layernorm and comes back as the
// epilogue's residual, element by element through the lane that overwrites it.)
template <int CK>
__global__ __launch_bounds__(512) void ff_proj_fused_kernel(GemmParams p, LnArgs ln, ProjArgs proj, const u16* __restrict__ W1p,
                                                            const u16* __restrict__ b1p, const u16* __restrict__ W2p, int nsteps) {
  ff_fused_body<CK, true>(p, ln, proj, W1p, b1p, W2p, nsteps);
}

// The same tail in precision "fp16" (PAR = 2): fp16 attention rows, weights and vectors, the fp32 residual stream X; the result is
// h + ff(norm3(h)) as an fp32 tensor or (the block hands the transformer's proj_out its operand) rounded once to fp16.
template <int CK>
__global__ __launch_bounds__(512) void ff_proj_fused_h16_kernel(GemmParams p, LnArgs ln, ProjArgs proj, const u16* __restrict__ W1p,
                                                                const u16* __restrict__ b1p, const u16* __restrict__ W2p, int nsteps) {
  ff_fused_body<CK, true, 2>(p, ln, proj, W1p, b1p, W2p, nsteps);
}

// Per-step packed copies of the feed-forward weights for ff_fused_kernel (once per layer, at load time):
//   W1p[s][r][k], r = 32 wn + i: unit u = 32 s + 16 wn + (i & 15); row i < 16 = hidden row u of W1 [2 HID, C], i >= 16 = gate row HID + u
//   b1p[s][r] the same rows of b1;   W2p[s][n][j] = W2[n][32 s + j]  (W2 [C, HID])
__global__ __launch_bounds__(256) void ff_prepare_kernel(const u16* W1, const u16* b1, const u16* W2, u16* W1p, u16* b1p, u16* W2p,
                                                         int C, int HID) {
  const int64_t n1 = (int64_t)2 * HID * C, n2 = (int64_t)C * HID;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id < n1) {
    const int k = (int)(id % C);
    const int64_t rr = id / C;
    const int r = (int)(rr % 64), s = (int)(rr / 64);
    const int wn = r >> 5, i = r & 31;
    const int u = 32 * s + 16 * wn + (i & 15);
    const int src = i < 16 ? u : HID + u;
    W1p[id] = W1[(int64_t)src * C + k];
    if (k == 0) b1p[rr] = b1 ? b1[src] : (u16)0;
  } else if (id < n1 + n2) {
    const int64_t e = id - n1;
    const int j = (int)(e % 32);
    const int64_t t = e / 32;
    const int n = (int)(t % C), s = (int)(t / C);
    W2p[e] = W2[(int64_t)n * HID + 32 * s + j];
  }
}

int ff_launch_prepare(hipStream_t st, const u16* W1, const u16* b1, const u16* W2, u16* W1p, u16* b1p, u16* W2p, int C, int hidden) {
  const int64_t total = (int64_t)3 * hidden * C;
  hipLaunchKernelGGL(ff_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W1, b1, W2, W1p, b1p, W2p, C, hidden);
  return dm4d_check_launch("ff_prepare_kernel");
}

int ff_launch_fused(hipStream_t st, const GemmParams& p, const LnArgs& ln, const u16* W1p, const u16* b1p, const u16* W2p, int nsteps) {
  hipLaunchKernelGGL((ff_fused_kernel<320>), dim3((unsigned)((p.M + FF_BM - 1) / FF_BM)), dim3(512), 0, st, p, ln, W1p, b1p, W2p, nsteps);
  return dm4d_check_launch("ff_fused_kernel");
}

int ff_launch_proj_fused(hipStream_t st, const GemmParams& p, const LnArgs& ln, const ProjArgs& proj, const u16* W1p, const u16* b1p,
                         const u16* W2p, int nsteps) {
  hipLaunchKernelGGL((ff_proj_fused_kernel<320>), dim3((unsigned)((p.M + FF_BM - 1) / FF_BM)), dim3(512), 0, st, p, ln, proj, W1p, b1p,
                     W2p, nsteps);
  return dm4d_check_launch("ff_proj_fused_kernel");
}

int ff_launch_proj_fused_h16(hipStream_t st, const GemmParams& p, const LnArgs& ln, const ProjArgs& proj, const u16* W1p, const u16* b1p,
                             const u16* W2p, int nsteps) {
  hipLaunchKernelGGL((ff_proj_fused_h16_kernel<320>), dim3((unsigned)((p.M + FF_BM - 1) / FF_BM)), dim3(512), 0, st, p, ln, proj, W1p,
                     b1p, W2p, nsteps);
  return dm4d_check_launch("ff_proj_fused_h16_kernel");
}

}  // namespace

extern "C" int dm4d_ff_geglu_prepare_bf16(void* stream, const void* W1, const void* b1, const void* W2, void* W1p, void* b1p, void* W2p,
                                          int C, int hidden) {
  if (!W1 || !W2 || !W1p || !b1p || !W2p || C <= 0 || hidden <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "ff_geglu_prepare: null pointer or empty shape");
  if (hidden % 32 != 0) return dm4d_set_error(DM4D_ERR_ARG, "ff_geglu_prepare: hidden size must be a multiple of 32");
  return ff_launch_prepare((hipStream_t)stream, (const u16*)W1, (const u16*)b1, (const u16*)W2, (u16*)W1p, (u16*)b1p, (u16*)W2p, C, hidden);
}

extern "C" int dm4d_ff_geglu_supported(int C, int hidden) { return (C == 320 && hidden >= 64 && hidden % 32 == 0) ? 1 : 0; }

extern "C" int dm4d_ff_geglu_fused_bf16(void* stream, const void* Y, int64_t ldy, const void* ln_gamma, const void* ln_beta, float ln_eps,
                                        const void* W1p, const void* b1p, const void* W2p, const void* b2, const void* residual,
                                        int64_t ld_res, void* Out, int64_t ldo, int M, int C, int hidden) {
  if (!Y || !W1p || !b1p || !W2p || !Out || M <= 0) return dm4d_set_error(DM4D_ERR_ARG, "ff_geglu_fused: null pointer or empty shape");
  if (!dm4d_ff_geglu_supported(C, hidden))
    return dm4d_set_error(DM4D_ERR_ARG, "ff_geglu_fused: built for C = 320 and a hidden size that is a multiple of 32 (use two dm4d_gemm_bf16 calls)");
  if ((ldy & 7) || (ldo & 7) || (residual && (ld_res & 7)) || ((((uintptr_t)Y) | ((uintptr_t)Out) | ((uintptr_t)W1p) | ((uintptr_t)W2p)) & 15))
    return dm4d_set_error(DM4D_ERR_ARG, "ff_geglu_fused: row strides must be multiples of 8 elements, pointers 16-byte aligned");
  if ((uint64_t)M * (uint64_t)ldy * 2u >= (1ull << 32))
    return dm4d_set_error(DM4D_ERR_ARG, "ff_geglu_fused: input of 4 GiB or more (split the rows)");
  if ((ln_gamma == nullptr) != (ln_beta == nullptr)) return dm4d_set_error(DM4D_ERR_ARG, "ff_geglu_fused: LayerNorm needs both gamma and beta");
  GemmParams p{};
  p.A = (const u16*)Y; p.lda = ldy; p.C = (u16*)Out; p.ldc = ldo; p.M = M; p.N = C; p.K = hidden;
  p.bias = (const u16*)b2; p.res = (const u16*)residual; p.ld_res = ld_res; p.flags = 0; p.out_scale = 1.0f; p.splits = 1;
  p.rows_per_rb = 1; p.tiles_n = 1;
  const LnArgs ln{(const u16*)ln_gamma, (const u16*)ln_beta, ln_eps};
  return ff_launch_fused((hipStream_t)stream, p, ln, (const u16*)W1p, (const u16*)b1p, (const u16*)W2p, hidden / FF_STEP);
}


extern "C" int dm4d_attn_out_ff_geglu_fused_bf16(void* stream, const void* A0, int64_t lda0, const void* Wo, const void* bo, const void* X,
                                                 int64_t ldx, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* W1p,
                                                 const void* b1p, const void* W2p, const void* b2, void* Out, int64_t ldo, int M, int C,
                                                 int hidden) {
  if (!A0 || !Wo || !X || !ln_gamma || !ln_beta || !W1p || !b1p || !W2p || !Out || M <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused: null pointer or empty shape");
  if (!dm4d_ff_geglu_supported(C, hidden))
    return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused: built for C = 320 and a hidden size that is a multiple of 32");
  if ((lda0 & 7) || (ldx & 7) || (ldo & 7) ||
      ((((uintptr_t)A0) | ((uintptr_t)Wo) | ((uintptr_t)X) | ((uintptr_t)Out) | ((uintptr_t)W1p) | ((uintptr_t)W2p)) & 15))
    return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused: row strides must be multiples of 8 elements, pointers 16-byte aligned");
  if ((uint64_t)M * (uint64_t)lda0 * 2u >= (1ull << 32))
    return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused: input of 4 GiB or more (split the rows)");
  if (Out == A0 || Out == X) return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused: the output may not alias an input");
  GemmParams p{};
  p.A = (const u16*)A0; p.lda = lda0; p.C = (u16*)Out; p.ldc = ldo; p.M = M; p.N = C; p.K = hidden;
  p.bias = (const u16*)b2; p.flags = 0; p.out_scale = 1.0f; p.splits = 1;
  if (FF_REG_LN) {  // h starts the accumulators: the epilogue has no residual to add
    p.res = nullptr; p.ld_res = 0;
  } else {
    p.res = (const u16*)Out; p.ld_res = ldo;
  }
  p.rows_per_rb = 1; p.tiles_n = 1;
  const LnArgs ln{(const u16*)ln_gamma, (const u16*)ln_beta, ln_eps};
  const ProjArgs proj{(const u16*)A0, lda0, (const u16*)Wo, (const u16*)bo, (const u16*)X, ldx};
  return ff_launch_proj_fused((hipStream_t)stream, p, ln, proj, (const u16*)W1p, (const u16*)b1p, (const u16*)W2p, hidden / FF_STEP);
}

// precision "fp16": see ff_proj_fused_h16_kernel.  W1p / b1p / W2p come from dm4d_ff_geglu_prepare_bf16 (a permutation of 16-bit words,
// whatever they encode).
extern "C" int dm4d_attn_out_ff_geglu_fused_f16(void* stream, const void* A0, int64_t lda0, const void* Wo, const void* bo, const float* X,
                                                int64_t ldx, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* W1p,
                                                const void* b1p, const void* W2p, const void* b2, void* Out, int64_t ldo, int out_f32,
                                                int M, int C, int hidden) {
  if (!A0 || !Wo || !X || !ln_gamma || !ln_beta || !W1p || !b1p || !W2p || !Out || M <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused_f16: null pointer or empty shape");
  if (!dm4d_ff_geglu_supported(C, hidden))
    return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused_f16: built for C = 320 and a hidden size that is a multiple of 32");
  if ((lda0 & 7) || (ldx & 3) || (ldo & 7) || ldx < C || ldo < C || ldo >= (1 << 23) ||
      ((((uintptr_t)A0) | ((uintptr_t)Wo) | ((uintptr_t)X) | ((uintptr_t)Out) | ((uintptr_t)W1p) | ((uintptr_t)W2p) | ((uintptr_t)ln_gamma) |
        ((uintptr_t)ln_beta)) & 15))
    return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused_f16: row strides must be multiples of 8 elements (4 for X), pointers 16-byte aligned");
  if ((uint64_t)M * (uint64_t)lda0 * 2u >= (1ull << 32))
    return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused_f16: input of 4 GiB or more (split the rows)");
  if (Out == A0 || Out == (const void*)X) return dm4d_set_error(DM4D_ERR_ARG, "attn_out_ff_geglu_fused_f16: the output may not alias an input");
  GemmParams p{};
  p.A = (const u16*)A0; p.lda = lda0; p.C = (u16*)Out; p.ldc = ldo; p.M = M; p.N = C; p.K = hidden;
  p.bias = (const u16*)b2; p.res = nullptr; p.ld_res = 0; p.flags = out_f32 ? DM4D_EPI_F32OUT : 0u; p.out_scale = 1.0f; p.splits = 1;
  p.rows_per_rb = 1; p.tiles_n = 1;
  const LnArgs ln{(const u16*)ln_gamma, (const u16*)ln_beta, ln_eps};
  const ProjArgs proj{(const u16*)A0, lda0, (const u16*)Wo, (const u16*)bo, (const u16*)X, ldx};
  return ff_launch_proj_fused_h16((hipStream_t)stream, p, ln, proj, (const u16*)W1p, (const u16*)b1p, (const u16*)W2p, hidden / FF_STEP);
}
