// conv64_kernel: round-6 experiment (tools/conv64/README.md), compiled by tools/conv64/conv64_tu.hip against csrc/gemm_common.h only.
#pragma once

// ------------------------------------------------------------------------------------------------
// conv64_kernel: the strip convolution above on FOUR waves (one per SIMD, the whole 512-register file each) with a hand-placed main
// loop.  Tiling, LDS image, DMA geometry, K order and MFMA operand roles are conv_strip2_kernel's -- the operands computed below are its
// operands, and every output element is the same sum of the same products in the same order -- but the wave tile is 128 x 64 (256 x 128
// workgroup tile) or 64 x 160 (256 x 160), the accumulators live in the accumulator file, and the K walk is ONE asm statement written by
// tools/conv64/cgen.py (conv64_asm.inc: register map, schedule, counted waits, static hazard check; tools/conv64/csim.py runs it on a
// numpy model of the workgroup against a direct convolution -- tests/test_conv64_sim.py).  Stride-1 3 x 3 convolutions without a
// kernel-row split; everything else stays on the kernels above (launch_conv64 returns DM4D_ERR_ARG for what it does not take).
// ------------------------------------------------------------------------------------------------
#ifndef CONV64_INC  // tuning builds (tools/conv64/cab.py) compile other schedules of the same stream
#define CONV64_INC "conv64_asm.inc"
#endif
#include CONV64_INC

template <int BASE>
__device__ __forceinline__ void acc_read16(f32x16_t& x) {
  float t[16];
#define CONV64_RD(i) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(t[i]) : "n"(BASE + i));
  CONV64_RD(0) CONV64_RD(1) CONV64_RD(2) CONV64_RD(3) CONV64_RD(4) CONV64_RD(5) CONV64_RD(6) CONV64_RD(7)
  CONV64_RD(8) CONV64_RD(9) CONV64_RD(10) CONV64_RD(11) CONV64_RD(12) CONV64_RD(13) CONV64_RD(14) CONV64_RD(15)
#undef CONV64_RD
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = t[i];
}
template <int MI, int NI, int Q = 0>
__device__ __forceinline__ void acc_read_all(f32x16_t (&acc)[MI][NI]) {
  if constexpr (Q < MI * NI) {
    acc_read16<Q * 16>(acc[Q / NI][Q % NI]);
    acc_read_all<MI, NI, Q + 1>(acc);
  }
}

#ifdef CONV64_TIMING
__device__ unsigned long long* g_conv64_dbg = nullptr;
__global__ void conv64_set_dbg(unsigned long long* ptr) { g_conv64_dbg = ptr; }
#endif

template <int BN, int WM, int WN, int PAR = 0>
__global__ __launch_bounds__(256, 1) void conv64_kernel(GemmParams p_in) {
  static_assert(WM * WN == 4 && (BN == 128 || BN == 160) && (PAR == 0 || PAR == 2), "the two tiles the stream is generated for");
  constexpr int BM = 256, KT = 3, BK = 64, ROWB = BK * 2, NW = 4;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int NA = (BM + 8) / 8, SR = BM + 16, ZROW = BM + 8, NB = BN / 8;
  constexpr int AW = (NA + NW - 1) / NW, BW = NB / NW;
  constexpr int A_BYTES = SR * ROWB, B_BYTES = BN * ROWB;
  constexpr int SMEM_MAIN = 2 * (A_BYTES + B_BYTES);
  constexpr int SMEM_EPI = NW * 32 * (EpiGeom<TN>::EPW + 4) * 4;
  constexpr int SMEM_BYTES = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  constexpr int BS0 = 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  GemmParams p = p_in;
  constexpr int xs = -1, ys = -1;
  const int tn = lid % p.tiles_n, tm = lid / p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int d_row = lane >> 3, d_pos = lane & 7;
  if (tid < 16) *reinterpret_cast<U4*>(smem + (tid >> 3) * A_BYTES + ZROW * ROWB + (tid & 7) * 16) = U4{0u, 0u, 0u, 0u};
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  {
    // ---- the stream's operands: conv_strip2_kernel's DMA lane offsets and fragment addresses, for all three kernel rows at once ----
    uint32_t avoff[KT][AW], wvoff[BW], arow[KT][KT][MI], aswz[KT][4], brd[4], biasw[NI];
#pragma unroll
    for (int ky = 0; ky < KT; ++ky)
#pragma unroll
      for (int i = 0; i < AW; ++i) {
        const int row = (wave + NW * i) * 8 + d_row;
        int px = m0 + xs + row + (ky + ys) * p.W;
        px = px < 0 ? 0 : (px > p.M - 1 ? p.M - 1 : px);
        avoff[ky][i] = (uint32_t)px * (uint32_t)(p.Cin * 2) + (uint32_t)((d_pos ^ ((row >> 1) & 7)) * 16);
      }
#pragma unroll
    for (int i = 0; i < BW; ++i) {
      const int row = (wave + NW * i) * 8 + d_row;
      const int chunk = d_pos ^ ((row >> 1) & 7);
      int n = n0 + row;
      if (n > p.N - 1) n = p.N - 1;
      wvoff[i] = ((uint32_t)n * (uint32_t)p.ldw + (uint32_t)chunk * 8u) * 2u;
    }
    unsigned edge[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      int m = m0 + wm * TM + i * 32 + l31;
      if (m > p.M - 1) m = p.M - 1;
      const int x = m % p.W, y = (m / p.W) % p.H;
      edge[i] = (x == 0 ? 1u : 0u) | (x == p.W - 1 ? 2u : 0u) | (y == 0 ? 4u : 0u) | (y == p.H - 1 ? 8u : 0u);
    }
#pragma unroll
    for (int kx = 0; kx < KT; ++kx) {
      const int swa = ((l31 + kx) >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) aswz[kx][ks] = ((ks * 2 + lh) ^ swa) * 16;
    }
#pragma unroll
    for (int ky = 0; ky < KT; ++ky)
#pragma unroll
      for (int kx = 0; kx < KT; ++kx)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const unsigned e = edge[i];
          const int ox = kx + xs, oy = ky + ys;
          const bool zero = (ox < 0 && (e & 1u)) || (ox > 0 && (e & 2u)) || (oy < 0 && (e & 4u)) || (oy > 0 && (e & 8u));
          arow[ky][kx][i] = lds0 + (uint32_t)(zero ? ZROW * ROWB : (wm * TM + i * 32 + l31 + kx) * ROWB);
        }
    {
      const int swb = (l31 >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) brd[ks] = lds0 + (uint32_t)(BS0 + (wn * TN + l31) * ROWB + (((ks * 2 + lh) ^ swb) * 16));
    }
    // the bias as the first k step (acc_init): this lane's bias value in k slot 0 of its column's row, 1.0 against it
    const bool use_bias = p.bias != nullptr;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      u16 bits = 0;
      if (use_bias) bits = p.bias[weight_row<TN>(p, n0, wn * TN + j * 32 + l31, false)];
      biasw[j] = lh ? 0u : (uint32_t)bits;
    }
    const uint32_t onew = lh ? 0u : (PAR == 2 ? 0x3c00u : 0x3f80u);
    const u16* abase = p.A;
    const u16* wbase = p.Wt;
    const uint32_t cin2 = (uint32_t)p.Cin * 2u, nci = (uint32_t)(p.Cin / BK);
    const uint32_t adst = lds0 + wave * 1024, bdst = lds0 + BS0 + wave * 1024;
    const uint32_t wave0 = (uint32_t)__builtin_amdgcn_readfirstlane(wave == 0 ? 1 : 0);
#ifdef CONV64_TIMING  // tuning builds only (tools/conv64/cab.py): s_memtime stamps of the stream (start, loop entry, end) per wave
    uint32_t ts[6];
#define CONV64_OUT [ts0] "=s"(ts[0]), [ts1] "=s"(ts[1]), [ts2] "=s"(ts[2]), [ts3] "=s"(ts[3]), [ts4] "=s"(ts[4]), [ts5] "=s"(ts[5])
#else
#define CONV64_OUT
#endif
    if constexpr (BN == 128) {
      if constexpr (PAR == 2) asm volatile(CONV64_ASM_256X128_F16 : CONV64_OUT : CONV64_OPERANDS_256X128 : CONV64_CLOBBERS_256X128);
      else asm volatile(CONV64_ASM_256X128_BF16 : CONV64_OUT : CONV64_OPERANDS_256X128 : CONV64_CLOBBERS_256X128);
    } else {
      if constexpr (PAR == 2) asm volatile(CONV64_ASM_256X160_F16 : CONV64_OUT : CONV64_OPERANDS_256X160 : CONV64_CLOBBERS_256X160);
      else asm volatile(CONV64_ASM_256X160_BF16 : CONV64_OUT : CONV64_OPERANDS_256X160 : CONV64_CLOBBERS_256X160);
    }
#undef CONV64_OUT
#ifdef CONV64_TIMING
    if (g_conv64_dbg && lane == 0) {
      unsigned long long* d = g_conv64_dbg + ((size_t)blockIdx.x * 4 + wave) * 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) d[i] = ((unsigned long long)ts[2 * i + 1] << 32) | ts[2 * i];
    }
#endif
  }
  // the stream ends behind a workgroup barrier with every DMA landed; block (i, j) of the wave tile = a[16 (i NI + j) ..]
  f32x16_t acc[MI][NI];
  acc_read_all<MI, NI>(acc);
  gemm_epilogue<MI, NI, TM, TN, false, PAR>(p, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, wave, lane);
}

// conv64_kernel takes stride-1 3x3 convolutions in whole 64-channel slabs without a kernel-row split (fast and fp16 precisions)
template <int BN, int WM, int WN, int PAR = 0>
int launch_conv64(hipStream_t st, GemmParams& p) {
  // (32-bit byte offsets from a uniform base, like the 8-wave strip kernels)
  if ((uint64_t)p.M * (uint64_t)p.Cin * 2u >= (1ull << 32) || (uint64_t)p.N * (uint64_t)p.ldw * 2u >= (1ull << 32) || p.splits > 1 || p.Cin % 64 != 0)
    return DM4D_ERR_ARG;
  const int tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.splits = 1;
  hipLaunchKernelGGL((conv64_kernel<BN, WM, WN, PAR>), dim3(tiles_m * p.tiles_n), dim3(256), 0, st, p);
  return dm4d_check_launch("conv64_kernel");
}

