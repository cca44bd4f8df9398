// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
//   out[M,N] = epilogue( A[M,K] @ W[N,K]^T )          fp32 accumulate, bf16 in/out
//
// One kernel family serves the Linear layers (dense A, optionally split over two sources = the
// up-block skip concat) and every 3x3 convolution of the UNet/VAE (A gathered on the fly from an
// NHWC tensor: K = 9*Cin ordered (ky, kx, ci), zero padding, stride 1/2, fused nearest-x2 upsample).
//
// Tiling (wave64, v_mfma_f32_32x32x16_bf16): a workgroup of 4 waves owns a BM x BN output tile, each
// wave a (BM/WM) x (BN/WN) sub-tile of 32x32 MFMA fragments.
//
//  * gemm_kernel_glds (main path, K-slab 64): A/B slabs go HBM -> LDS directly with
//    global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass -- the register-staged version was
//    LDS-write-bound), double buffered, the DMA of slab k+1 in flight under the MFMAs of slab k, one
//    barrier per slab.  The LDS image is lane-linear per DMA instruction (8 rows x 128 B), so the
//    bank-conflict swizzle lives in the SOURCE address: LDS position (row, pos) holds the 16-byte
//    chunk pos ^ ((row >> 1) & 7) of that row, and fragment reads apply the same XOR -- conflict-free
//    ds_read_b128 for every 16-lane group.  Convolution padding / stride / upsample only change the
//    per-lane source address (out-of-image taps read a 16-byte zero page).
//  * conv_strip2_kernel (stride-1 convs): same staging, but the A operand is loaded once per kernel row
//    and read at three row offsets for kx = 0,1,2 (see the comment at the kernel); KT = 2 = one phase of the
//    x2-upsampling convolution.
//  * gemm_kernel_pipe (K-slab 32, 3-4 LDS stages, counted vmcnt + raw s_barrier, 4 or 8 waves): the
//    Linear layers, whose short K leaves the two-stage loop latency-bound.
//  * gemm_kernel (fallback, K-slab 32, register staged, padded LDS rows): shapes whose K (or conv
//    Cin) is a multiple of 32 but not of 64.
//
//  * gemm_lin2_kernel (Linear layers: uniform-base DMA, scheduled fragment reads; 256x128 on the 74 KB geometry for
//    short K, 256x256 / 128x128 / 256x320 tiles for the wide, deep and N = 320 layers).
//
// Every MFMA takes its operands swapped (weights as A, activations as B), so a lane holds one output ROW and runs of four
// consecutive columns (mfma_t below).  The bias enters as the first k step of the product (acc_init); GEGLU / SiLU are
// applied in registers; the tile then goes through the LDS -- packed bf16 when nothing else is applied, fp32 when the row
// bias (time-embedding add), residual or scale follow -- so that every global store is a coalesced 16-byte row write.
#include "common.h"
#include "gemm_common.h"

// Build switches that rounds 1-2 used for A/B and ablation runs (GEMM_FAST_EPI, GEMM_EPI_STRAIGHT, STRIP2_SCHED, GEGLU_ERF, GEMM_ABLATE,
// STRIP2_NOWAIT) are resolved to the shipped values: shift / 32-bit-offset index arithmetic in the epilogue read-back loop, the
// branch-free read-back of residual / row-bias blocks, sched_group_barrier placement of the fragment reads in the strip and
// second-form loops, the erf form of GELU in the GEGLU epilogue (DESIGN.md section 4 has the measurements).

namespace {

// ------------------------------------------------------------------------------------------------
// main path: K-slab 64, direct-to-LDS DMA, source-swizzled lane-linear LDS image
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool CONV, int PAR = 0>
__global__ __launch_bounds__(256) void gemm_kernel_glds(GemmParams p) {
  constexpr int BK = 64;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AW = BM / 32, BW = BN / 32;  // 1-KiB DMA instructions per wave per slab (8 rows each)
  static_assert(WM * WN == 4, "4 waves");
  constexpr int SMEM_MAIN = 2 * (BM + BN) * BK * 2;
  constexpr int SMEM_EPI = 4 * 32 * (EpiGeom<TN>::EPW + 4) * 4;
  constexpr int SMEM_BYTES = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  __shared__ __attribute__((aligned(16))) u16 smem[SMEM_BYTES / 2];
  u16* As = smem;                // [2][BM][64]
  u16* Bs = smem + 2 * BM * BK;  // [2][BN][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
  const int m0 = tm * BM;
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const int n0 = tn * (geglu ? BN / 2 : BN);

  // DMA lane mapping: instruction `ia` covers tile rows [8*ia, 8*ia+8); lane -> (row, LDS position)
  const int d_row = lane >> 3, d_pos = lane & 7;
  const u16* a_src[AW];
  const u16* a2_src[AW];
  int iy0[AW], ix0[AW], img_off[AW], tap_off[AW];  // tap_off: element offset of the current tap's pixel, -1 = padding
#pragma unroll
  for (int i = 0; i < AW; ++i) {
    img_off[i] = tap_off[i] = 0;
    const int row = (wave + 4 * i) * 8 + d_row;
    const int chunk = d_pos ^ ((row >> 1) & 7);
    int m = m0 + row;
    if (m > p.M - 1) m = p.M - 1;
    if constexpr (CONV) {
      int ox = m % p.Wo;
      int t = m / p.Wo;
      int oy = t % p.Ho;
      int b = t / p.Ho;
      iy0[i] = oy * p.stride - p.pad;
      ix0[i] = ox * p.stride - p.pad;
      img_off[i] = b * p.H * p.W * p.Cin + chunk * 8;  // element offset of (image b, chunk); tensors are < 2^31 elements
      a_src[i] = p.A;
      a2_src[i] = nullptr;
    } else {
      a_src[i] = p.A + (int64_t)m * p.lda + chunk * 8;
      a2_src[i] = p.A2 ? p.A2 + (int64_t)m * p.lda2 + chunk * 8 : nullptr;
      iy0[i] = ix0[i] = 0;
    }
  }
  const u16* w_src[BW];
#pragma unroll
  for (int i = 0; i < BW; ++i) {
    const int row = (wave + 4 * i) * 8 + d_row;
    const int chunk = d_pos ^ ((row >> 1) & 7);
    w_src[i] = p.Wt + (int64_t)weight_row<TN>(p, n0, row, geglu) * p.ldw + chunk * 8;
  }

  f32x16_t acc[MI][NI];
  acc_init<MI, NI, TN, PAR>(p, acc, n0, wn, lane, geglu);

  const int nk = p.K / BK;
  const int Hin = p.upsample ? 2 * p.H : p.H, Win = p.upsample ? 2 * p.W : p.W;
  int ky = 0, kx = 0, ci0 = 0;  // conv tap of the NEXT slab to be issued

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue_slab = [&](int kt, int buf) {
    if constexpr (CONV) {
      if (ci0 == 0) {  // first slab of a (ky,kx) tap: the only place the gather geometry is evaluated
#pragma unroll
        for (int i = 0; i < AW; ++i) {
          int iy = iy0[i] + ky, ix = ix0[i] + kx;
          bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
          int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
          tap_off[i] = ok ? img_off[i] + (sy * p.W + sx) * p.Cin : -1;
        }
      }
    } else {
      if (p.A2 != nullptr && kt * BK == p.K1) {  // crossed into the second source of the split A (once)
#pragma unroll
        for (int i = 0; i < AW; ++i) a_src[i] = a2_src[i] - p.K1;
      }
    }
#pragma unroll
    for (int i = 0; i < AW; ++i) {
      const u16* src;
      if constexpr (CONV) {
        src = tap_off[i] >= 0 ? p.A + ci0 + tap_off[i] : reinterpret_cast<const u16*>(g_zero16);
      } else {
        src = a_src[i] + kt * BK;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (buf * BM + (wave + 4 * i) * 8) * BK), 16, 0, 0);
    }
    if constexpr (CONV) {
      ci0 += BK;
      if (ci0 >= p.Cin) {
        ci0 = 0;
        if (++kx == 3) {
          kx = 0;
          ++ky;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < BW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + kt * BK), (lptr_t)(Bs + (buf * BN + (wave + 4 * i) * 8) * BK), 16, 0, 0);
  };

  const int sw = (l31 >> 1) & 7;  // fragment rows are (32-aligned base + l31): same swizzle key for all fragments
  issue_slab(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue_slab(kt + 1, buf ^ 1);
    // fragment reads are software-pipelined one k-step ahead of the MFMAs that consume them
    bf16x8_t af[2][MI], bfr[2][NI];
    auto read_frags = [&](int ks, int slot) {
      const int pos = ((ks * 2 + lh) ^ sw) * 8;
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[slot][i] = *reinterpret_cast<const bf16x8_t*>(As + (buf * BM + wm * TM + i * 32 + l31) * BK + pos);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bfr[slot][j] = *reinterpret_cast<const bf16x8_t*>(Bs + (buf * BN + wn * TN + j * 32 + l31) * BK + pos);
    };
    read_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) read_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma_t<PAR>(af[ks & 1][i], bfr[ks & 1][j], acc[i][j]);
    }
    __syncthreads();  // all waves done with `buf`; the DMA into buf^1 has landed (vmcnt(0) before the barrier)
  }
  gemm_epilogue<MI, NI, TM, TN, EpiBudget<WM * WN, MI, NI, SMEM_BYTES>::STRAIGHT, PAR>(p, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// stride-1 3x3 convolution with horizontal tap reuse ("strip" tiling; the kernel is conv_strip2_kernel below)
//
// The M tile is BM consecutive output pixels of the flattened (b, y, x) index.  For one kernel row ky the three
// taps kx = 0,1,2 read the SAME input pixels shifted by one flat position, so the A operand is staged once per
// (ky, 64-channel slab) as a strip of BM+2 pixel rows (flat pixels m0-1 .. m0+BM, moved by (ky-1) image rows) and
// the three taps read it at row offsets 0/1/2: A traffic into LDS drops 3x, total DMA bytes by about a third for a
// 128x128 tile.  A strip row is zero-filled when ITS centre pixel's row y+ky-1 leaves the image; the only other
// out-of-image case -- x-1 at x == 0 for kx = 0, x+1 at x == W-1 for kx = 2, where the flat shift wraps into the
// neighbouring image row -- is removed by pointing those lanes' fragment reads at a zero row in LDS (one address
// select per fragment row and tap; masking the fragment registers instead costs 32 VALU instructions per 16 MFMAs,
// and VALU work is not hidden behind other waves' MFMAs on this chip).  K order is (ky, ci-slab, kx).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Strip convolution kernel.  (Round 1 shipped this tiling as conv_strip_kernel; this form keeps its LDS layout, K order and MFMA
// accumulation order -- results were bit-identical over every shape, tools/dev/strip_ab.py in the round-2 tree -- and replaced it.)
// A main loop that issues nothing but MFMAs, LDS reads, DMA instructions and scalar
// bookkeeping.  The ISA of conv_strip_kernel<256,128,4,2> spends about six VALU instructions per MFMA in its loop:
// 64-bit source pointers rebuilt per DMA instruction (plus a select against a zero page for rows outside the image and a
// PC-relative address of that page), fragment addresses rebuilt per read (swizzle XOR, buffer parity, edge selects); and
// it waits for lgkmcnt(0) in front of every pair of MFMAs.  On this chip a wave's VALU instructions are not hidden
// behind the MFMAs of the wave it shares a SIMD with (issue probe), so all of that is paid in MFMA
// time.  Here:
//   * DMA sources are `uniform 64-bit base (SGPRs) + loop-invariant 32-bit lane offset`: rows outside the image are NOT
//     zero-filled on the way in (their source pixel is clamped to a valid one) ...
//   * ... instead the fragment READS of taps that fall outside the image -- left / right edge as before, now also the
//     rows above / below it -- point at a zero row kept in each A buffer behind the strip (rows BM+8..: never DMA'd);
//   * every fragment read address is a precomputed VGPR (per tap column kx, fragment row block and 16-wide k step; they
//     change only with the kernel row ky, i.e. three times per launch); the buffer parity is added once per step;
//   * the DMA issue is laid out statically per tap column: no per-step branches on the step counters.
// ------------------------------------------------------------------------------------------------
// Scheduling directive for one 16-wide k step: Q MFMAs with the R fragment reads of the NEXT k step spread behind the
// first of them (one basic block; the reads then land in the second fragment register set while the MFMAs run).
template <int Q, int R, int q = 0>
__device__ __forceinline__ void sched_mfma_with_reads() {
  if constexpr (q < Q) {
    constexpr int lo = (q * R) / Q, hi = ((q + 1) * R) / Q;
    if constexpr (hi > lo) __builtin_amdgcn_sched_group_barrier(0x100, hi - lo, 0);  // DS reads (ahead of the MFMA: the
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                               // last read is not the last issue)
    sched_mfma_with_reads<Q, R, q + 1>();
  }
}

// The same for the single-register-set order: MFMA (i, j) is followed by the reload of A fragment i when j is the last
// column block, and of B fragment j when i is the last row block.
template <int MI, int NI, bool RELOAD, int q = 0>
__device__ __forceinline__ void sched_mfma_reload() {
  if constexpr (q < MI * NI) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int i = q / NI, j = q % NI;
    constexpr int n = RELOAD ? ((j == NI - 1 ? 1 : 0) + (i == MI - 1 ? 1 : 0)) : 0;
    if constexpr (n > 0) __builtin_amdgcn_sched_group_barrier(0x100, n, 0);
    sched_mfma_reload<MI, NI, RELOAD, q + 1>();
  }
}

// KT = 2: one PHASE of a 3x3 convolution over a nearest-neighbour x2 upsampled input (Upsample2D).  Output pixel
// (2y + py, 2x + px) reads upsampled rows 2y + py - 1 .. 2y + py + 1, i.e. the low-resolution rows y - 1 + py and y + py
// (one of them twice), and the same along x: each of the four output phases is a 2x2 convolution of the LOW-resolution
// input whose weights are sums of the 3x3 taps that land on the same pixel (dm4d_conv_up2x_prepare_bf16) -- 4 / 9 of the
// multiply-adds of the fused gather the first round shipped.  Same strip machinery: tap (dy, dx) of phase (py, px) reads
// pixel (y + dy - 1 + py, x + dx - 1 + px); the grid carries the phase in its slowest dimension.
template <int BM, int BN, int WM, int WN, int KT = 3, int PAR = 0>
__global__ __launch_bounds__(WM* WN * 64, (BN % 64 != 0 ? 2 : 1)) void conv_strip2_kernel(GemmParams p_in) {
  static_assert(KT == 2 || KT == 3, "3x3 taps, or the 2x2 taps of one upsampling phase");
  constexpr int BK = 64, ROWB = BK * 2;  // bytes per LDS row
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int NA = (BM + 8) / 8;  // 1-KiB DMA instructions per strip (BM + 2 rows needed, 8 rows per instruction)
  constexpr int SR = BM + 16;       // rows per A buffer: the strip, then the zero row at BM + 8
  constexpr int ZROW = BM + 8;
  constexpr int NB = BN / 8;  // 1-KiB DMA instructions per weight slab; a last round may be partial (BN = 160 on 8 waves)
  constexpr int AW = (NA + NW - 1) / NW, BW = (NB + NW - 1) / NW;
  constexpr int A_BYTES = SR * ROWB, B_BYTES = BN * ROWB;
  constexpr int SMEM_MAIN = 2 * (A_BYTES + B_BYTES);
  constexpr int SMEM_EPI = NW * 32 * (EpiGeom<TN>::EPW + 4) * 4;
  constexpr int SMEM_BYTES = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  static_assert(SMEM_BYTES <= 160 * 1024, "tile does not fit the LDS");
  static_assert(2 * A_BYTES + 2 * B_BYTES - 1 + 0 < (1 << 20), "LDS offsets");
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  constexpr int BS0 = 2 * A_BYTES;  // byte offset of the B buffers

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  GemmParams p = p_in;
  int xs = -1, ys = -1;  // tap (ky, kx) reads pixel (y + ky + ys, x + kx + xs)
  if constexpr (KT == 2) {
    const int per = gridDim.x / 4, phase = lid / per;
    lid -= phase * per;
    const int py = phase >> 1, px = phase & 1;
    xs = px - 1;
    ys = py - 1;
    p.Wt = p_in.Wt + (int64_t)phase * p_in.N * p_in.ldw;
    // first output pixel of the phase; C counts 2-byte elements unless the launch stores fp32 (DM4D_EPI_F32OUT: precision "fp16")
    p.C = p_in.C + ((int64_t)py * 2 * p_in.W + px) * p_in.ldc * ((PAR && (p_in.flags & DM4D_EPI_F32OUT)) ? 2 : 1);
  }
  int split = 0, tm, tn;
  if (KT == 3 && p.splits > 1) {
    const int rows = gridDim.x / p.tiles_n;  // tile rows x splits
    const int tms = lid % rows;
    tn = lid / rows;
    split = tms / (rows / p.splits);
    tm = tms % (rows / p.splits);
  } else {
    tn = lid % p.tiles_n;
    tm = lid / p.tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int d_row = lane >> 3, d_pos = lane & 7;
  if (tid < 16) *reinterpret_cast<U4*>(smem + (tid >> 3) * A_BYTES + ZROW * ROWB + (tid & 7) * 16) = U4{0u, 0u, 0u, 0u};

  // ---- DMA lane offsets (bytes): A per kernel row (three times per launch), W fixed ---------------------------------
  uint32_t a_voff[AW];  // the strip row's source pixel for the kernel row being ISSUED, clamped into the tensor
  auto set_a_voff = [&](int ky) {
#pragma unroll
    for (int i = 0; i < AW; ++i) {
      const int row = (wave + NW * i) * 8 + d_row;
      int px = m0 + xs + row + (ky + ys) * p.W;  // strip row `row` holds flat pixel m0 + xs + row, moved by the kernel row
      px = px < 0 ? 0 : (px > p.M - 1 ? p.M - 1 : px);
      a_voff[i] = (uint32_t)px * (uint32_t)(p.Cin * 2) + (uint32_t)((d_pos ^ ((row >> 1) & 7)) * 16);
    }
  };
  uint32_t w_voff[BW];
#pragma unroll
  for (int i = 0; i < BW; ++i) {
    const int row = (wave + NW * i) * 8 + d_row;
    const int chunk = d_pos ^ ((row >> 1) & 7);
    int n = n0 + row;
    if (n > p.N - 1) n = p.N - 1;
    w_voff[i] = ((uint32_t)n * (uint32_t)p.ldw + (uint32_t)chunk * 8u) * 2u;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;

  // ---- fragment read addresses (LDS byte offsets) ------------------------------------------------------------------
  unsigned edge[MI];  // bit 0: x == 0, 1: x == W-1, 2: y == 0, 3: y == H-1 of this lane's output pixel in row block i
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int m = m0 + wm * TM + i * 32 + l31;
    if (m > p.M - 1) m = p.M - 1;
    const int x = m % p.W, y = (m / p.W) % p.H;
    edge[i] = (x == 0 ? 1u : 0u) | (x == p.W - 1 ? 2u : 0u) | (y == 0 ? 4u : 0u) | (y == p.H - 1 ? 8u : 0u);
  }
  // address of a fragment = a_row (row block i of tap column kx: the strip row, or the zero row) + a_swz (k step ks under
  // the strip row's swizzle key) + an immediate for the buffer parity
  int a_row[KT][MI], a_swz[KT][4];
#pragma unroll
  for (int kx = 0; kx < KT; ++kx) {
    const int swa = ((l31 + kx) >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_swz[kx][ks] = ((ks * 2 + lh) ^ swa) * 16;
  }
  auto set_a_addrs = [&](int ky) {
#pragma unroll
    for (int kx = 0; kx < KT; ++kx)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const unsigned e = edge[i];
        const int ox = kx + xs, oy = ky + ys;  // -1 / 0 / +1: the tap's pixel relative to the output pixel
        const bool zero = (ox < 0 && (e & 1u)) || (ox > 0 && (e & 2u)) || (oy < 0 && (e & 4u)) || (oy > 0 && (e & 8u));
        a_row[kx][i] = zero ? ZROW * ROWB : (wm * TM + i * 32 + l31 + kx) * ROWB;
      }
  };
  int b_rd[4];  // [k step]; column block and buffer parity come as immediates
  {
    const int swb = (l31 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_rd[ks] = BS0 + (wn * TN + l31) * ROWB + (((ks * 2 + lh) ^ swb) * 16);
  }

  f32x16_t acc[MI][NI];
  acc_init<MI, NI, TN, PAR>(p, acc, n0, wn, lane, false);

  // ---- the K walk: ky (kernel row) > cs (64-channel slab) > kx (tap column); one DMA'd step ahead -------------------
  const int nci = p.Cin / BK;
  // (the 160-wide tiles never run split: with a split index -- the result of an integer division, i.e. of the vector ALU -- in
  // the weight pointer, the compiler built that pointer in VGPRs for those instantiations, which the DMA's scalar base cannot take)
  constexpr bool SPLITTABLE = KT == 3 && BN % 64 == 0;
  const int ky0 = (SPLITTABLE && p.splits > 1) ? split : 0, ky1 = (SPLITTABLE && p.splits > 1) ? split + 1 : KT;
  const int nstrips = (ky1 - ky0) * nci;
  // DMA issue, statically laid out per tap column (no per-step bookkeeping branches): a weight slab is 8 * NW rows per
  // instruction round, the A strip NA pieces of 8 rows of which the last round is partial
  const uint32_t a_dst0 = lds0 + wave * 1024, b_dst0 = lds0 + BS0 + wave * 1024;
  auto issue_b = [&](const u16* wbase, int par) {  // slab at wbase -> B buffer `par`
    const uint32_t dst = b_dst0 + (par ? B_BYTES : 0);
#pragma unroll
    for (int i = 0; i < BW; ++i) {
      if (NW * (i + 1) <= NB) {
        dma16_sv(wbase, w_voff[i], dst + NW * i * 1024);
      } else if (wave < NB - NW * i) {
        dma16_sv(wbase, w_voff[i], dst + NW * i * 1024);
      }
    }
  };
  auto issue_a = [&](const u16* abase, int par) {  // strip at abase (+ a_voff) -> A buffer `par`
    const uint32_t dst = a_dst0 + (par ? A_BYTES : 0);
#pragma unroll
    for (int i = 0; i < AW; ++i) {
      if (NW * (i + 1) <= NA) {
        dma16_sv(abase, a_voff[i], dst + NW * i * 1024);
      } else if (wave < NA - NW * i) {
        dma16_sv(abase, a_voff[i], dst + NW * i * 1024);
      }
    }
  };
  auto dma_wait_barrier = [&]() {
    // lgkmcnt: the fragment reads of this step must have executed before any wave's next DMA (or the epilogue's staging
    // stores) may overwrite the buffers they read
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // One step (64 K elements of one tap).  Small wave tiles keep two fragment register sets and read k step ks+1 while
  // the MFMAs of ks run; the 64x160 tile (160 accumulators) keeps one set and reloads each fragment right behind the
  // MFMA that used it last in this k step -- at least four MFMAs (128 cycles) before its next use.
  constexpr bool TWO_SETS = MI * NI <= 8;
  auto compute = [&](int abuf, auto kx_c) {  // abuf: strip parity (uniform)
    constexpr int KX = decltype(kx_c)::value;
    const int a_par = abuf ? A_BYTES : 0;
    const int b_par = (KT == 3 ? (abuf ^ (KX & 1)) : (KX & 1)) ? B_BYTES : 0;  // B buffer parity: (KT strip + kx) & 1
    int arow[MI], brow[4];  // the step's own base addresses: MI + 4 adds per step, every read is base + immediate
#pragma unroll
    for (int i = 0; i < MI; ++i) arow[i] = a_row[KX][i] + a_par;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) brow[ks] = b_rd[ks] + b_par;
    auto read_a = [&](int ks, int i) { return *reinterpret_cast<const bf16x8_t*>(smem + arow[i] + a_swz[KX][ks]); };
    auto read_b = [&](int ks, int j) { return *reinterpret_cast<const bf16x8_t*>(smem + brow[ks] + j * 32 * ROWB); };
    if constexpr (TWO_SETS) {
      bf16x8_t af[2][MI], bfr[2][NI];
      auto read_frags = [&](int ks, int slot) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[slot][i] = read_a(ks, i);
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[slot][j] = read_b(ks, j);
      };
      read_frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) read_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = mfma_t<PAR>(af[ks & 1][i], bfr[ks & 1][j], acc[i][j]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);  // fragments of k step 0
      sched_mfma_with_reads<MI * NI, MI + NI>();
      sched_mfma_with_reads<MI * NI, MI + NI>();
      sched_mfma_with_reads<MI * NI, MI + NI>();
      sched_mfma_with_reads<MI * NI, 0>();
    } else {
      bf16x8_t af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = read_a(0, i);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = read_b(0, j);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            acc[i][j] = mfma_t<PAR>(af[i], bfr[j], acc[i][j]);
            if (ks + 1 < 4) {
              if (j == NI - 1) af[i] = read_a(ks + 1, i);
              if (i == MI - 1) bfr[j] = read_b(ks + 1, j);
            }
          }
      __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
      sched_mfma_reload<MI, NI, true>();
      sched_mfma_reload<MI, NI, true>();
      sched_mfma_reload<MI, NI, true>();
      sched_mfma_reload<MI, NI, false>();
    }
  };
  // weights of (ky, cs, kx): Wt + (ky * KT + kx) * Cin + cs * 64; strip of (ky, cs): A + cs * 64 (+ a_voff of ky)
  const int cin = p.Cin;
  int c_ky = ky0, c_cs = 0;
  const u16* wp = p.Wt + ky0 * KT * cin;
  set_a_voff(ky0);
  issue_a(p.A, 0);
  issue_b(wp, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // also publishes the zero rows
  for (int g = 0; g < nstrips; ++g) {
    if (c_cs == 0) set_a_addrs(c_ky);
    const int abuf = g & 1;
    const int b0 = KT == 3 ? abuf : 0;  // weight buffer of the strip's first step
    issue_b(wp + cin, b0 ^ 1);  // (g, kx = 1)
    compute(abuf, std::integral_constant<int, 0>{});
    dma_wait_barrier();
    if constexpr (KT == 3) {
      issue_b(wp + 2 * cin, b0);  // (g, kx = 2)
      compute(abuf, std::integral_constant<int, 1>{});
      dma_wait_barrier();
    }
    if (++c_cs == nci) {
      c_cs = 0;
      ++c_ky;
    }
    if (g + 1 < nstrips) {  // the next strip's A rows and its first weight slab
      if (c_cs == 0) set_a_voff(c_ky);
      wp = p.Wt + (c_ky * KT * cin + c_cs * BK);
      issue_a(p.A + c_cs * BK, abuf ^ 1);
      issue_b(wp, KT == 3 ? (abuf ^ 1) : 0);
    }
    compute(abuf, std::integral_constant<int, KT - 1>{});
    dma_wait_barrier();
  }
  if (KT == 3 && p.splits > 1) {
    // raw fp32 partial sums (splitk_reduce_kernel adds the splits in a fixed order and applies the real epilogue): the
    // same staged, row-coalesced write-out with the workspace slice as an fp32 output matrix and nothing else applied
    p.C = reinterpret_cast<u16*>(p.ws + (int64_t)split * p.M * p.N);
    p.ldc = p.N;
    p.flags = DM4D_EPI_F32OUT;
    p.bias = p.rowbias = p.res = nullptr;
    p.out_scale = 1.0f;
  }
  gemm_epilogue<MI, NI, TM, TN, EpiBudget<WM * WN, MI, NI, SMEM_BYTES>::STRAIGHT, PAR>(p, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// deep pipeline: K-slab 32, NST LDS stages, NST-1 slabs of DMA in flight across barriers
// (counted s_waitcnt vmcnt + raw s_barrier: a __syncthreads() would drain the DMA queue), 4 or 8
// waves.  Ablation of the 2-stage kernel showed the DMA side, not the MFMAs, bounds it (one slab of
// prefetch cannot cover the L2->LDS latency, and a 128x128 tile needs the CU's whole L1 bandwidth at
// MFMA peak); larger tiles halve the bytes per flop, more stages cover the latency.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NST, bool CONV, int BK = 32, int PAR = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel_pipe(GemmParams p) {
  constexpr int NW = WM * WN;
  static_assert(BK == 32 || BK == 64, "K-slab of 32 (64-byte row pieces) or 64 (whole 128-byte lines)");
  constexpr int RPI = 1024 / (BK * 2);  // tile rows per 1-KiB DMA instruction: 16 (BK 32) or 8 (BK 64)
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AW = BM / RPI / NW, BW = BN / RPI / NW;  // 1-KiB DMA instructions (RPI rows of BK bf16) per wave per slab
  static_assert(AW >= 1 && BW >= 1 && NST >= 2 && NST <= 4, "tile/wave/stage combination");
  constexpr int LD = AW + BW;
  constexpr int STAGE = (BM + BN) * BK;  // elements per stage
  constexpr int SMEM_MAIN = NST * STAGE * 2;
  constexpr int SMEM_EPI = NW * 32 * (EpiGeom<TN>::EPW + 4) * 4;
  constexpr int SMEM_BYTES = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  __shared__ __attribute__((aligned(16))) u16 smem[SMEM_BYTES / 2];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
  const int m0 = tm * BM;
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const int n0 = tn * (geglu ? BN / 2 : BN);

  // DMA lane mapping.  BK 32: 16 rows x 4 chunks per instruction, LDS position (row, pos) holds chunk pos ^ ((row>>2)&3).
  // BK 64: 8 rows x 8 chunks (whole 128-byte lines: a 64-byte piece costs a full line of fabric bandwidth when the row
  // comes from HBM / Infinity Cache -- tools/probes/fill_probe.hip: 6 vs 12 B/cycle/CU), position holds chunk pos ^ ((row>>1)&7).
  constexpr int CPRW = BK / 8;  // 16-byte chunks per tile row
  const int d_row = lane / CPRW, d_pos = lane % CPRW;
  // source chunk of tile row `row` for this lane's LDS position (BK 32: instructions start at multiples of 16 rows, so
  // the key depends on d_row only; BK 64: they start at multiples of 8, bit 2 of the key comes from the instruction)
  auto src_chunk = [&](int row) { return BK == 32 ? (d_pos ^ ((row >> 2) & 3)) : (d_pos ^ ((row >> 1) & 7)); };
  const u16* a_src[AW];
  const u16* a2_src[AW];
  int iy0[AW], ix0[AW], img_off[AW], tap_off[AW];  // tap_off: element offset of the current tap's pixel, -1 = padding
#pragma unroll
  for (int i = 0; i < AW; ++i) {
    img_off[i] = tap_off[i] = 0;
    const int d_chunk = src_chunk((wave + NW * i) * RPI + d_row);
    int m = m0 + (wave + NW * i) * RPI + d_row;
    if (m > p.M - 1) m = p.M - 1;
    if constexpr (CONV) {
      int ox = m % p.Wo;
      int t = m / p.Wo;
      int oy = t % p.Ho;
      int b = t / p.Ho;
      iy0[i] = oy * p.stride - p.pad;
      ix0[i] = ox * p.stride - p.pad;
      img_off[i] = b * p.H * p.W * p.Cin + d_chunk * 8;
      a_src[i] = p.A;
      a2_src[i] = nullptr;
    } else {
      a_src[i] = p.A + (int64_t)m * p.lda + d_chunk * 8;
      a2_src[i] = p.A2 ? p.A2 + (int64_t)m * p.lda2 + d_chunk * 8 : nullptr;
      iy0[i] = ix0[i] = 0;
    }
  }
  const u16* w_src[BW];
#pragma unroll
  for (int i = 0; i < BW; ++i)
    w_src[i] = p.Wt + (int64_t)weight_row<TN>(p, n0, (wave + NW * i) * RPI + d_row, geglu) * p.ldw +
               src_chunk((wave + NW * i) * RPI + d_row) * 8;

  f32x16_t acc[MI][NI];
  acc_init<MI, NI, TN, PAR>(p, acc, n0, wn, lane, geglu);

  const int nk = p.K / BK;
  const int Hin = p.upsample ? 2 * p.H : p.H, Win = p.upsample ? 2 * p.W : p.W;
  int ky = 0, kx = 0, ci0 = 0;

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue_slab = [&](int kt, int st) {
    u16* As = smem + st * STAGE;
    u16* Bs = As + BM * BK;
    if constexpr (CONV) {
      if (ci0 == 0) {  // first slab of a (ky,kx) tap: the only place the gather geometry is evaluated
#pragma unroll
        for (int i = 0; i < AW; ++i) {
          int iy = iy0[i] + ky, ix = ix0[i] + kx;
          bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
          int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
          tap_off[i] = ok ? img_off[i] + (sy * p.W + sx) * p.Cin : -1;
        }
      }
    } else {
      if (p.A2 != nullptr && kt * BK == p.K1) {  // crossed into the second source of the split A (once)
#pragma unroll
        for (int i = 0; i < AW; ++i) a_src[i] = a2_src[i] - p.K1;
      }
    }
#pragma unroll
    for (int i = 0; i < AW; ++i) {
      const u16* src;
      if constexpr (CONV) {
        src = tap_off[i] >= 0 ? p.A + ci0 + tap_off[i] : reinterpret_cast<const u16*>(g_zero16);
      } else {
        src = a_src[i] + kt * BK;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (wave + NW * i) * RPI * BK), 16, 0, 0);
    }
    if constexpr (CONV) {
      ci0 += BK;
      if (ci0 >= p.Cin) {
        ci0 = 0;
        if (++kx == 3) {
          kx = 0;
          ++ky;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < BW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + kt * BK), (lptr_t)(Bs + (wave + NW * i) * RPI * BK), 16, 0, 0);
  };

  const int sw = BK == 32 ? ((l31 >> 2) & 3) : ((l31 >> 1) & 7);
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue_slab(s, s);
  int st = 0, st_issue = NST - 1;  // stage holding slab kt / stage receiving slab kt + NST - 1
  for (int kt = 0; kt < nk; ++kt) {
    // slab kt has landed once at most min(NST-2, nk-1-kt) younger slabs of THIS wave are outstanding
    const int younger = nk - 1 - kt;
    if (NST >= 4 && younger >= 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LD) : "memory");
    } else if (NST >= 3 && younger >= 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LD) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // every wave's part of slab kt is in LDS; everyone is done reading slab kt-1
    asm volatile("" ::: "memory");
    if (kt + NST - 1 < nk) issue_slab(kt + NST - 1, st_issue);  // overwrites the stage slab kt-1 lived in
    const u16* As = smem + st * STAGE;
    const u16* Bs = As + BM * BK;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int pos = ((ks * 2 + lh) ^ sw) * 8;
      bf16x8_t af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * TM + i * 32 + l31) * BK + pos);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * TN + j * 32 + l31) * BK + pos);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_t<PAR>(af[i], bfr[j], acc[i][j]);
    }
    st = (st + 1 == NST) ? 0 : st + 1;
    st_issue = (st_issue + 1 == NST) ? 0 : st_issue + 1;
  }
  gemm_epilogue<MI, NI, TM, TN, EpiBudget<WM * WN, MI, NI, SMEM_BYTES>::STRAIGHT, PAR>(p, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// Linear layers, second form (the counterpart of conv_strip2_kernel for dense A): K-slab 64 (whole 128-byte DMA rows),
// NST-stage ring with NST-1 slabs in flight under a counted vmcnt, DMA sources as `uniform base + loop-invariant 32-bit
// lane offset`, fragment reads at `per-step base + immediate`, reads of k step s+1 (or single-set reloads) placed among
// the MFMAs of step s by scheduling directives.  Same K order as every other kernel of this file (ascending 16-wide k
// steps into each accumulator), so results are bit-identical to gemm_kernel_pipe / gemm_kernel_glds.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NST, int BK = 64, int PAR = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_lin2_kernel(GemmParams p) {
  static_assert(BK == 32 || BK == 64, "K-slab of 32 (64-byte rows, 16 per DMA instruction) or 64 (128-byte rows, 8 per instruction)");
  constexpr int ROWB = BK * 2, RPI = 1024 / ROWB, CPR = BK / 8, KS = BK / 16;  // row bytes, rows per DMA instruction, chunks per row, k steps
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AW = BM / RPI / NW, BW = BN / RPI / NW;  // 1-KiB DMA instructions per wave per slab
  static_assert(AW >= 1 && BW >= 1 && NST >= 2 && NST <= 3, "tile/wave/stage combination");
  constexpr int LD = AW + BW;
  constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  constexpr int SMEM_MAIN = NST * STAGE_BYTES;
  constexpr int SMEM_EPI = NW * 32 * (EpiGeom<TN>::EPW + 4) * 4;
  constexpr int SMEM_BYTES = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  static_assert(SMEM_BYTES <= 160 * 1024, "tile does not fit the LDS");
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
  const int m0 = tm * BM;
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const int n0 = tn * (geglu ? BN / 2 : BN);
  const int d_row = lane / CPR, d_pos = lane % CPR;
  // LDS position (row, pos) holds chunk pos ^ key(row): key = (row >> 1) & 7 over 8 chunks, (row >> 2) & 3 over 4
  auto key = [](int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  uint32_t a_voff[AW], a2_voff[AW], w_voff[BW];
#pragma unroll
  for (int i = 0; i < AW; ++i) {
    const int row = (wave + NW * i) * RPI + d_row;
    const uint32_t chunk = (uint32_t)(d_pos ^ key(row)) * 16u;
    int m = m0 + row;
    if (m > p.M - 1) m = p.M - 1;
    a_voff[i] = (uint32_t)m * (uint32_t)(p.lda * 2) + chunk;
    a2_voff[i] = (uint32_t)m * (uint32_t)(p.lda2 * 2) + chunk;
  }
#pragma unroll
  for (int i = 0; i < BW; ++i) {
    const int row = (wave + NW * i) * RPI + d_row;
    w_voff[i] = (uint32_t)weight_row<TN>(p, n0, row, geglu) * (uint32_t)(p.ldw * 2) + (uint32_t)(d_pos ^ key(row)) * 16u;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const uint32_t a_dst0 = lds0 + wave * 1024, b_dst0 = lds0 + BM * ROWB + wave * 1024;
  const int k1 = p.A2 ? p.K1 : p.K;  // slabs below k1 come from A, the others from A2 (split A: K1 % 64 == 0)
  auto issue_slab = [&](int kt, int st) {
    const int k = kt * BK;
    const uint32_t so = st * STAGE_BYTES;
    if (k < k1) {
      const u16* ab = p.A + k;
#pragma unroll
      for (int i = 0; i < AW; ++i) dma16_sv(ab, a_voff[i], a_dst0 + so + NW * i * 1024);
    } else {
      const u16* ab = p.A2 + (k - k1);
#pragma unroll
      for (int i = 0; i < AW; ++i) dma16_sv(ab, a2_voff[i], a_dst0 + so + NW * i * 1024);
    }
    const u16* wb = p.Wt + k;
#pragma unroll
    for (int i = 0; i < BW; ++i) dma16_sv(wb, w_voff[i], b_dst0 + so + NW * i * 1024);
  };

  // fragment read addresses: row base + k-step swizzle (A and B rows share the key: both are l31 + a multiple of 32)
  int a_rd[KS], b_rd[KS];
  {
    const int sw = key(l31);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int z = ((ks * 2 + lh) ^ sw) * 16;
      a_rd[ks] = (wm * TM + l31) * ROWB + z;
      b_rd[ks] = BM * ROWB + (wn * TN + l31) * ROWB + z;
    }
  }

  f32x16_t acc[MI][NI];
  acc_init<MI, NI, TN, PAR>(p, acc, n0, wn, lane, geglu);

  constexpr bool TWO_SETS = MI * NI <= 8;
  auto compute = [&](int st) {
    const int so = st * STAGE_BYTES;
    int ar[KS], br[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      ar[ks] = a_rd[ks] + so;
      br[ks] = b_rd[ks] + so;
    }
    auto read_a = [&](int ks, int i) { return *reinterpret_cast<const bf16x8_t*>(smem + ar[ks] + i * 32 * ROWB); };
    auto read_b = [&](int ks, int j) { return *reinterpret_cast<const bf16x8_t*>(smem + br[ks] + j * 32 * ROWB); };
    if constexpr (TWO_SETS) {
      bf16x8_t af[2][MI], bfr[2][NI];
      auto read_frags = [&](int ks, int slot) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[slot][i] = read_a(ks, i);
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[slot][j] = read_b(ks, j);
      };
      read_frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) read_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = mfma_t<PAR>(af[ks & 1][i], bfr[ks & 1][j], acc[i][j]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
      sched_mfma_with_reads<MI * NI, MI + NI>();
      if constexpr (KS == 4) {
        sched_mfma_with_reads<MI * NI, MI + NI>();
        sched_mfma_with_reads<MI * NI, MI + NI>();
      }
      sched_mfma_with_reads<MI * NI, 0>();
    } else {
      bf16x8_t af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = read_a(0, i);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = read_b(0, j);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            acc[i][j] = mfma_t<PAR>(af[i], bfr[j], acc[i][j]);
            if (ks + 1 < KS) {
              if (j == NI - 1) af[i] = read_a(ks + 1, i);
              if (i == MI - 1) bfr[j] = read_b(ks + 1, j);
            }
          }
      __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
      sched_mfma_reload<MI, NI, true>();
      if constexpr (KS == 4) {
        sched_mfma_reload<MI, NI, true>();
        sched_mfma_reload<MI, NI, true>();
      }
      sched_mfma_reload<MI, NI, false>();
    }
  };

  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue_slab(s, s);
  int st = 0, st_issue = NST - 1;  // stage holding slab kt / stage receiving slab kt + NST - 1
  for (int kt = 0; kt < nk; ++kt) {
    // slab kt has landed once at most min(NST-2, nk-1-kt) younger slabs of THIS wave are outstanding; lgkmcnt(0): this
    // wave's fragment reads of slab kt-1 are done before anyone overwrites its stage
    if (NST >= 3 && nk - 1 - kt >= 1) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LD) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NST - 1 < nk) issue_slab(kt + NST - 1, st_issue);  // overwrites the stage slab kt-1 lived in
    compute(st);
    st = (st + 1 == NST) ? 0 : st + 1;
    st_issue = (st_issue + 1 == NST) ? 0 : st_issue + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // all fragment reads done: the epilogue stages through the same LDS
  asm volatile("" ::: "memory");
  gemm_epilogue<MI, NI, TM, TN, EpiBudget<WM * WN, MI, NI, SMEM_BYTES>::STRAIGHT, PAR>(p, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// fallback: K-slab 32, register staged, padded LDS rows (80 B stride => conflict-free fragment reads)
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool CONV, int PAR = 0>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int LDK = 40;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AI = BM / 64, BI = BN / 64;
  static_assert(WM * WN == 4, "4 waves");
  constexpr int SMEM_MAIN = 2 * (BM + BN) * LDK * 2;
  constexpr int SMEM_EPI = 4 * 32 * (EpiGeom<TN>::EPW + 4) * 4;
  constexpr int SMEM_BYTES = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  __shared__ __attribute__((aligned(16))) u16 smem[SMEM_BYTES / 2];
  u16* As = smem;
  u16* Bs = smem + 2 * BM * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
  const int m0 = tm * BM;
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const int n0 = tn * (geglu ? BN / 2 : BN);
  const int a_c = tid & 3, a_r = tid >> 2;

  const u16* a_base[AI];
  const u16* a2_base[AI];
  int iy0[AI], ix0[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + a_r + 64 * i;
    if (m > p.M - 1) m = p.M - 1;
    if constexpr (CONV) {
      int ox = m % p.Wo;
      int t = m / p.Wo;
      int oy = t % p.Ho;
      int b = t / p.Ho;
      iy0[i] = oy * p.stride - p.pad;
      ix0[i] = ox * p.stride - p.pad;
      a_base[i] = p.A + (int64_t)b * p.H * p.W * p.Cin;
      a2_base[i] = nullptr;
    } else {
      a_base[i] = p.A + (int64_t)m * p.lda;
      a2_base[i] = p.A2 ? p.A2 + (int64_t)m * p.lda2 : nullptr;
      iy0[i] = ix0[i] = 0;
    }
  }
  const u16* w_base[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) w_base[i] = p.Wt + (int64_t)weight_row<TN>(p, n0, a_r + 64 * i, geglu) * p.ldw;

  f32x16_t acc[MI][NI];
  acc_init<MI, NI, TN, PAR>(p, acc, n0, wn, lane, geglu);

  const int nk = p.K / 32;
  const int Hin = p.upsample ? 2 * p.H : p.H, Win = p.upsample ? 2 * p.W : p.W;
  int ky = 0, kx = 0, ci0 = 0;

  U4 ra[AI], rb[BI];
  auto load_slab = [&](int kt) {
    if constexpr (CONV) {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        int iy = iy0[i] + ky, ix = ix0[i] + kx;
        bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
        U4 z = {0u, 0u, 0u, 0u};
        ra[i] = ok ? ldg16(a_base[i] + ((int64_t)sy * p.W + sx) * p.Cin + ci0 + a_c * 8) : z;
      }
      ci0 += 32;
      if (ci0 >= p.Cin) {
        ci0 = 0;
        if (++kx == 3) {
          kx = 0;
          ++ky;
        }
      }
    } else {
      const int kcol = kt * 32;
      const bool second = (p.A2 != nullptr) && (kcol >= p.K1);
#pragma unroll
      for (int i = 0; i < AI; ++i)
        ra[i] = second ? ldg16(a2_base[i] + (kcol - p.K1) + a_c * 8) : ldg16(a_base[i] + kcol + a_c * 8);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = ldg16(w_base[i] + kt * 32 + a_c * 8);
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AI; ++i) *reinterpret_cast<U4*>(As + (buf * BM + a_r + 64 * i) * LDK + a_c * 8) = ra[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<U4*>(Bs + (buf * BN + a_r + 64 * i) * LDK + a_c * 8) = rb[i];
  };

  load_slab(0);
  store_slab(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_slab(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const bf16x8_t*>(As + (buf * BM + wm * TM + i * 32 + l31) * LDK + ks * 16 + lh * 8);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bs + (buf * BN + wn * TN + j * 32 + l31) * LDK + ks * 16 + lh * 8);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma_t<PAR>(af[i], bfr[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_slab(buf ^ 1);
    __syncthreads();
  }
  gemm_epilogue<MI, NI, TM, TN, EpiBudget<WM * WN, MI, NI, SMEM_BYTES>::STRAIGHT, PAR>(p, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, wave, lane);
}

template <int BM, int BN, int WM, int WN, bool CONV, bool GLDS, int PAR = 0>
int launch_cfg(hipStream_t st, GemmParams& p) {
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const int bn_out = geglu ? BN / 2 : BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + bn_out - 1) / bn_out;
  if constexpr (GLDS) {
    hipLaunchKernelGGL((gemm_kernel_glds<BM, BN, WM, WN, CONV, PAR>), dim3(tiles_m * p.tiles_n), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, CONV, PAR>), dim3(tiles_m * p.tiles_n), dim3(256), 0, st, p);
  }
  return dm4d_check_launch("gemm_kernel");
}

template <int BM, int BN, int WM, int WN, int NST, bool CONV, int BK = 32, int PAR = 0>
int launch_pipe(hipStream_t st, GemmParams& p) {
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const int bn_out = geglu ? BN / 2 : BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + bn_out - 1) / bn_out;
  hipLaunchKernelGGL((gemm_kernel_pipe<BM, BN, WM, WN, NST, CONV, BK, PAR>), dim3(tiles_m * p.tiles_n), dim3(WM * WN * 64), 0, st, p);
  return dm4d_check_launch("gemm_kernel_pipe");
}

// the second form addresses A, A2 and W with 32-bit byte offsets from a uniform base and walks K in slabs of 64
__host__ inline bool lin2_ok(const GemmParams& p) {
  return p.K % 64 == 0 && (!p.A2 || p.K1 % 64 == 0) && (uint64_t)p.M * (uint64_t)p.lda * 2u < (1ull << 32) &&
         (!p.A2 || (uint64_t)p.M * (uint64_t)p.lda2 * 2u < (1ull << 32)) &&
         (uint64_t)(2 * (uint64_t)p.N) * (uint64_t)p.ldw * 2u < (1ull << 32);
}

template <int BM, int BN, int WM, int WN, int NST, int BK = 64, int PAR = 0>
int launch_lin2(hipStream_t st, GemmParams& p) {
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const int bn_out = geglu ? BN / 2 : BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + bn_out - 1) / bn_out;
  hipLaunchKernelGGL((gemm_lin2_kernel<BM, BN, WM, WN, NST, BK, PAR>), dim3(tiles_m * p.tiles_n), dim3(WM * WN * 64), 0, st, p);
  return dm4d_check_launch("gemm_lin2_kernel");
}

// out = epilogue(ws[0] + ws[1] + ws[2]) for the split strip convolution: 8 columns per thread, 16-byte stores.
// PAR = 2 (precision "fp16"): fp16 bias, fp32 row bias / residual (DM4D_EPI_F32SIDE), fp32 (DM4D_EPI_F32OUT) or fp16 output.
template <int PAR = 0>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  const int nv = p.N / 8;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)p.M * nv) return;
  const int m = (int)(id / nv), n = (int)(id % nv) * 8;
  const int64_t plane = (int64_t)p.M * p.N;
  const float* w = p.ws + (int64_t)m * p.N + n;
  float v[8];
  {
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(w), a1 = *reinterpret_cast<const f32x4_t*>(w + 4);
    v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3];
    v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
  }
  for (int s = 1; s < p.splits; ++s) {
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(w + s * plane), a1 = *reinterpret_cast<const f32x4_t*>(w + s * plane + 4);
    v[0] += a0[0]; v[1] += a0[1]; v[2] += a0[2]; v[3] += a0[3];
    v[4] += a1[0]; v[5] += a1[1]; v[6] += a1[2]; v[7] += a1[3];
  }
  const bool f32side = PAR == 2 && (p.flags & DM4D_EPI_F32SIDE) != 0;
  auto side = [&](const u16* base, int64_t off, float* t) {
    if (f32side) {
      const float* f = reinterpret_cast<const float*>(base) + off;
      const f32x4_t t0 = *reinterpret_cast<const f32x4_t*>(f), t1 = *reinterpret_cast<const f32x4_t*>(f + 4);
      t[0] = t0[0]; t[1] = t0[1]; t[2] = t0[2]; t[3] = t0[3];
      t[4] = t1[0]; t[5] = t1[1]; t[6] = t1[2]; t[7] = t1[3];
    } else if constexpr (PAR == 2) {
      unpack8h(ldg16(base + off), t);
    } else {
      unpack8(ldg16(base + off), t);
    }
  };
  float t[8];
  if (p.bias) {
    if constexpr (PAR == 2) unpack8h(ldg16(p.bias + n), t);
    else unpack8(ldg16(p.bias + n), t);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += t[e];
  }
  if (p.rowbias) {
    side(p.rowbias, (int64_t)(m / p.rows_per_rb) * p.ld_rb + n, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += t[e];
  }
  if (p.res) {
    side(p.res, (int64_t)m * p.ld_res + n, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += t[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
  if (PAR == 2 && (p.flags & DM4D_EPI_F32OUT)) {
    float* cf = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
    const f32x4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4_t*>(cf) = o0;
    *reinterpret_cast<f32x4_t*>(cf + 4) = o1;
  } else if constexpr (PAR == 2) {
    stg16(p.C + (int64_t)m * p.ldc + n, pack8h(v));
  } else {
    stg16(p.C + (int64_t)m * p.ldc + n, pack8(v));
  }
}

// Split-K applies to stride-1 convolutions on small images (the 9x5 level of the UNet: M = B*45 rows against a
// 11520- or 23040-deep K).  The rule looks at the per-image geometry only, never at the batch, so a frame-sharded
// run (fewer frames per rank) sums in the same order as the unsharded one.
__host__ inline bool strip_split_ok(const GemmParams& p) {
  return p.H * p.W <= 64 && p.Cin >= 512 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.res || (p.ld_res & 7) == 0) &&
         (!p.rowbias || (p.ld_rb & 7) == 0) && (p.flags & ~(DM4D_EPI_F32OUT | DM4D_EPI_F32SIDE | DM4D_EPI_H16)) == 0 &&
         ((p.flags & DM4D_EPI_H16) || p.flags == 0);  // the parity precision (F32OUT / F32SIDE without H16) never splits
}

// the strip kernel addresses A and W with 32-bit byte offsets from a uniform base
__host__ inline bool strip2_ok(const GemmParams& p) {
  return (uint64_t)p.M * (uint64_t)p.Cin * 2u < (1ull << 32) && (uint64_t)p.N * (uint64_t)p.ldw * 2u < (1ull << 32);
}

template <int BM, int BN, int WM, int WN, int PAR = 0>
int launch_strip2(hipStream_t st, GemmParams& p) {
  if (!strip2_ok(p)) return DM4D_ERR_ARG;  // 4 GiB or more of input or weights: the gather kernels take such a launch
  if (BN % 64 != 0 && p.splits > 1) return DM4D_ERR_ARG;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  if (p.splits < 1) p.splits = 1;
  hipLaunchKernelGGL((conv_strip2_kernel<BM, BN, WM, WN, 3, PAR>), dim3(tiles_m * p.tiles_n * p.splits), dim3(WM * WN * 64), 0, st, p);
  int rc = dm4d_check_launch("conv_strip2_kernel");
  if (rc || p.splits == 1) return rc;
  const int64_t nthreads = (int64_t)p.M * (p.N / 8);
  hipLaunchKernelGGL((splitk_reduce_kernel<PAR>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, p);
  return dm4d_check_launch("splitk_reduce_kernel");
}

// Phase-decomposed x2 upsampling convolution: grid = 4 phases x tiles, weights [4][N][4 Cin] from up2x_prepare_kernel
template <int BM, int BN, int WM, int WN, int PAR = 0>
int launch_up2x(hipStream_t st, GemmParams& p) {
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.splits = 1;
  hipLaunchKernelGGL((conv_strip2_kernel<BM, BN, WM, WN, 2, PAR>), dim3(4 * tiles_m * p.tiles_n), dim3(WM * WN * 64), 0, st, p);
  return dm4d_check_launch("conv_strip2_kernel<up2x>");
}

// Wp[phase = 2 py + px][n][(dy, dx, ci)] = sum of W[n][(ky, kx, ci)] over the 3x3 taps that read low-resolution pixel
// (dy, dx) of the phase: rows py = 0: {0} | {1, 2}, py = 1: {0, 1} | {2}; same along x.  fp32 sums, one rounding to bf16 (H16: fp16
// weights in, one rounding to fp16).
template <bool H16 = false>
__global__ __launch_bounds__(256) void up2x_prepare_kernel(const u16* W, u16* Wp, int N, int Cin) {
  const int64_t total = (int64_t)4 * N * 4 * Cin;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int ci = (int)(id % Cin);
  int64_t t = id / Cin;
  const int dx = (int)(t & 1), dy = (int)((t >> 1) & 1);
  t >>= 2;
  const int n = (int)(t % N), phase = (int)(t / N);
  const int py = phase >> 1, px = phase & 1;
  // taps of the 3-tap axis that land on low-resolution offset d of phase ph: ph 0: d 0 <- {0}, d 1 <- {1, 2}; ph 1: d 0 <- {0, 1}, d 1 <- {2}
  const int ky_lo = py == 0 ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2), ky_hi = py == 0 ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
  const int kx_lo = px == 0 ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2), kx_hi = px == 0 ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
  float acc = 0.f;
  for (int ky = ky_lo; ky <= ky_hi; ++ky)
    for (int kx = kx_lo; kx <= kx_hi; ++kx) {
      const u16 w = W[(int64_t)n * 9 * Cin + (ky * 3 + kx) * Cin + ci];
      acc += H16 ? h2f(w) : bf2f(w);
    }
  Wp[id] = H16 ? f2h(acc) : f2bf(acc);
}

int g_tune_cfg = 0;  // 0 = heuristic; otherwise a kernel-configuration id (tuning hook, dm4d_tune_set_gemm_config)

// Kernel configurations.  glds: K-slab 64, 2 LDS stages, 4 waves.  pipe: K-slab 32, 3-4 stages, 4 or 8 waves.
template <bool CONV, int PAR = 0>
int launch_by_id(int id, hipStream_t st, GemmParams& p) {
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const bool k64 = CONV ? (p.Cin % 64 == 0) : (p.K % 64 == 0 && (!p.A2 || p.K1 % 64 == 0));
  if (id >= 1 && id <= 4 && !k64) return DM4D_ERR_ARG;
  switch (id) {
    case 1: return launch_cfg<128, 128, 2, 2, CONV, true, PAR>(st, p);
    case 2: return launch_cfg<256, 64, 4, 1, CONV, true, PAR>(st, p);
    case 3: return launch_cfg<128, 64, 4, 1, CONV, true, PAR>(st, p);
    case 4: return geglu ? DM4D_ERR_ARG : launch_cfg<64, 64, 2, 2, CONV, true, PAR>(st, p);
    case 13: return launch_pipe<256, 256, 2, 4, 4, CONV, 32, PAR>(st, p);
    case 14: return launch_pipe<256, 128, 4, 2, 3, CONV, 32, PAR>(st, p);
    // 16 waves (one 1024-thread workgroup per CU, four waves per SIMD like two 8-wave workgroups) on a 256x256 tile: the
    // per-wave work of id 14 (64x64) with 2/3 of its L2->LDS bytes per flop
    case 20: return launch_pipe<256, 256, 4, 4, 4, CONV, 32, PAR>(st, p);
    // 320-wide tile (see id 35), K-slab 64, 2 stages = 144 KB; per-wave tile 64 x 160 = 5 column blocks: no GEGLU pairing
    case 46: return (k64 && !geglu) ? launch_pipe<256, 320, 4, 2, 2, CONV, 64, PAR>(st, p) : DM4D_ERR_ARG;
    // stride-1 3x3 convolutions on the strip kernel (same K order on every tile, so the choice never changes a result)
    case 31: case 32: case 33: case 34: case 35: case 36: case 37:
      if constexpr (CONV) {
        if (!(k64 && p.stride == 1 && p.pad == 1 && !p.upsample && p.Ho == p.H && p.Wo == p.W)) return DM4D_ERR_ARG;
        if (id == 31) return launch_strip2<128, 128, 2, 2, PAR>(st, p);
        if (id == 32) return launch_strip2<256, 128, 4, 2, PAR>(st, p);
        if (id == 33) return launch_strip2<128, 64, 4, 1, PAR>(st, p);
        // every channel count of an SD-class UNet is a multiple of 320: a 320-wide tile reads the A strip once per
        // kernel row for N = 320 (level 0) and feeds 40 MFMAs per wave between two barriers
        if (id == 35) return launch_strip2<256, 320, 4, 2, PAR>(st, p);
        // 160-wide tiles for N = 320 (two column tiles, no padding): 4 waves x (32 x 160) at 78 KB = two workgroups per CU, or 8 waves
        if (id == 36) return launch_strip2<128, 160, 4, 1, PAR>(st, p);
        if (id == 37) return launch_strip2<256, 160, 8, 1, PAR>(st, p);
        return launch_strip2<256, 256, 2, 4, PAR>(st, p);
      } else {
        return DM4D_ERR_ARG;
      }
    // Linear layers on gemm_lin2_kernel: 61 = 256x128, 8 waves, K-slab 64, 3 stages; 63 = 128x128
    // with 4 waves (2 workgroups per CU); 64 = 128x128 with 8 waves; 65 = 256x128, 8 waves, K-slab 32, 3 stages = 74 KB (two
    // workgroups per CU); 67 = 256x256 on eight waves (128x64 per wave), K-slab 64, 2 stages = 128 KB
    case 61: case 63: case 64: case 65: case 67: case 69:
      if constexpr (!CONV) {
        if (!lin2_ok(p)) return DM4D_ERR_ARG;
        if (id == 67) return launch_lin2<256, 256, 2, 4, 2, 64, PAR>(st, p);
        if (id == 65) return launch_lin2<256, 128, 4, 2, 3, 32, PAR>(st, p);
        if (id == 61) return launch_lin2<256, 128, 4, 2, 3, 64, PAR>(st, p);
        // N = 320 on a tall problem (level 0: proj_in, attention output, proj_out) without padded columns: 128x160 on 4 waves
        // (32 x 160 per wave) at 74 KB = two workgroups per CU
        if (id == 69) return geglu ? DM4D_ERR_ARG : launch_lin2<128, 160, 4, 1, 2, 64, PAR>(st, p);
        if (id == 63) return launch_lin2<128, 128, 2, 2, 2, 64, PAR>(st, p);
        return launch_lin2<128, 128, 4, 2, 3, 64, PAR>(st, p);
      } else {
        return DM4D_ERR_ARG;
      }
    case 21: return launch_cfg<256, 128, 2, 2, CONV, false, PAR>(st, p);
    case 22: return launch_cfg<128, 128, 2, 2, CONV, false, PAR>(st, p);
    case 23: return launch_cfg<256, 64, 4, 1, CONV, false, PAR>(st, p);
    case 24: return launch_cfg<128, 64, 4, 1, CONV, false, PAR>(st, p);
    default: return DM4D_ERR_ARG;
  }
}

// Heuristic (tuned on the UNet shapes at 72x40 latents, profiles/r01_gemm_tune.log)
template <bool CONV>
int choose_cfg(const GemmParams& p) {
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const bool k64 = CONV ? (p.Cin % 64 == 0) : (p.K % 64 == 0 && (!p.A2 || p.K1 % 64 == 0));
  const bool n128 = geglu || (p.N % 128 == 0) || (p.N > 1024);
  if (!k64) {  // K-slab 32 register-staged fallback
    const long tiles_big = (long)((p.M + 255) / 256) * ((p.N + (geglu ? 63 : 127)) / (geglu ? 64 : 128));
    if (n128) return tiles_big >= 384 ? 21 : 22;
    return (long)((p.M + 255) / 256) * ((p.N + 63) / 64) >= 384 ? 23 : 24;
  }
  const int bn = geglu ? 64 : 128;  // output columns of a 128-wide B tile
  const long tm256 = (p.M + 255) / 256, tm128 = (p.M + 127) / 128, tn = (p.N + bn - 1) / bn;
  if (!CONV) {
    // N = 320 on a tall problem (level 0: proj_in, attention output projection, proj_out; the feed-forward's output projection when
    // the fused kernel is off): two 160-wide column tiles, no padded third tile -- cold-cache sweep profiles/r03_lin_160_tiles.log:
    // 59.2 vs 70.0 us at K = 320 and CFG batch 32, 85.6 vs 92.8 at 48; 126 vs 141 / 197 vs 200 at K = 1280 (where the 320-wide
    // tiles, ids 46 / 62 of round 2, used to be ahead; 46 stays for inputs the second form's 32-bit offsets cannot address)
    if (!geglu && p.N == 320 && tm256 >= 256) {
      if (lin2_ok(p)) return 69;
      if (p.K >= 1024) return 46;
    }
#ifndef DM4D_NO_STACK_CFG  // (-DDM4D_NO_STACK_CFG: the table as tuned on single tasks, for the A/B)
    // Round 6: the launches of a 2-task stack (CFG batch 64 / 96: 720 / 1080 row tiles of 256 at level 0) -- sweep of every id on those
    // shapes, profiles/r06_gemm_tune_stacks.log: the level-0 QKV projection (N = 960, K = 320) on the 320-wide tile (three column tiles,
    // no padded fourth: 209 -> 181 us, 301 -> 262 us), level 1's feed-forward output projection on the 160-wide one (197 -> 181, 288 -> 276)
    if (!geglu && k64 && p.K == 320 && p.N == 960 && tm256 >= 700) return 46;
    if (!geglu && p.N == 640 && p.K >= 2560 && tm256 >= 180 && lin2_ok(p)) return 69;
#endif
    // deep-K layers (K >= 1280): the second form (gemm_lin2_kernel), bit-identical, -4..-17 % per launch
    // (profiles/r02_lin2_ab.log): 128x128 tiles with two workgroups per CU wherever they fill the chip, the 8-wave
    // 3-stage 128x128 tile for the few-row, very deep output projections of the deepest level, and the 256x128 K-slab-64
    // tile for that level's GEGLU projection.  Shorter K needs two resident workgroups (a tile is mostly prologue and
    // epilogue): see id 65 below.
    if (lin2_ok(p)) {
      // 256x256 tiles on 8 waves (128x64 per wave: 6 fragment reads feed 8 MFMAs instead of 4 feeding 4), one workgroup per
      // CU.  Chosen from TWO sweeps of every id (all bit-identical): the usual timing loop (profiles/r02_lin_tiles_256.log) and
      // single launches after a cache flush with only the activations re-touched (profiles/r02_lin_cold.log) -- the state a
      // layer meets inside a UNet pass, where this tile's exposed prologue costs more.  It wins both ways on the deep-K wide
      // layers (K >= 1280: GEGLU projection of level 2 182 -> 156 us hot, 176 -> 156 cold; QKV of level 2 at CFG batch 48
      // 95 -> 80 / 98 -> 82) and on the K = 640 ones only when the rounds of 256 tiles are nearly full; at K = 320 the
      // two-workgroup 74 KB tile (id 65) is 8 % ahead cold and stays.
      {
        const long nw = geglu ? 2L * p.N : p.N, tn256 = (nw + 255) / 256, t = tm256 * tn256;
        const double fill = (double)t / (double)(((t + 255) / 256) * 256) * (double)nw / (double)(tn256 * 256);
        if (geglu && ((p.K >= 1280 && fill >= 0.85) || (p.K >= 640 && fill >= 0.95))) return 67;
        // (round 6, second sweep of the stacked launches: id 67 for level 1's GEGLU projection at fill 0.94 and id 61 for level 2's K = 5120
        // output projection were 6-8 % ahead per launch in the timing loop and 0.15 ms BEHIND over the Linear family of a bench step:
        // profiles/r06_stackcfg2.log; not taken)
        if (!geglu && p.K >= 640 && p.N >= 1280 && fill >= 0.85) return 67;
        // one partial round (160-256 tiles) of the N = 1280 projections of level 2 at CFG batch 48: 42 vs 45 us, 122 vs 135 us
        if (!geglu && p.K >= 1280 && p.N == 1280 && t >= 160 && t <= 256) return 67;
      }
      if (geglu) {
        if (tm256 <= 12 && p.K >= 1280) return 61;
      } else if (p.K >= 1280 && p.N >= 640) {
        if (tm128 * ((p.N + 127) / 128) >= 256) return 63;
        if (p.K >= 2560) return 64;
      }
    }
    // Linear layers stream A once with little reuse (K = C or 4C): they are bound by L2->LDS bytes and DMA latency,
    // so the 8-wave 256x128 tile with 2 slabs of DMA in flight wins whenever it still fills the chip (1.2-1.35x)
    const long t = tm256 * tn;  // one 8-wave workgroup per CU => 256 slots per round; avoid a mostly empty last round
    if (t >= 256 && 5 * t >= 4 * ((t + 255) / 256) * 256) {
      // the same 74 KB geometry (two workgroups per CU) in the second form: -2..-9 % on the GEGLU projections of levels 0-2,
      // -9..-16 % on the K = 640 layers of level 1, +-1 % on the narrow K = 320 layers; the wide K = 320 QKV projection
      // (N = 960) is the one shape where it is not ahead at both batch sizes (profiles/r02_lin2_ab.log, id 65 vs auto)
      // (id 61, the 147 KB three-stage tile, for the residual layers with K <= 640 -- ahead in the cold sweep, behind in the timing
      // loop -- measured in a bench step: Linear family 27.6 -> 27.8 ms, profiles/r02_lin_heuristic_cold_ab.log; not taken)
      if (lin2_ok(p) && (geglu || p.K >= 640 || p.N <= 640)) return 65;
      return 14;
    }
  } else {
    // stride-1 convs: the strip kernels stage A once per kernel row (profiles/r01_conv_strip.log)
    if (p.stride == 1 && p.pad == 1 && !p.upsample && p.Ho == p.H && p.Wo == p.W && strip2_ok(p)) {
      if (!n128) {
        // N = 320 on a tall problem (level 0 of the UNet): two 160-wide column tiles, no padded columns and a third of the A re-reads
        // of the 64-wide tile -- 8 waves on 256 rows when the last round of 256 workgroups is at least half full, else 4 waves on
        // 128 rows with two workgroups per CU (-6..-12 % per launch against ids 33 / 35 at CFG batch 32 and 48, cold-cache sweep
        // profiles/r03_strip_160_tiles.log); same K order as every strip kernel, so the choice never changes a result
        if (p.N == 320 && tm256 >= 256) {
#ifndef DM4D_NO_STACK_CFG
          // round 6, stacked launches (profiles/r06_gemm_tune_stacks.log): with 720 / 1080 row tiles the 320-wide tile -- the A strip staged
          // once per kernel row for all of N -- is ahead of the 160-wide ones: 320 -> 320 292 -> 269 us, 960 -> 320 845 -> 758 / 1298 -> 1196,
          // 640 -> 320 570 -> 510 / 874 -> 817 (at CFG batch 32 / 48 the 160-wide tiles stay: r03_strip_160_tiles.log)
          if (tm256 >= 700) return 35;
#endif
          const long t2 = tm256 * 2, last = t2 % 256;
          return (last == 0 || last >= 128) ? 37 : 36;
        }
        if (tm128 * ((p.N + 63) / 64) >= 256) return 33;
      } else {
        const long t = tm256 * tn;
        // (round 6, second sweep of the stacked launches, two passes of 12 launches, profiles/r06_gemm_tune_stacks_p1.log / _p2.log: the
        // 160-wide tiles for level 1 (N = 640) and 256x128 for level 2 at CFG batch 96 are 4-7 % ahead per launch and take 0.7 ms off the
        // convolution family of a one-stack-at-a-time pass -- and ADD 0.2-0.3 ms to the bench step with three stacks in flight
        // (profiles/r06_stackcfg3.log): not taken)
        // N = 640 (level 1) when two 320-wide column tiles make ONE nearly full round of 256 workgroups (CFG batch 32: 180): the A strip
        // is read twice instead of five times, -3..-9 % per launch in both cold sweeps (r02_strip_cold.log, r03_strip_160_tiles.log);
        // at batch 48 the same tile needs a second, nearly empty round and loses 20 %
        if (p.N == 640 && p.Cin >= 640 && tm256 * 2 >= 160 && tm256 * 2 <= 256) return 35;
        if (p.N % 256 == 0 && tm256 * (p.N / 256) >= 160) return 34;
        // one partial round of 256x128 tiles against a nearly full round of 128x128 tiles at two workgroups per CU (level 2 at CFG
        // batch 32: 230 vs 450 of 512): the small tile is 4-6 % ahead in both cold-cache sweeps (r02_strip_cold.log, r03_strip_160_tiles.log)
        if (t <= 256 && tm128 * tn >= 384 && tm128 * tn <= 512) return 31;
        if (t >= 200 && 5 * t >= 4 * ((t + 255) / 256) * 256) return 32;
        if (tm128 * tn >= 256) return 31;
      }
    }
    // upsample-fused convs (Upsample2D): the gather reads every input pixel four times, so tiles that cut the A traffic
    // win: 320-wide for N = 640 (578 vs 698-760 us), 16-wave 256x256 for N = 1280 (634 vs 685-740 us); all of these walk K
    // in the same (ky, kx, ci) order as the other gather kernels
    if (p.upsample && k64) {
      if (p.N % 320 == 0 && p.N <= 640 && tm256 >= 64) return 46;
      if (p.N % 256 == 0 && tm256 * (p.N / 256) >= 256) return 20;
    }
    // other convs: 128x128 / 2 workgroups per CU is best except for wide, tall problems
    if (!geglu && p.N % 256 == 0 && tm256 * (p.N / 256) >= 384) return 13;
  }
  if (n128) {
    if (tm128 * tn >= 256) return 1;
    if (geglu) return 3;
    return tm128 * tn >= 200 ? 3 : 4;  // deepest UNet level: shrink the tile until the grid covers the 256 CUs
  }
  return tm256 * ((p.N + 63) / 64) >= 384 ? 2 : 3;
}

#ifndef DM4D_GEMM_H16_TU
// Parity-precision launches (DM4D_EPI_F32SIDE / DM4D_EPI_SPLITOUT: fp32 side inputs, two-term output) run on their own
// instantiations of a few tile geometries (PAR = true), chosen by the tail of the heuristic above; every kernel of this file walks
// K in the same order, so the choice never changes a result.  The fast kernels do not carry that epilogue code.
template <bool CONV>
int launch_par(hipStream_t st, GemmParams& p) {
  const bool geglu = (p.flags & DM4D_EPI_GEGLU) != 0;
  const bool k64 = CONV ? (p.Cin % 64 == 0) : (p.K % 64 == 0 && (!p.A2 || p.K1 % 64 == 0));
  const bool n128 = geglu || (p.N % 128 == 0) || (p.N > 1024);
  const long tm128 = (p.M + 127) / 128, tm256 = (p.M + 255) / 256, tn = (p.N + (geglu ? 63 : 127)) / (geglu ? 64 : 128);
  if constexpr (CONV) {
    if (k64 && p.stride == 1 && p.pad == 1 && !p.upsample && p.Ho == p.H && p.Wo == p.W && strip2_ok(p)) {
      if (!n128) return launch_strip2<128, 64, 4, 1, true>(st, p);
      if (tm256 * tn >= 200) return launch_strip2<256, 128, 4, 2, true>(st, p);
      return launch_strip2<128, 128, 2, 2, true>(st, p);
    }
    if (!k64) return dm4d_set_error(DM4D_ERR_ARG, "conv3x3: parity-precision launches need Cin % 64 == 0 (a two-term operand of Cin / 2 channels)");
  } else {
    if (!k64) {  // K-slab 32, register staged (P V of the VAE mid block: K = 3 Lp with Lp a multiple of 32)
      if (n128) return launch_cfg<128, 128, 2, 2, false, false, true>(st, p);
      return launch_cfg<128, 64, 4, 1, false, false, true>(st, p);
    }
    // Linear layers: the tiles the fast precision picks for this shape (K here is the doubled K of the two-term operand), PAR epilogue
    if (lin2_ok(p)) {
      switch (choose_cfg<false>(p)) {
        case 67: return launch_lin2<256, 256, 2, 4, 2, 64, true>(st, p);
        case 65: return launch_lin2<256, 128, 4, 2, 3, 32, true>(st, p);
        case 61: return launch_lin2<256, 128, 4, 2, 3, 64, true>(st, p);
        case 69: if (!geglu) return launch_lin2<128, 160, 4, 1, 2, 64, true>(st, p); break;
        case 63: return launch_lin2<128, 128, 2, 2, 2, 64, true>(st, p);
        case 64: return launch_lin2<128, 128, 4, 2, 3, 64, true>(st, p);
        default: break;
      }
    }
  }
  if (n128) {
    if (tm128 * tn >= 256 || geglu) return launch_cfg<128, 128, 2, 2, CONV, true, true>(st, p);
    return tm128 * tn >= 200 ? launch_cfg<128, 64, 4, 1, CONV, true, true>(st, p) : launch_cfg<64, 64, 2, 2, CONV, true, true>(st, p);
  }
  return launch_cfg<128, 64, 4, 1, CONV, true, true>(st, p);
}

template <bool CONV>
int launch(hipStream_t st, GemmParams& p) {
  p.splits = 1;
  if (p.flags & (DM4D_EPI_F32SIDE | DM4D_EPI_SPLITOUT)) return launch_par<CONV>(st, p);
  if (g_tune_cfg) {
    int rc = launch_by_id<CONV>(g_tune_cfg, st, p);
    if (rc == DM4D_ERR_ARG) return dm4d_set_error(DM4D_ERR_ARG, "gemm: forced configuration does not support this shape");
    return rc;
  }
  if constexpr (CONV) {
    if (p.ws && p.stride == 1 && p.pad == 1 && !p.upsample && p.Ho == p.H && p.Wo == p.W && p.Cin % 64 == 0 &&
        strip_split_ok(p) && strip2_ok(p)) {
      p.splits = 3;
      return launch_by_id<CONV>(31, st, p);  // 128x128 strip tiles x 3 kernel rows
    }
  }
  return launch_by_id<CONV>(choose_cfg<CONV>(p), st, p);
}

#else  // DM4D_GEMM_H16_TU: this translation unit (gemm_h16.hip) instantiates the PAR = 2 kernels only
// Precision "fp16": every tile geometry of the fast precision with fp16 operands (v_mfma_f32_32x32x16_f16), chosen by the same
// heuristic -- the problem shapes are the fast precision's (K is not doubled) -- including the split over the kernel rows at the
// 9x5 level.  fp32 side inputs and outputs go through the general epilogue loop (gemm_common.h, PAR = 2).
template <bool CONV>
int launch_h16(hipStream_t st, GemmParams& p) {
  p.splits = 1;
  p.flags |= DM4D_EPI_H16;
  if constexpr (CONV) {
    if (p.ws && p.stride == 1 && p.pad == 1 && !p.upsample && p.Ho == p.H && p.Wo == p.W && p.Cin % 64 == 0 &&
        strip_split_ok(p) && strip2_ok(p)) {
      p.splits = 3;
      return launch_by_id<CONV, 2>(31, st, p);
    }
  }
  return launch_by_id<CONV, 2>(choose_cfg<CONV>(p), st, p);
}
#endif

}  // namespace

#ifdef DM4D_GEMM_H16_TU
extern "C" int dm4d_gemm_f16(void* stream, const void* A, int64_t lda, const void* A2, int64_t lda2, int K1, const void* W,
                             int64_t ldw, void* C, int64_t ldc, int M, int N, int K, const void* bias, const void* rowbias,
                             int64_t ld_rowbias, int rows_per_rowbias, const void* residual, int64_t ld_res, unsigned flags,
                             float out_scale, int scale_cols, float col_scale) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return dm4d_set_error(DM4D_ERR_ARG, "gemm_f16: null pointer or empty shape");
  if (K % 32 != 0) return dm4d_set_error(DM4D_ERR_ARG, "gemm_f16: K must be a multiple of 32");
  if ((lda & 7) || (ldw & 7)) return dm4d_set_error(DM4D_ERR_ARG, "gemm_f16: lda/ldw must be multiples of 8");
  if (A2 && ((K1 % 32) != 0 || K1 <= 0 || K1 >= K || (lda2 & 7)))
    return dm4d_set_error(DM4D_ERR_ARG, "gemm_f16: bad split-A arguments");
  if (rowbias && rows_per_rowbias <= 0) return dm4d_set_error(DM4D_ERR_ARG, "gemm_f16: rows_per_rowbias <= 0");
  if ((flags & DM4D_EPI_GEGLU) && (N % 32 != 0)) return dm4d_set_error(DM4D_ERR_ARG, "gemm_f16: GEGLU needs N % 32 == 0");
  if (flags & ~(DM4D_EPI_GEGLU | DM4D_EPI_SILU | DM4D_EPI_F32OUT | DM4D_EPI_F32SIDE))
    return dm4d_set_error(DM4D_ERR_ARG, "gemm_f16: only GEGLU, SILU, F32OUT and F32SIDE apply");
  if (scale_cols < 0 || scale_cols > N || (scale_cols & 7)) return dm4d_set_error(DM4D_ERR_ARG, "gemm_f16: scale_cols must be a multiple of 8 in [0, N]");
  GemmParams p{};
  p.A = (const u16*)A; p.lda = lda; p.A2 = (const u16*)A2; p.lda2 = lda2; p.K1 = K1;
  p.Wt = (const u16*)W; p.ldw = ldw; p.C = (u16*)C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.bias = (const u16*)bias; p.rowbias = (const u16*)rowbias; p.ld_rb = ld_rowbias; p.rows_per_rb = rows_per_rowbias;
  p.res = (const u16*)residual; p.ld_res = ld_res; p.flags = flags; p.out_scale = out_scale;
  p.scale_cols = scale_cols; p.col_scale = col_scale;
  return launch_h16<false>((hipStream_t)stream, p);
}

extern "C" int dm4d_conv3x3_nhwc_f16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y, int Ho,
                                     int Wo, int Cout, int stride, int pad, int upsample, const void* bias, const void* rowbias,
                                     int64_t ld_rowbias, const void* residual, int64_t ld_res, float out_scale, unsigned flags,
                                     void* ws, size_t ws_bytes) {
  if (!X || !Wt || !Y || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "conv3x3_f16: null pointer or empty shape");
  if (Cin % 32 != 0) return dm4d_set_error(DM4D_ERR_ARG, "conv3x3_f16: Cin must be a multiple of 32 (pad the input)");
  if (stride != 1 && stride != 2) return dm4d_set_error(DM4D_ERR_ARG, "conv3x3_f16: stride must be 1 or 2");
  if (upsample && stride != 1) return dm4d_set_error(DM4D_ERR_ARG, "conv3x3_f16: upsample needs stride 1");
  if (flags & ~(DM4D_EPI_F32OUT | DM4D_EPI_F32SIDE))
    return dm4d_set_error(DM4D_ERR_ARG, "conv3x3_f16: only DM4D_EPI_F32OUT and DM4D_EPI_F32SIDE apply to a convolution");
  if ((int64_t)B * H * W * Cin >= (int64_t)1 << 31 || (int64_t)B * Ho * Wo * Cout >= (int64_t)1 << 31)
    return dm4d_set_error(DM4D_ERR_ARG, "conv3x3_f16: tensors of 2^31 or more elements are not supported (split the batch)");
  GemmParams p{};
  p.A = (const u16*)X; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.pad = pad;
  p.upsample = upsample;
  p.Wt = (const u16*)Wt; p.ldw = (int64_t)9 * Cin; p.C = (u16*)Y; p.ldc = Cout;
  p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
  p.bias = (const u16*)bias; p.rowbias = (const u16*)rowbias; p.ld_rb = ld_rowbias; p.rows_per_rb = Ho * Wo;
  p.res = (const u16*)residual; p.ld_res = ld_res; p.flags = flags; p.out_scale = out_scale;
  const bool strip = stride == 1 && pad == 1 && !upsample && Ho == H && Wo == W && Cin % 64 == 0;
  GemmParams q = p;
  q.flags |= DM4D_EPI_H16;
  const size_t need = strip && strip_split_ok(q) ? (size_t)3 * B * Ho * Wo * Cout * sizeof(float) : 0;  // = dm4d_conv3x3_ws_bytes
  p.ws = (ws && need && ws_bytes >= need) ? (float*)ws : nullptr;
  return launch_h16<true>((hipStream_t)stream, p);
}

extern "C" int dm4d_conv_up2x_prepare_f16(void* stream, const void* W, void* Wp, int Cout, int Cin) {
  if (!W || !Wp || Cout <= 0 || Cin <= 0) return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x_prepare_f16: null pointer or empty shape");
  const int64_t total = (int64_t)16 * Cout * Cin;
  hipLaunchKernelGGL(up2x_prepare_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)W, (u16*)Wp, Cout, Cin);
  return dm4d_check_launch("up2x_prepare_kernel");
}

extern "C" int dm4d_conv_up2x_nhwc_f16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wp, void* Y, int Cout,
                                       const void* bias, unsigned flags) {
  if (!X || !Wp || !Y || B <= 0 || H <= 0 || W <= 0 || Cout <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x_f16: null pointer or empty shape");
  if (Cin % 64 != 0 || (Cout & 7) != 0)
    return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x_f16: Cin must be a multiple of 64 and Cout of 8 (use conv3x3_f16 with upsample = 1)");
  if (flags & ~DM4D_EPI_F32OUT) return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x_f16: only DM4D_EPI_F32OUT applies");
  if ((int64_t)B * H * W * Cin >= (int64_t)1 << 31 || (int64_t)B * 4 * H * W * Cout >= (int64_t)1 << 31)
    return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x_f16: tensors of 2^31 or more elements are not supported (split the batch)");
  GemmParams p{};
  p.A = (const u16*)X; p.H = H; p.W = W; p.Cin = Cin; p.Ho = H; p.Wo = W; p.stride = 1; p.pad = 1; p.upsample = 0;
  p.Wt = (const u16*)Wp; p.ldw = (int64_t)4 * Cin; p.C = (u16*)Y; p.ldc = Cout;
  p.M = B * H * W; p.N = Cout; p.K = 4 * Cin;
  p.bias = (const u16*)bias; p.rows_per_rb = H * W; p.flags = flags | DM4D_EPI_H16; p.out_scale = 1.0f; p.up_w = W;
  if (!strip2_ok(p)) return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x_f16: input or weights of 4 GiB or more");
  hipStream_t st = (hipStream_t)stream;
  const long tm256 = (p.M + 255) / 256;  // tile choice as dm4d_conv_up2x_nhwc_bf16
  if (Cout % 128 == 0 && tm256 * (Cout / 128) >= 200) return launch_up2x<256, 128, 4, 2, 2>(st, p);
  if (Cout % 128 == 0) return launch_up2x<128, 128, 2, 2, 2>(st, p);
  return launch_up2x<128, 64, 4, 1, 2>(st, p);
}
#else

extern "C" int dm4d_tune_set_gemm_config(int id) {
  g_tune_cfg = id;
  return DM4D_OK;
}

extern "C" int dm4d_gemm_bf16(void* stream, const void* A, int64_t lda, const void* A2, int64_t lda2, int K1,
                              const void* W, int64_t ldw, void* C, int64_t ldc, int M, int N, int K, const void* bias,
                              const void* rowbias, int64_t ld_rowbias, int rows_per_rowbias, const void* residual,
                              int64_t ld_res, unsigned flags, float out_scale) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return dm4d_set_error(DM4D_ERR_ARG, "gemm: null pointer or empty shape");
  if (K % 32 != 0) return dm4d_set_error(DM4D_ERR_ARG, "gemm: K must be a multiple of 32");
  if ((lda & 7) || (ldw & 7)) return dm4d_set_error(DM4D_ERR_ARG, "gemm: lda/ldw must be multiples of 8");
  if (A2 && ((K1 % 32) != 0 || K1 <= 0 || K1 >= K || (lda2 & 7)))
    return dm4d_set_error(DM4D_ERR_ARG, "gemm: bad split-A arguments");
  if (rowbias && rows_per_rowbias <= 0) return dm4d_set_error(DM4D_ERR_ARG, "gemm: rows_per_rowbias <= 0");
  if ((flags & DM4D_EPI_GEGLU) && (N % 32 != 0)) return dm4d_set_error(DM4D_ERR_ARG, "gemm: GEGLU needs N % 32 == 0");
  GemmParams p{};
  p.A = (const u16*)A; p.lda = lda; p.A2 = (const u16*)A2; p.lda2 = lda2; p.K1 = K1;
  p.Wt = (const u16*)W; p.ldw = ldw; p.C = (u16*)C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.bias = (const u16*)bias; p.rowbias = (const u16*)rowbias; p.ld_rb = ld_rowbias; p.rows_per_rb = rows_per_rowbias;
  p.res = (const u16*)residual; p.ld_res = ld_res; p.flags = flags; p.out_scale = out_scale;
  return launch<false>((hipStream_t)stream, p);
}

extern "C" size_t dm4d_conv3x3_ws_bytes(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride, int pad,
                                        int upsample) {
  GemmParams p{};
  p.H = H; p.W = W; p.Cin = Cin; p.N = Cout; p.ldc = Cout;
  const bool strip = stride == 1 && pad == 1 && !upsample && Ho == H && Wo == W && Cin % 64 == 0;
  return strip && strip_split_ok(p) ? (size_t)3 * B * Ho * Wo * Cout * sizeof(float) : 0;
}

static int conv3x3_impl(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y, int Ho, int Wo,
                        int Cout, int stride, int pad, int upsample, const void* bias, const void* rowbias,
                        int64_t ld_rowbias, const void* residual, int64_t ld_res, float out_scale, void* ws,
                        size_t ws_bytes, unsigned flags = 0) {
  if (!X || !Wt || !Y || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "conv3x3: null pointer or empty shape");
  if (Cin % 32 != 0) return dm4d_set_error(DM4D_ERR_ARG, "conv3x3: Cin must be a multiple of 32 (pad the input)");
  if (stride != 1 && stride != 2) return dm4d_set_error(DM4D_ERR_ARG, "conv3x3: stride must be 1 or 2");
  if (upsample && stride != 1) return dm4d_set_error(DM4D_ERR_ARG, "conv3x3: upsample needs stride 1");
  if ((int64_t)B * H * W * Cin >= (int64_t)1 << 31 || (int64_t)B * Ho * Wo * Cout >= (int64_t)1 << 31)
    return dm4d_set_error(DM4D_ERR_ARG, "conv3x3: tensors of 2^31 or more elements are not supported (split the batch)");
  GemmParams p{};
  p.A = (const u16*)X; p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.pad = pad;
  p.upsample = upsample;
  p.Wt = (const u16*)Wt; p.ldw = (int64_t)9 * Cin; p.C = (u16*)Y; p.ldc = Cout;
  p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * Cin;
  p.bias = (const u16*)bias; p.rowbias = (const u16*)rowbias; p.ld_rb = ld_rowbias; p.rows_per_rb = Ho * Wo;
  p.res = (const u16*)residual; p.ld_res = ld_res; p.flags = flags; p.out_scale = out_scale;
  const size_t need = dm4d_conv3x3_ws_bytes(B, H, W, Cin, Ho, Wo, Cout, stride, pad, upsample);
  p.ws = (ws && need && ws_bytes >= need) ? (float*)ws : nullptr;  // without a workspace the un-split kernels run
  return launch<true>((hipStream_t)stream, p);
}

extern "C" int dm4d_conv3x3_nhwc_bf16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y,
                                      int Ho, int Wo, int Cout, int stride, int pad, int upsample, const void* bias,
                                      const void* rowbias, int64_t ld_rowbias, const void* residual, int64_t ld_res,
                                      float out_scale) {
  return conv3x3_impl(stream, X, B, H, W, Cin, Wt, Y, Ho, Wo, Cout, stride, pad, upsample, bias, rowbias, ld_rowbias,
                      residual, ld_res, out_scale, nullptr, 0);
}

extern "C" int dm4d_conv3x3_nhwc_bf16_flags(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y,
                                            int Ho, int Wo, int Cout, int stride, int pad, int upsample, const void* bias,
                                            const void* rowbias, int64_t ld_rowbias, const void* residual, int64_t ld_res,
                                            float out_scale, unsigned flags) {
  if (flags & ~(DM4D_EPI_F32OUT | DM4D_EPI_F32SIDE))
    return dm4d_set_error(DM4D_ERR_ARG, "conv3x3: only DM4D_EPI_F32OUT and DM4D_EPI_F32SIDE apply to a convolution");
  return conv3x3_impl(stream, X, B, H, W, Cin, Wt, Y, Ho, Wo, Cout, stride, pad, upsample, bias, rowbias, ld_rowbias,
                      residual, ld_res, out_scale, nullptr, 0, flags);
}

extern "C" int dm4d_conv_up2x_prepare_bf16(void* stream, const void* W, void* Wp, int Cout, int Cin) {
  if (!W || !Wp || Cout <= 0 || Cin <= 0) return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x_prepare: null pointer or empty shape");
  const int64_t total = (int64_t)16 * Cout * Cin;
  hipLaunchKernelGGL(up2x_prepare_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)W, (u16*)Wp, Cout, Cin);
  return dm4d_check_launch("up2x_prepare_kernel");
}

extern "C" int dm4d_conv_up2x_nhwc_bf16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wp, void* Y,
                                        int Cout, const void* bias) {
  if (!X || !Wp || !Y || B <= 0 || H <= 0 || W <= 0 || Cout <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x: null pointer or empty shape");
  if (Cin % 64 != 0 || (Cout & 7) != 0)
    return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x: Cin must be a multiple of 64 and Cout of 8 (use conv3x3 with upsample = 1)");
  if ((int64_t)B * H * W * Cin >= (int64_t)1 << 31 || (int64_t)B * 4 * H * W * Cout >= (int64_t)1 << 31)
    return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x: tensors of 2^31 or more elements are not supported (split the batch)");
  GemmParams p{};
  p.A = (const u16*)X; p.H = H; p.W = W; p.Cin = Cin; p.Ho = H; p.Wo = W; p.stride = 1; p.pad = 1; p.upsample = 0;
  p.Wt = (const u16*)Wp; p.ldw = (int64_t)4 * Cin; p.C = (u16*)Y; p.ldc = Cout;
  p.M = B * H * W; p.N = Cout; p.K = 4 * Cin;
  p.bias = (const u16*)bias; p.rows_per_rb = H * W; p.flags = 0; p.out_scale = 1.0f; p.up_w = W;
  if (!strip2_ok(p)) return dm4d_set_error(DM4D_ERR_ARG, "conv_up2x: input or weights of 4 GiB or more");
  hipStream_t st = (hipStream_t)stream;
  // tile choice as for the stride-1 strips: 256-row tiles where a phase alone fills the chip's 256 CUs
  const long tm256 = (p.M + 255) / 256, tm128 = (p.M + 127) / 128;
  if (Cout % 128 == 0 && tm256 * (Cout / 128) >= 200) return launch_up2x<256, 128, 4, 2>(st, p);
  if (Cout % 128 == 0) return launch_up2x<128, 128, 2, 2>(st, p);
  (void)tm128;
  return launch_up2x<128, 64, 4, 1>(st, p);
}

extern "C" int dm4d_conv3x3_nhwc_bf16_ws(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y,
                                         int Ho, int Wo, int Cout, int stride, int pad, int upsample, const void* bias,
                                         const void* rowbias, int64_t ld_rowbias, const void* residual, int64_t ld_res,
                                         float out_scale, void* ws, size_t ws_bytes) {
  return conv3x3_impl(stream, X, B, H, W, Cin, Wt, Y, Ho, Wo, Cout, stride, pad, upsample, bias, rowbias, ld_rowbias,
                      residual, ld_res, out_scale, ws, ws_bytes);
}
#endif  // DM4D_GEMM_H16_TU
