// Flash-style self-attention for gfx950, head_dim 64, bf16 in/out, fp32 online softmax.
//
// Serves both the per-frame (2-D) and the frame-folded view/time (3-D) attention of
// MultiviewTransformerBlock (reference attention.py:68-83): the fold "(b t) hw c -> b (t hw) c" is
// free because activations are token-major, so the kernel just sees batch = B/F, L = F*HW.
//
// Work decomposition: a workgroup = 8 waves, each wave owns 32 query rows; K/V tiles of 64 keys live in two LDS
// stages each.  Math (v_mfma_f32_32x32x16_bf16), per 32-key block:
//   S^T = K Q^T   ("swapped" QK^T): lane (q = lane&31) holds 16 of the 32 scores of its query row,
//                  so the row max needs a single cross-half exchange and the row sum none at all;
//   O^T = V^T P^T: the P^T B-operand is exactly the packed bf16 (v_cvt_pk_bf16_f32) of the S^T
//                  accumulator registers: k-slot e of lane half h is key 4h + (e&3) + 8*(e>>2), and the
//                  two transposing reads (ds_read_b64_tr_b16; keys 4h..4h+3 and 8+4h..) deliver V in that same
//                  order, so there are no lane shuffles between the two MFMAs.
// Main loop (kv_loop_pipelined): K/V tiles go HBM -> LDS by DMA (global_load_lds_dwordx4, one 1-KiB piece = 8 key
// rows per wave and tile, no VGPR round trip, no ds_write), issued at the top of a step and awaited at its end, so
// a whole step of MFMA work covers the latency; QK^T runs one tile ahead of its softmax.  LDS rows are 128 B,
// un-padded (the DMA writes lane-linear), made conflict-free by permuting the 16-byte chunks of a row on the
// SOURCE side: K chunk c of row r sits in slot c ^ ((r >> 1) & 7) (ds_read_b128 fragments), V chunk c in slot
// c ^ (4 * ((r >> 1) & 1)) (tr reads of [4 keys][16 d] blocks).
// Keys may outnumber queries (Lk >= Lq): frame-sharded 3-D attention runs local queries against all-gathered K/V.
// Softmax: an optimistic pass (row max taken from the first tile only) with an exact running-max pass (lazy O/l
// rescale when a row max grows by more than 2^8; register-staged, padded LDS rows) as the in-kernel fallback.
// FOLD = true (dm4d_attention_qscaled_kv_bf16): Q already carries scale*log2(e), the QK^T accumulator starts from -m
// (a constant register set used as the first MFMA's C operand) and P = exp2(S) needs no per-score v_fma.
// The 1-D grid is remapped so that all query tiles of one (batch, head) run on one XCD and share its L2 copy of K/V.
#include "common.h"
#include "dm4d.h"
#include "errors.h"
#ifndef ATTN64_INC  // tuning builds (tools/attn64/ab.py) compile other schedules of the same stream
#define ATTN64_INC "attn64_asm.inc"
#endif
#include ATTN64_INC
#include <stdlib.h>

namespace {

struct AttnParams {
  const u16 *Q, *K, *V;
  u16* O;
  int64_t ldq, ldk, ldv, ldo;
  int L, Lk, heads, nqt;  // L = queries per (batch, head), Lk = keys (== L for plain self-attention)
  float c;         // scale * log2(e)   (unused when Q is pre-scaled)
  int exact_only;  // tuning / debugging: skip the optimistic pass (DM4D_ATTN_EXACT=1)
#ifdef ATTN64_TIMING  // tuning builds only: s_memtime stamps of attn64_kernel's stream (start, loop entry, end) per wave
  unsigned long long* dbg;
#endif
};
#ifdef ATTN64_TIMING
static unsigned long long* g_attn64_dbg = nullptr;
extern "C" void dm4d_attn64_set_debug(void* ptr) { g_attn64_dbg = (unsigned long long*)ptr; }
#endif

constexpr int KV = 64;       // keys per staged tile
constexpr int LDS_LD = 72;   // exact loop, K rows: 64 + 8 pad bf16 = 144 B  (conflict-free ds_read_b128 fragments)
constexpr int LDS_LDV = 96;  // exact loop, V rows: 64 + 32 pad bf16 = 192 B (4 consecutive rows tile the 64 banks)
constexpr int LDS_LDO = 72;  // O staging rows: 64 + 8 pad bf16 = 144 B
constexpr int TILE = KV * 64;  // elements of one un-padded [64 keys][64 d] stage of the pipelined loop
// Two stages per operand ring: one DMA round and one barrier per tile, DMA one tile ahead.  (A four-stage ring with one
// barrier per two tiles, a software-pipelined softmax with sched_group_barrier slots, static priority for the younger half of
// the workgroup and a one-wave-per-SIMD form with 64 query rows per wave were all built and measured in rounds 1-2 at
// 0.88-1.01x of this loop -- DESIGN.md section 4 -- and removed.)
constexpr int RING = 2, AHEAD = 1;
// waves per workgroup (template parameter NW): 8 = 256 query rows per workgroup, one workgroup per CU; 4 = 128 rows, two
// independent workgroups per CU, so that a wave waiting at its workgroup's barrier shares its SIMD with a wave that is not
constexpr float RESCALE_THR = 8.0f;  // in log2 units

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA pieces (global_load_lds_dwordx4: lane i's 16 bytes land at lds_dst + 16 i) as volatile asm.  The builtin form
// makes hipcc put an s_waitcnt vmcnt(0) in front of the next ds_read of the same array (it cannot tell the stages
// apart), i.e. right behind the issue; an asm DMA has no VGPR destination, so hiding it from the compiler's counter
// is register-safe, and its completion is awaited by hand (dma_wait_barrier).  M0 is written and restored inside the
// statement; `lds_dst` and `uniform_base` must be wave-uniform.
__device__ __forceinline__ void dma16(const void* uniform_base, uint32_t byte_off, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(byte_off), "s"(uniform_base), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void dma16(const void* ptr, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(ptr), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* shared_ptr) {
  return (uint32_t)(uintptr_t)(lptr_t)shared_ptr;  // byte offset inside the workgroup's LDS allocation
}

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t b = __builtin_convertvector(v, bf16x2_t);  // v_cvt_pk_bf16_f32
  return *reinterpret_cast<uint32_t*>(&b);
}

// H16 (template parameter of the kernels; precision "fp16", dm4d_attention_qscaled_kv_f16): Q / K / V / O and the probabilities are
// IEEE fp16 instead of bf16 -- v_mfma_f32_32x32x16_f16 and v_cvt_pk_f16_f32 in place of their bf16 twins, same rate and fragment
// layouts, so the loop is the same instruction for instruction.  What changes is the range: fp16 stops at 65504, so the optimistic
// pass subtracts H16_OFF more than the first tile's row maximum (its largest probability is 2^-H16_OFF; later tiles may outgrow the
// first by 2^(16 + H16_OFF) before a probability overflows, values below 2^-14 go subnormal with an absolute error of 2^-25, i.e.
// 2^-17 of the row's largest term) and the row-sum check that sends a workgroup to the exact loop is l < 2^16 instead of 2^60
// (a sum below 2^16 has no term above it; the first tile alone puts l >= 2^-H16_OFF).  The exact loop keeps every probability below
// 2^RESCALE_THR = 256 already.
constexpr float H16_OFF = 8.0f;
// fp32 row sums are taken before (8-wave kernel) or after (attn64: on the matrix pipe) the fp16 rounding of the probabilities; a term in
// [65520, 65536) rounds to +inf while its fp32 sum can still sit below 2^16, so the bound that sends a workgroup to the exact loop is
// 2^15: a row sum below it has no term above 32768, which fp16 holds.
constexpr float H16_L_MAX = 0x1p15f;
template <bool H16>
__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {
  if constexpr (H16) return pack_h2(lo, hi);
  else return cvt_pk_bf16(lo, hi);
}
template <bool H16>
__device__ __forceinline__ f32x16_t mfma16(const bf16x8_t& a, const bf16x8_t& b, const f32x16_t& c) {
  if constexpr (H16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// "Optimistic" softmax variant (default).  tools/probes/issue_probe.hip shows that on a gfx950 SIMD a wave that
// streams MFMAs starves its co-resident waves' VALU instructions, so a tile costs (MFMA cycles + VALU cycles), and at
// head_dim 64 the softmax VALU work (826 of ~1340 cycles per 64-key tile per wave) is the larger half.  A quarter of
// it is the running row max (v_max_f32 costs 6.7 cycles, v_max3_f32 6.2) whose only purpose is to keep exp2() in
// range: bf16 has fp32's exponent range, so P = exp2(s - m) is as accurate for m = "max of the first tile" as for
// the true running max as long as nothing overflows.  So: take m from tile 0, never look at the max again, and
// check the row sums at the end; a workgroup in which some sum left [0, 2^60) (or is NaN) simply redoes its rows
// with the exact running-max loop (SAFE).  Results do not depend on how rows are grouped into waves (no vote).
// ------------------------------------------------------------------------------------------------
template <bool SAFE, bool FOLD, int NW, bool H16 = false>
__device__ __forceinline__ void kv_loop(const AttnParams& p, const u16* Kb, const u16* Vb, u16* Ks, u16* Vs, const u16* v_lane,
                                        const bf16x8_t (&qf)[4], f32x16_t (&o)[2], float& m_run, float& l_run, int tid,
                                        int l31, int lh) {
  const int Lk = p.Lk;
  const float cs = FOLD ? 1.0f : p.c;  // FOLD: Q already carries scale*log2(e)
  constexpr int RPT = 8 / NW;  // key rows per thread and tile (NW * 64 threads stage 64 rows of 8 chunks)
  U4 rk[RPT], rv[RPT];
  const int s_key = tid >> 3, s_c = tid & 7;
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      int key = t * KV + s_key + i * (NW * 8);
      if (key > Lk - 1) key = Lk - 1;
      rk[i] = ldg16(Kb + (int64_t)key * p.ldk + s_c * 8);
      rv[i] = ldg16(Vb + (int64_t)key * p.ldv + s_c * 8);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      *reinterpret_cast<U4*>(Ks + (buf * KV + s_key + i * (NW * 8)) * LDS_LD + s_c * 8) = rk[i];
      *reinterpret_cast<U4*>(Vs + (buf * KV + s_key + i * (NW * 8)) * LDS_LDV + s_c * 8) = rv[i];
    }
  };
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
        bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (buf * KV + kb * 32 + l31) * LDS_LD + j * 16 + lh * 8);
        s[kb] = mfma16<H16>(kf, qf[j], s[kb]);
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
    if (SAFE || t == 0) {
      float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (!SAFE) {
        m_run = mx;  // o and l are still zero: nothing to rescale
      } else if (__any((mx - m_run) * cs > RESCALE_THR)) {
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
      float pv[16];
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(s[kb][r] * cs - mc);
        sum += pv[r];
      }
      l_run += sum;
      bf16x8_t pf[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        U4 w;
        w.x = cvt_pk<H16>(pv[jj * 8 + 0], pv[jj * 8 + 1]);
        w.y = cvt_pk<H16>(pv[jj * 8 + 2], pv[jj * 8 + 3]);
        w.z = cvt_pk<H16>(pv[jj * 8 + 4], pv[jj * 8 + 5]);
        w.w = cvt_pk<H16>(pv[jj * 8 + 6], pv[jj * 8 + 7]);
        pf[jj] = *reinterpret_cast<bf16x8_t*>(&w);
      }
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const u16* vp = v_lane + (buf * KV + kb * 32 + jj * 16) * LDS_LDV + db * 32;
          s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)vp);
          s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vp + 8 * LDS_LDV));
          s16x8_t v01 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
          bf16x8_t vf = *reinterpret_cast<bf16x8_t*>(&v01);
          o[db] = mfma16<H16>(vf, pf[jj], o[db]);
        }
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined optimistic loop (the one that normally runs).  Step t: issue the DMA of K(t+2) and V(t+1) into
// the stages that held K(t) / V(t-1); QK^T of tile t+1, softmax + PV of tile t (straight-line code, the compiler
// interleaves MFMAs and VALU work); s_waitcnt vmcnt(0) + s_barrier.  An empty asm that "reads" the step's MFMA
// results sits in front of the wait: without it the scheduler sinks half of the MFMAs below the barrier into the next
// step, which puts the wait a few MFMAs after the issue and exposes the L2/HBM latency once per tile.
// ------------------------------------------------------------------------------------------------
template <bool FOLD, int NW, bool H16 = false>
__device__ __forceinline__ void kv_loop_pipelined(const AttnParams& p, const u16* Kb, const u16* Vb, u16* Ks, u16* Vs,
                                                  const bf16x8_t (&qf)[4], f32x16_t (&o)[2], float& m_run, float& l_run,
                                                  int lane, int wave, int l31, int lh) {
  const int Lk = p.Lk;
  const int nt = (Lk + KV - 1) / KV;
  const float cs = FOLD ? 1.0f : p.c;
  // DMA geometry: a tile is 8 pieces of 8 key rows; wave w moves pieces w, w + NW, ...; lane i fills slot (i & 7) of
  // row 8 * piece + (i >> 3).  Pieces of one wave are NW * 8 rows apart (a multiple of 32), so they share the swizzle.
  constexpr int PPW = 8 / NW;
  const int d_row = wave * 8 + (lane >> 3), d_slot = lane & 7;
  const int k_chunk = d_slot ^ ((d_row >> 1) & 7), v_chunk = d_slot ^ (((d_row >> 1) & 1) << 2);
  const uint32_t koff = (uint32_t)d_row * (uint32_t)p.ldk + (uint32_t)k_chunk * 8u;  // elements inside a full tile
  const uint32_t voff = (uint32_t)d_row * (uint32_t)p.ldv + (uint32_t)v_chunk * 8u;
  const uint32_t k_dst = lds_addr(Ks) + wave * 1024, v_dst = lds_addr(Vs) + wave * 1024;  // + stage * 8192 bytes
  auto issue_k = [&](int t, int stage, bool clamp) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int r0 = i * NW * 8;  // first row of this wave's i-th piece, relative to its first piece
      if (clamp) {
        int key = t * KV + d_row + r0;
        key = key > Lk - 1 ? Lk - 1 : key;
        dma16(Kb + (int64_t)key * p.ldk + k_chunk * 8, k_dst + stage * (TILE * 2) + r0 * 128);
      } else {
        dma16(Kb + ((int64_t)t * KV + r0) * p.ldk, koff * 2u, k_dst + stage * (TILE * 2) + r0 * 128);
      }
    }
  };
  auto issue_v = [&](int t, int stage, bool clamp) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int r0 = i * NW * 8;
      if (clamp) {
        int key = t * KV + d_row + r0;
        key = key > Lk - 1 ? Lk - 1 : key;
        dma16(Vb + (int64_t)key * p.ldv + v_chunk * 8, v_dst + stage * (TILE * 2) + r0 * 128);
      } else {
        dma16(Vb + ((int64_t)t * KV + r0) * p.ldv, voff * 2u, v_dst + stage * (TILE * 2) + r0 * 128);
      }
    }
  };
  auto dma_wait_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed
    __builtin_amdgcn_s_barrier();                      // ... and everybody else's; all reads of the old stages are done
    asm volatile("" ::: "memory");
  };
  // fragment addresses (elements): K row l31, chunk (2j + lh) ^ swizzle;  V row 4*lh + ((lane & 15) >> 2),
  // chunk (4 db + 2 ((lane >> 4) & 1) + ((lane & 3) >> 1)) ^ swizzle, 8-byte half (lane & 1)
  int k_lane[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) k_lane[j] = l31 * 64 + (((2 * j + lh) ^ ((l31 >> 1) & 7)) * 8);
  int v_lane[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
    v_lane[db] = (4 * lh + ((lane & 15) >> 2)) * 64 + ((4 * (db ^ ((lane >> 3) & 1)) + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) * 8) +
                 4 * (lane & 1);

  f32x16_t negm;  // -m in every register: the C operand of the first QK^T MFMA of a block (FOLD)
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  auto qk_block = [&](int stage, int kb, f32x16_t& s) {
    if (!FOLD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + stage * TILE + kb * 32 * 64 + k_lane[j]);
      s = mfma16<H16>(kf, qf[j], (FOLD && j == 0) ? negm : s);
    }
  };
  auto mask_tail = [&](int t, f32x16_t (&s)[2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int key0 = t * KV + kb * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (key0 + (r & 3) + 8 * (r >> 2) >= Lk) s[kb][r] = -1e30f;
    }
  };
  float mc = 0.f;
  auto softmax_block = [&](const f32x16_t& s, bf16x8_t (&pf)[2]) {
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pv[r] = FOLD ? __builtin_amdgcn_exp2f(s[r]) : __builtin_amdgcn_exp2f(s[r] * cs - mc);
    float sum = pv[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) sum += pv[r];
    l_run += sum;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      U4 w;
      w.x = cvt_pk<H16>(pv[jj * 8 + 0], pv[jj * 8 + 1]);
      w.y = cvt_pk<H16>(pv[jj * 8 + 2], pv[jj * 8 + 3]);
      w.z = cvt_pk<H16>(pv[jj * 8 + 4], pv[jj * 8 + 5]);
      w.w = cvt_pk<H16>(pv[jj * 8 + 6], pv[jj * 8 + 7]);
      pf[jj] = *reinterpret_cast<bf16x8_t*>(&w);
    }
  };
  auto pv_block = [&](int stage, int kb, const bf16x8_t (&pf)[2]) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const u16* vp = Vs + stage * TILE + (kb * 32 + jj * 16) * 64 + v_lane[db];
        s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)vp);
        s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vp + 8 * 64));
        s16x8_t v01 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        bf16x8_t vf = *reinterpret_cast<bf16x8_t*>(&v01);
        o[db] = mfma16<H16>(vf, pf[jj], o[db]);
      }
  };

  // prologue: K(0) .. K(AHEAD), V(0) .. V(AHEAD-1) -> LDS (clamped past the end of the sequence); tile i lives in
  // stage i % RING of its ring
  issue_k(0, 0, true);
  issue_v(0, 0, true);
  issue_k(1, 1, true);
  dma_wait_barrier();
  f32x16_t s_cur[2], s_nxt[2];
  qk_block(0, 0, s_cur[0]);  // negm is still zero here
  qk_block(0, 1, s_cur[1]);
  if (nt == 1 && (Lk % KV) != 0) mask_tail(0, s_cur);
  {
    float mx = fmaxf(s_cur[0][0], s_cur[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s_cur[0][r], s_cur[1][r]));
    m_run = fmaxf(mx, __shfl_xor(mx, 32));
    if constexpr (H16) m_run += H16_OFF / cs;  // the largest probability of the first tile is 2^-H16_OFF (see H16 above)
    mc = m_run * cs;
    if (FOLD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        negm[r] = -m_run;
        s_cur[0][r] -= m_run;
        s_cur[1][r] -= m_run;
      }
    }
  }
  __syncthreads();  // every wave has read K(0): step 0 overwrites its stage

  // One pipelined step: consumes S(t) from `sa`, produces S(t+1) into `sb`; straight-line code (no branches).
  // Before it K(t+1 .. t+AHEAD) and V(t .. t+AHEAD-1) are resident; its DMA (issue = true) brings K(t+AHEAD+1) over
  // K(t+AHEAD+1-RING) and V(t+AHEAD) over V(t+AHEAD-RING), both last read before the previous barrier.
  auto compute = [&](int t, f32x16_t (&sa)[2], f32x16_t (&sb)[2]) {
    const int kst = (t + 1) % RING, vst = t % RING;
    bf16x8_t pf[2];
    qk_block(kst, 0, sb[0]);
    softmax_block(sa[0], pf);
    pv_block(vst, 0, pf);
    qk_block(kst, 1, sb[1]);
    softmax_block(sa[1], pf);
    pv_block(vst, 1, pf);
    asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(sb[0]), "v"(sb[1]));  // every MFMA of the step is issued before what follows
  };
  auto step = [&](int t, f32x16_t (&sa)[2], f32x16_t (&sb)[2], bool clamp) {
    issue_k(t + AHEAD + 1, (t + AHEAD + 1) % RING, clamp);
    issue_v(t + AHEAD, (t + AHEAD) % RING, clamp);
    compute(t, sa, sb);
    dma_wait_barrier();
  };
  int t = 0;
  const int n_full = Lk / KV;
  // main loop: every tile it loads (up to t + 3) is a full tile, so the uniform-base addressing applies
  for (; t + 3 < n_full; t += 2) {  // two steps per trip: the S register sets swap roles instead of being copied
    step(t, s_cur, s_nxt, false);
    step(t + 1, s_nxt, s_cur, false);
  }
  for (; t < nt; ++t) {  // last tiles: clamped source rows, tail mask, no look-ahead on the final one
    if (t + 1 < nt) {
      step(t, s_cur, s_nxt, true);
      if (t + 2 == nt && (Lk % KV) != 0) mask_tail(t + 1, s_nxt);
      s_cur[0] = s_nxt[0];
      s_cur[1] = s_nxt[1];
    } else {
      bf16x8_t pf[2];
      softmax_block(s_cur[0], pf);
      pv_block(t % RING, 0, pf);
      softmax_block(s_cur[1], pf);
      pv_block(t % RING, 1, pf);
      __syncthreads();
    }
  }
}

template <bool FOLD, int NW, bool H16 = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_kernel(AttnParams p) {  // two waves per SIMD: 8-wave workgroup alone, or two 4-wave workgroups
  constexpr int SMEM_EXACT = 2 * KV * LDS_LD + 2 * KV * LDS_LDV, SMEM_RINGS = 2 * RING * TILE;  // u16 elements
  __shared__ __attribute__((aligned(16))) u16 smem[SMEM_EXACT > SMEM_RINGS ? SMEM_EXACT : SMEM_RINGS];  // >= NW * 32 * LDS_LDO
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

  bf16x8_t qf[4];
  {
    int q = q_tile0 + l31;
    if (q > L - 1) q = L - 1;
    const u16* qp = Qb + (int64_t)q * p.ldq + lh * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      U4 v = ldg16(qp + j * 16);
      qf[j] = *reinterpret_cast<bf16x8_t*>(&v);
    }
  }

  f32x16_t o[2];
  float m_run = -1e30f, l_run = 0.f;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float l_tot = -1.f;
  if (!p.exact_only) {
    kv_loop_pipelined<FOLD, NW, H16>(p, Kb, Vb, smem, smem + RING * TILE, qf, o, m_run, l_run, lane, wave, l31, lh);
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  // out of range (a later tile outgrew the first tile's max by more than 2^60) or NaN: redo with the exact loop
  if (__syncthreads_or(!(l_tot < (H16 ? H16_L_MAX : 0x1p60f)) || !(l_tot > 0x1p-100f))) {
    m_run = -1e30f;
    l_run = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    u16* Ks = smem;
    u16* Vs = smem + 2 * KV * LDS_LD;
    const u16* v_lane = Vs + (4 * (lane >> 5) + ((lane & 15) >> 2)) * LDS_LDV + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    kv_loop<true, FOLD, NW, H16>(p, Kb, Vb, Ks, Vs, v_lane, qf, o, m_run, l_run, tid, l31, lh);
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  // Both loops end with a workgroup barrier, so the K/V stages are free: O goes through a wave-private LDS tile
  // (32 rows x 144 B) and leaves as whole 128-byte rows, 4 dwordx4 stores per wave (per-lane stores at the row
  // stride touch a different row per lane: 8 dwordx2 instructions whose every lane opens its own line).
  const float inv = 1.0f / l_tot;
  u16* Os = smem + wave * (32 * LDS_LDO);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = cvt_pk<H16>(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv);
      w.y = cvt_pk<H16>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(Os + l31 * LDS_LDO + db * 32 + 8 * g + 4 * lh) = w;
    }
  // the same wave reads back what it wrote (the LDS executes a wave's accesses in order): no barrier
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = 8 * k + (lane >> 3), ch = lane & 7;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(Os + row * LDS_LDO + ch * 8);
    const int q = q_tile0 + row;
    if (q < L) *reinterpret_cast<u32x4_t*>(Ob + (int64_t)q * p.ldo + ch * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// attn64_kernel: the hand-placed form.  4 waves = 256 query rows per workgroup, ONE wave per SIMD, 64 query rows and the whole
// 512-register file per wave; the main loop is a single asm statement with its own register allocation (tools/attn64/gen.py writes
// attn64_asm.inc: register map, schedule, counted waits, static hazard check; tools/attn64/sim.py executes the stream on a numpy model of
// the workgroup -- tests/test_attn64_sim.py).  What the C++ side does: the lane-constant addresses the stream takes as operands, the
// epilogue (accumulator file -> normalise -> wave-private LDS tile -> whole 128-byte rows), and the exact running-max loop as the
// in-kernel fallback of the optimistic soft-max, run per 32-row block with the 8-wave kernel's kv_loop<SAFE>.
// Pre-scaled Q only (FOLD); Lk must be a multiple of 64 and at least three tiles (attention_launch falls back to attn_kernel otherwise).
// Row sums come from the matrix pipe: v_mfma_f32_16x16x32 of a constant ones pattern (lanes 0 / 32: row 0, lanes 17 / 49: row 1) against
// the PACKED probabilities, so the normaliser is the sum of exactly the values the PV product used.
// ------------------------------------------------------------------------------------------------
template <int BASE>
__device__ __forceinline__ void acc_read16(f32x16_t& x) {
  float t[16];
#define ATTN64_RD(i) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(t[i]) : "n"(BASE + i));
  ATTN64_RD(0) ATTN64_RD(1) ATTN64_RD(2) ATTN64_RD(3) ATTN64_RD(4) ATTN64_RD(5) ATTN64_RD(6) ATTN64_RD(7)
  ATTN64_RD(8) ATTN64_RD(9) ATTN64_RD(10) ATTN64_RD(11) ATTN64_RD(12) ATTN64_RD(13) ATTN64_RD(14) ATTN64_RD(15)
#undef ATTN64_RD
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = t[i];
}
template <int BASE>
__device__ __forceinline__ float acc_read1() {
  float t;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(t) : "n"(BASE));
  return t;
}

// 32 rows of O^T (two d-blocks of one row block) * inv -> wave-private LDS tile -> 4 row-wide stores
template <bool H16>
__device__ __forceinline__ void store_rows32(const f32x16_t (&o)[2], float inv, u16* Os, u16* Ob, int64_t ldo, int q_first, int L, int lane,
                                             int l31, int lh) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = cvt_pk<H16>(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv);
      w.y = cvt_pk<H16>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(Os + l31 * LDS_LDO + db * 32 + 8 * g + 4 * lh) = w;
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = 8 * k + (lane >> 3), ch = lane & 7;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(Os + row * LDS_LDO + ch * 8);
    const int q = q_first + row;
    if (q < L) *reinterpret_cast<u32x4_t*>(Ob + (int64_t)q * ldo + ch * 8) = v;
  }
}

template <bool H16>
__global__ __launch_bounds__(256, 1) void attn64_kernel(AttnParams p) {
  constexpr int SMEM_EXACT = 2 * KV * LDS_LD + 2 * KV * LDS_LDV, SMEM_RINGS = 2 * RING * TILE;  // u16 elements; >= 4 * 32 * LDS_LDO
  __shared__ __attribute__((aligned(16))) u16 smem[SMEM_EXACT > SMEM_RINGS ? SMEM_EXACT : SMEM_RINGS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int L = p.L, Lk = p.Lk;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = lid % p.nqt, bh = lid / p.nqt;
  const int head = bh % p.heads, batch = bh / p.heads;
  const int q_tile0 = qt * 256 + wave * 64;
  const u16* Qb = p.Q + (int64_t)batch * L * p.ldq + head * 64;
  const u16* Kb = p.K + (int64_t)batch * Lk * p.ldk + head * 64;
  const u16* Vb = p.V + (int64_t)batch * Lk * p.ldv + head * 64;
  u16* Ob = p.O + (int64_t)batch * L * p.ldo + head * 64;
  u16* Ks = smem;
  u16* Vs = smem + RING * TILE;
  {
    // operands of the stream (tools/attn64/sim.py::wave_inputs restates these): fragment byte addresses inside stage 0 of each ring,
    // DMA source offsets of this lane's two pieces per operand, the first-row pointers of the two row blocks, the ones pattern
    uint32_t kfa[4], vfa[2], dk[2], dv[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) kfa[j] = lds_addr(Ks) + 2u * (uint32_t)(l31 * 64 + (((2 * j + lh) ^ ((l31 >> 1) & 7)) * 8));
#pragma unroll
    for (int db = 0; db < 2; ++db)
      vfa[db] = lds_addr(Vs) + 2u * (uint32_t)((4 * lh + ((lane & 15) >> 2)) * 64 +
                                                ((4 * (db ^ ((lane >> 3) & 1)) + 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) * 8) + 4 * (lane & 1));
    const int d_row = wave * 8 + (lane >> 3), d_slot = lane & 7;
    const int k_chunk = d_slot ^ ((d_row >> 1) & 7), v_chunk = d_slot ^ (((d_row >> 1) & 1) << 2);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      dk[pc] = 2u * ((uint32_t)(d_row + 32 * pc) * (uint32_t)p.ldk + (uint32_t)k_chunk * 8u);
      dv[pc] = 2u * ((uint32_t)(d_row + 32 * pc) * (uint32_t)p.ldv + (uint32_t)v_chunk * 8u);
    }
    const u16* qa[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      int q = q_tile0 + qb * 32 + l31;
      q = q > L - 1 ? L - 1 : q;
      qa[qb] = Qb + (int64_t)q * p.ldq + lh * 8;
    }
    const uint32_t ones = (lane == 0 || lane == 32 || lane == 17 || lane == 49) ? (H16 ? 0x3C003C00u : 0x3F803F80u) : 0u;
    const uint32_t kstride = 128u * (uint32_t)p.ldk, vstride = 128u * (uint32_t)p.ldv;
    const uint32_t nt = (uint32_t)(Lk / KV);
    const uint32_t m0k = lds_addr(Ks) + wave * 1024, m0v = lds_addr(Vs) + wave * 1024;
#define ATTN64_OPERANDS                                                                                                                        \
  [kfa0] "v"(kfa[0]), [kfa1] "v"(kfa[1]), [kfa2] "v"(kfa[2]), [kfa3] "v"(kfa[3]), [vfa0] "v"(vfa[0]), [vfa1] "v"(vfa[1]), [dk0] "v"(dk[0]),    \
      [dk1] "v"(dk[1]), [dv0] "v"(dv[0]), [dv1] "v"(dv[1]), [qa0] "v"(qa[0]), [qa1] "v"(qa[1]), [ones] "v"(ones), [kbase] "s"(Kb),             \
      [vbase] "s"(Vb), [kstride] "s"(kstride), [vstride] "s"(vstride), [nt] "s"(nt), [m0k] "s"(m0k), [m0v] "s"(m0v)
#ifdef ATTN64_TIMING
    uint32_t ts[6];
#define ATTN64_STAMPS [ts0] "=s"(ts[0]), [ts1] "=s"(ts[1]), [ts2] "=s"(ts[2]), [ts3] "=s"(ts[3]), [ts4] "=s"(ts[4]), [ts5] "=s"(ts[5])
    if constexpr (H16) asm volatile(ATTN64_ASM_F16 : ATTN64_STAMPS : ATTN64_OPERANDS : ATTN64_CLOBBERS);
    else asm volatile(ATTN64_ASM_BF16 : ATTN64_STAMPS : ATTN64_OPERANDS : ATTN64_CLOBBERS);
    if (p.dbg && lane == 0) {
      unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 3;
#pragma unroll
      for (int i = 0; i < 3; ++i) d[i] = ((unsigned long long)ts[2 * i + 1] << 32) | ts[2 * i];
    }
#undef ATTN64_STAMPS
#else
    if constexpr (H16) asm volatile(ATTN64_ASM_F16 : : ATTN64_OPERANDS : ATTN64_CLOBBERS);
    else asm volatile(ATTN64_ASM_BF16 : : ATTN64_OPERANDS : ATTN64_CLOBBERS);
#endif
#undef ATTN64_OPERANDS
  }
  // the stream ends behind a workgroup barrier with every DMA landed: the rings are free.  O^T (db, qb) = a[16 (2 db + qb) ..],
  // row sums of block qb = a[64 + 4 qb] (rows 0..15, lanes 0..15) and a[65 + 4 qb] (rows 16..31)
  f32x16_t o[2][2];  // [qb][db]
  acc_read16<0>(o[0][0]);
  acc_read16<16>(o[1][0]);
  acc_read16<32>(o[0][1]);
  acc_read16<48>(o[1][1]);
  float l_tot[2];
  if constexpr (ATTN64_ROWSUM_VALU) {  // a schedule with VALU row sums (tools/attn64/gen.py rowsum = pk | add): each lane's half of its row
    const float a = acc_read1<64>(), b = acc_read1<68>();
    l_tot[0] = a + __shfl_xor(a, 32);
    l_tot[1] = b + __shfl_xor(b, 32);
  } else {
    const float a0 = __shfl(acc_read1<64>(), l31 & 15), a1 = __shfl(acc_read1<65>(), l31 & 15);
    const float b0 = __shfl(acc_read1<68>(), l31 & 15), b1 = __shfl(acc_read1<69>(), l31 & 15);
    l_tot[0] = l31 < 16 ? a0 : a1;
    l_tot[1] = l31 < 16 ? b0 : b1;
  }
  const float l_max = H16 ? H16_L_MAX : 0x1p60f;
  const bool bad = !(l_tot[0] < l_max) || !(l_tot[0] > 0x1p-100f) || !(l_tot[1] < l_max) || !(l_tot[1] > 0x1p-100f);
  u16* Os = smem + wave * (32 * LDS_LDO);
  if (!__syncthreads_or(bad)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) store_rows32<H16>(o[qb], 1.0f / l_tot[qb], Os, Ob, p.ldo, q_tile0 + qb * 32, L, lane, l31, lh);
    return;
  }
  // out of range (a later tile outgrew the first tile's maximum by more than the operand type holds) or NaN: both row blocks again
  // with the exact running-max loop
  u16* Vx = smem + 2 * KV * LDS_LD;
  const u16* v_lane = Vx + (4 * (lane >> 5) + ((lane & 15) >> 2)) * LDS_LDV + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll 1
  for (int qb = 0; qb < 2; ++qb) {
    bf16x8_t qf[4];
    int q = q_tile0 + qb * 32 + l31;
    q = q > L - 1 ? L - 1 : q;
    const u16* qp = Qb + (int64_t)q * p.ldq + lh * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      U4 v = ldg16(qp + j * 16);
      qf[j] = *reinterpret_cast<bf16x8_t*>(&v);
    }
    f32x16_t ox[2];
    float m_run = -1e30f, l_run = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) ox[db][r] = 0.f;
    kv_loop<true, true, 4, H16>(p, Kb, Vb, smem, Vx, v_lane, qf, ox, m_run, l_run, tid, l31, lh);  // ends with a workgroup barrier
    const float lt = l_run + __shfl_xor(l_run, 32);
    store_rows32<H16>(ox, 1.0f / lt, Os, Ob, p.ldo, q_tile0 + qb * 32, L, lane, l31, lh);
    __syncthreads();  // the staging tiles alias the exact loop's K stage
  }
}

}  // namespace

static int attention_launch(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq, int64_t ldk,
                            int64_t ldv, int64_t ldo, int batch, int heads, int Lq, int Lk, float scale, bool q_scaled,
                            bool h16 = false) {
  if (!Q || !K || !V || !O || batch <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "attention: null pointer or empty shape");
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7))
    return dm4d_set_error(DM4D_ERR_ARG, "attention: row strides must be multiples of 8 elements");
  if ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V) | ((uintptr_t)O)) & 15)
    return dm4d_set_error(DM4D_ERR_ARG, "attention: Q, K, V and O must be 16-byte aligned");
  if (ldk <= 0 || ldv <= 0 || ldk >= (1 << 24) || ldv >= (1 << 24))
    return dm4d_set_error(DM4D_ERR_ARG, "attention: K/V row stride out of range");
  static const int exact = [] { const char* e = getenv("DM4D_ATTN_EXACT"); return e ? atoi(e) : 0; }();  // tuning aid
  AttnParams p{(const u16*)Q, (const u16*)K, (const u16*)V, (u16*)O, ldq, ldk, ldv, ldo, Lq, Lk, heads, 0,
               scale * 1.4426950408889634f, exact};
#ifdef ATTN64_TIMING
  p.dbg = g_attn64_dbg;
#endif
  // 8 waves per workgroup.  The kernels are written for NW = 4 as well (two independent 128-row workgroups per CU);
  // measured in one call: +1..3 % on the 2-D L0 shapes, -3..-5 % on the 3-D ones (profiles/r01_attn_nw4_ab.log)
  // the hand-placed 4 x 64 form: pre-scaled Q, whole key tiles, at least three of them (DM4D_ATTN64=0: tuning aid, the 8-wave kernel)
  static const int use64 = [] { const char* e = getenv("DM4D_ATTN64"); return e ? atoi(e) : 1; }();
  if (use64 && q_scaled && !exact && (Lk % KV) == 0 && Lk >= 3 * KV) {
    p.nqt = (Lq + 255) / 256;
    const long nwg64 = (long)p.nqt * heads * batch;
    if (nwg64 > 0x7fffffffL) return dm4d_set_error(DM4D_ERR_ARG, "attention: grid too large");
    const dim3 grid64((unsigned)nwg64), block64(256);
    if (h16) hipLaunchKernelGGL((attn64_kernel<true>), grid64, block64, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn64_kernel<false>), grid64, block64, 0, (hipStream_t)stream, p);
    return dm4d_check_launch("attn64_kernel");
  }
  constexpr int nw = 8;
  const int rows = nw * 32;  // 32 query rows per wave (64 rows per wave measured slower)
  p.nqt = (Lq + rows - 1) / rows;
  const long nwg = (long)p.nqt * heads * batch;
  if (nwg > 0x7fffffffL) return dm4d_set_error(DM4D_ERR_ARG, "attention: grid too large");
  const dim3 grid((unsigned)nwg), block(nw * 64);
  hipStream_t st = (hipStream_t)stream;
  if (h16) hipLaunchKernelGGL((attn_kernel<true, nw, true>), grid, block, 0, st, p);  // fp16 operands: pre-scaled Q only
  else if (q_scaled) hipLaunchKernelGGL((attn_kernel<true, nw>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((attn_kernel<false, nw>), grid, block, 0, st, p);
  return dm4d_check_launch("attn_kernel");
}

extern "C" int dm4d_attention_kv_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq,
                                      int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads, int Lq, int Lk,
                                      float scale) {
  return attention_launch(stream, Q, K, V, O, ldq, ldk, ldv, ldo, batch, heads, Lq, Lk, scale, false);
}

extern "C" int dm4d_attention_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq,
                                   int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads, int L, float scale) {
  return attention_launch(stream, Q, K, V, O, ldq, ldk, ldv, ldo, batch, heads, L, L, scale, false);
}

extern "C" int dm4d_attention_qscaled_kv_bf16(void* stream, const void* Q, const void* K, const void* V, void* O,
                                              int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads,
                                              int Lq, int Lk) {
  return attention_launch(stream, Q, K, V, O, ldq, ldk, ldv, ldo, batch, heads, Lq, Lk, 1.0f, true);
}

extern "C" int dm4d_attention_qscaled_kv_f16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq,
                                             int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads, int Lq, int Lk) {
  return attention_launch(stream, Q, K, V, O, ldq, ldk, ldv, ldo, batch, heads, Lq, Lk, 1.0f, true, true);
}
