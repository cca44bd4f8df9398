// Flash-style self-attention for gfx950, head_dim 64, bf16 in/out, fp32 online softmax.
//
// Serves both the per-frame (2-D) and the frame-folded view/time (3-D) attention of
// MultiviewTransformerBlock (reference attention.py:68-83): the fold "(b t) hw c -> b (t hw) c" is
// free because activations are token-major, so the kernel just sees batch = B/F, L = F*HW.
//
// Work decomposition: a workgroup = 8 waves, each wave owns 32 query rows; K/V tiles of 64 keys are staged
// register-prefetch -> LDS, double buffered, one barrier per tile, both row-major:
//   Ks[key][d]  144-byte rows  -> conflict-free ds_read_b128 A-fragments of S^T = K Q^T
//   Vs[key][d]  192-byte rows  -> the V^T A-fragments of O^T = V^T P^T come out of ds_read_b64_tr_b16 (hardware
//               transposing read of a [4 keys][16 d] block per 16-lane group; 4 consecutive rows tile the 64 banks)
// Math (v_mfma_f32_32x32x16_bf16), per 32-key block, the two blocks of a tile software-pipelined:
//   S^T = K Q^T   ("swapped" QK^T): lane (q = lane&31) holds 16 of the 32 scores of its query row,
//                  so the row max needs a single cross-half exchange and the row sum none at all;
//   O^T = V^T P^T: the P^T B-operand is exactly the packed bf16 (v_cvt_pk_bf16_f32) of the S^T
//                  accumulator registers: k-slot e of lane half h is key 4h + (e&3) + 8*(e>>2), and the
//                  two tr-reads (keys 4h..4h+3 and 8+4h..) deliver V in that same order, so there are
//                  no lane shuffles between the two MFMAs.
// Keys may outnumber queries (Lk >= Lq): frame-sharded 3-D attention runs local queries against all-gathered K/V.
// Softmax: an optimistic pass (row max taken from the first tile only) with an exact running-max pass (lazy O/l
// rescale when a row max grows by more than 2^8) as the in-kernel fallback; see the comments at kv_loop and
// kv_loop_pipelined (the software-pipelined optimistic loop that normally runs).
// The 1-D grid is remapped so that all query tiles of one (batch, head) run on one XCD and share
// its L2 copy of K/V.
#include "common.h"
#include "dm4d.h"
#include "errors.h"
#include <stdlib.h>

namespace {

struct AttnParams {
  const u16 *Q, *K, *V;
  u16* O;
  int64_t ldq, ldk, ldv, ldo;
  int L, Lk, heads, nqt;  // L = queries per (batch, head), Lk = keys (== L for plain self-attention)
  float c;         // scale * log2(e)
  int exact_only;  // tuning / debugging: skip the optimistic pass (DM4D_ATTN_EXACT=1)
};

constexpr int KV = 64;      // keys per staged tile
constexpr int LDS_LD = 72;   // K rows: 64 + 8 pad bf16 = 144 B  (conflict-free ds_read_b128 fragments)
constexpr int LDS_LDV = 96;  // V rows: 64 + 32 pad bf16 = 192 B (4 consecutive rows tile the 64 banks for the tr reads)
constexpr int NW = 8;       // waves per workgroup
constexpr float RESCALE_THR = 8.0f;  // in log2 units

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t b = __builtin_convertvector(v, bf16x2_t);  // v_cvt_pk_bf16_f32
  return *reinterpret_cast<uint32_t*>(&b);
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
template <bool SAFE>
__device__ __forceinline__ void kv_loop(const AttnParams& p, const u16* Kb, const u16* Vb, u16* Ks, u16* Vs, const u16* v_lane,
                                        const bf16x8_t (&qf)[4], f32x16_t (&o)[2], float& m_run, float& l_run, int tid,
                                        int l31, int lh) {
  const int Lk = p.Lk;
  U4 rk, rv;
  const int s_key = tid >> 3, s_c = tid & 7;
  auto load_tile = [&](int t) {
    int key = t * KV + s_key;
    if (key > Lk - 1) key = Lk - 1;
    rk = ldg16(Kb + (int64_t)key * p.ldk + s_c * 8);
    rv = ldg16(Vb + (int64_t)key * p.ldv + s_c * 8);
  };
  auto store_tile = [&](int buf) {
    *reinterpret_cast<U4*>(Ks + (buf * KV + s_key) * LDS_LD + s_c * 8) = rk;
    *reinterpret_cast<U4*>(Vs + (buf * KV + s_key) * LDS_LDV + s_c * 8) = rv;
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
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[j], s[kb], 0, 0, 0);
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
      } else if (__any((mx - m_run) * p.c > RESCALE_THR)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      }
    }
    const float mc = m_run * p.c;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float pv[16];
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(s[kb][r] * p.c - mc);
        sum += pv[r];
      }
      l_run += sum;
      bf16x8_t pf[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        U4 w;
        w.x = cvt_pk_bf16(pv[jj * 8 + 0], pv[jj * 8 + 1]);
        w.y = cvt_pk_bf16(pv[jj * 8 + 2], pv[jj * 8 + 3]);
        w.z = cvt_pk_bf16(pv[jj * 8 + 4], pv[jj * 8 + 5]);
        w.w = cvt_pk_bf16(pv[jj * 8 + 6], pv[jj * 8 + 7]);
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
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[jj], o[db], 0, 0, 0);
        }
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined form of the optimistic loop.  A wave can issue VALU instructions in the shadow of its OWN
// MFMAs (issue probe: 1 MFMA + 4 v_fma = 34 cycles, + 4 v_exp = 48, against 32 for the bare MFMA), but only if
// independent VALU work sits next to every MFMA and the MFMA's LDS fragments were requested well before it.  QK^T
// therefore runs one tile ahead of its softmax (K staged one tile ahead of V in LDS) and the loop body is straight-line
// code in which the compiler alternates MFMAs and VALU work (216 VGPRs, one workgroup per CU).  +3..7 % on every UNet
// shape; forcing the order with sched_group_barrier ({reads ; 1 MFMA ; 8 VALU} x 16) was 2 % slower than the
// compiler's own interleave.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void kv_loop_pipelined(const AttnParams& p, const u16* Kb, const u16* Vb, u16* Ks, u16* Vs,
                                                  const u16* v_lane, const bf16x8_t (&qf)[4], f32x16_t (&o)[2], float& m_run,
                                                  float& l_run, int tid, int l31, int lh) {
  const int Lk = p.Lk;
  const int nt = (Lk + KV - 1) / KV;
  const int s_key = tid >> 3, s_c = tid & 7;
  auto key_row = [&](int t) {
    int key = t * KV + s_key;
    return key > Lk - 1 ? Lk - 1 : key;
  };
  auto qk_block = [&](int buf, int kb, f32x16_t& s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (buf * KV + kb * 32 + l31) * LDS_LD + j * 16 + lh * 8);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[j], s, 0, 0, 0);
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
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = __builtin_amdgcn_exp2f(s[r] * p.c - mc);
      sum += pv[r];
    }
    l_run += sum;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      U4 w;
      w.x = cvt_pk_bf16(pv[jj * 8 + 0], pv[jj * 8 + 1]);
      w.y = cvt_pk_bf16(pv[jj * 8 + 2], pv[jj * 8 + 3]);
      w.z = cvt_pk_bf16(pv[jj * 8 + 4], pv[jj * 8 + 5]);
      w.w = cvt_pk_bf16(pv[jj * 8 + 6], pv[jj * 8 + 7]);
      pf[jj] = *reinterpret_cast<bf16x8_t*>(&w);
    }
  };
  auto pv_block = [&](int buf, int kb, const bf16x8_t (&pf)[2]) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const u16* vp = v_lane + (buf * KV + kb * 32 + jj * 16) * LDS_LDV + db * 32;
        s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)vp);
        s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vp + 8 * LDS_LDV));
        s16x8_t v01 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        bf16x8_t vf = *reinterpret_cast<bf16x8_t*>(&v01);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[jj], o[db], 0, 0, 0);
      }
  };

  {  // prologue: K(0), V(0), K(1) -> LDS
    U4 k0 = ldg16(Kb + (int64_t)key_row(0) * p.ldk + s_c * 8);
    U4 v0 = ldg16(Vb + (int64_t)key_row(0) * p.ldv + s_c * 8);
    U4 k1 = ldg16(Kb + (int64_t)key_row(1) * p.ldk + s_c * 8);  // clamped when nt == 1
    *reinterpret_cast<U4*>(Ks + (0 * KV + s_key) * LDS_LD + s_c * 8) = k0;
    *reinterpret_cast<U4*>(Vs + (0 * KV + s_key) * LDS_LDV + s_c * 8) = v0;
    *reinterpret_cast<U4*>(Ks + (1 * KV + s_key) * LDS_LD + s_c * 8) = k1;
  }
  __syncthreads();
  f32x16_t s_cur[2], s_nxt[2];
  qk_block(0, 0, s_cur[0]);
  qk_block(0, 1, s_cur[1]);
  if (nt == 1 && (Lk % KV) != 0) mask_tail(0, s_cur);
  {
    float mx = fmaxf(s_cur[0][0], s_cur[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s_cur[0][r], s_cur[1][r]));
    m_run = fmaxf(mx, __shfl_xor(mx, 32));
    mc = m_run * p.c;
  }
  __syncthreads();  // every wave has read K(0): its buffer is overwritten at the end of iteration 0

  // one pipelined step: consumes S(t) from `sa`, produces S(t+1) into `sb`; straight-line code (no branches)
  auto step = [&](int t, f32x16_t (&sa)[2], f32x16_t (&sb)[2]) {
    const int kbuf = (t + 1) & 1, vbuf = t & 1;
    const U4 rk = ldg16(Kb + (int64_t)key_row(t + 2) * p.ldk + s_c * 8);  // clamped past the end of the sequence
    const U4 rv = ldg16(Vb + (int64_t)key_row(t + 1) * p.ldv + s_c * 8);
    bf16x8_t pf[2];
    qk_block(kbuf, 0, sb[0]);
    softmax_block(sa[0], pf);
    pv_block(vbuf, 0, pf);
    qk_block(kbuf, 1, sb[1]);
    softmax_block(sa[1], pf);
    pv_block(vbuf, 1, pf);
    *reinterpret_cast<U4*>(Ks + ((t & 1) * KV + s_key) * LDS_LD + s_c * 8) = rk;         // K(t+2) over K(t)
    *reinterpret_cast<U4*>(Vs + (((t + 1) & 1) * KV + s_key) * LDS_LDV + s_c * 8) = rv;  // V(t+1) over V(t-1)
    __syncthreads();
  };
  int t = 0;
  for (; t + 3 < nt; t += 2) {  // two steps per trip: the S register sets swap roles instead of being copied
    step(t, s_cur, s_nxt);
    step(t + 1, s_nxt, s_cur);
  }
  for (; t < nt; ++t) {  // last <= 3 tiles: tail mask, no look-ahead on the final one
    if (t + 1 < nt) {
      step(t, s_cur, s_nxt);
      if (t + 2 == nt && (Lk % KV) != 0) mask_tail(t + 1, s_nxt);
      s_cur[0] = s_nxt[0];
      s_cur[1] = s_nxt[1];
    } else {
      bf16x8_t pf[2];
      softmax_block(s_cur[0], pf);
      pv_block(t & 1, 0, pf);
      softmax_block(s_cur[1], pf);
      pv_block(t & 1, 1, pf);
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(NW * 64) void attn_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) u16 smem[2 * KV * LDS_LD + 2 * KV * LDS_LDV];
  u16* Ks = smem;
  u16* Vs = smem + 2 * KV * LDS_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const u16* v_lane = Vs + (4 * (lane >> 5) + ((lane & 15) >> 2)) * LDS_LDV + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  f32x16_t o[2];
  float m_run = -1e30f, l_run = 0.f;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float l_tot = -1.f;
  if (!p.exact_only) {
    kv_loop_pipelined(p, Kb, Vb, Ks, Vs, v_lane, qf, o, m_run, l_run, tid, l31, lh);
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  // out of range (a later tile outgrew the first tile's max by more than 2^60) or NaN: redo with the exact loop
  if (__syncthreads_or(!(l_tot < 0x1p60f) || !(l_tot > 0x1p-100f))) {
    m_run = -1e30f;
    l_run = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    kv_loop<true>(p, Kb, Vb, Ks, Vs, v_lane, qf, o, m_run, l_run, tid, l31, lh);
    l_tot = l_run + __shfl_xor(l_run, 32);
  }
  const int q = q_tile0 + l31;
  const float inv = 1.0f / l_tot;
  if (q < L) {
    u16* op = Ob + (int64_t)q * p.ldo;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = cvt_pk_bf16(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv);
        w.y = cvt_pk_bf16(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + db * 32 + 8 * g + 4 * lh) = w;
      }
  }
}

}  // namespace

extern "C" int dm4d_attention_kv_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq,
                                      int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads, int Lq, int Lk,
                                      float scale) {
  if (!Q || !K || !V || !O || batch <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "attention: null pointer or empty shape");
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3))
    return dm4d_set_error(DM4D_ERR_ARG, "attention: row strides must be multiples of 8 elements");
  static const int exact = [] { const char* e = getenv("DM4D_ATTN_EXACT"); return e ? atoi(e) : 0; }();  // tuning aid
  AttnParams p{(const u16*)Q, (const u16*)K, (const u16*)V, (u16*)O, ldq, ldk, ldv, ldo, Lq, Lk, heads, 0,
               scale * 1.4426950408889634f, exact};
  const int rows = NW * 32;  // 32 query rows per wave: 128 VGPRs, 4 waves per SIMD (64 rows per wave measured slower)
  p.nqt = (Lq + rows - 1) / rows;
  const long nwg = (long)p.nqt * heads * batch;
  if (nwg > 0x7fffffffL) return dm4d_set_error(DM4D_ERR_ARG, "attention: grid too large");
  hipLaunchKernelGGL(attn_kernel, dim3((unsigned)nwg), dim3(NW * 64), 0, (hipStream_t)stream, p);
  return dm4d_check_launch("attn_kernel");
}

extern "C" int dm4d_attention_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq,
                                   int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads, int L, float scale) {
  return dm4d_attention_kv_bf16(stream, Q, K, V, O, ldq, ldk, ldv, ldo, batch, heads, L, L, scale);
}
