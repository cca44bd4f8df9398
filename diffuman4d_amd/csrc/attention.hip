// Flash-style self-attention for gfx950, head_dim 64, bf16 in/out, fp32 online softmax.
//
// Serves both the per-frame (2-D) and the frame-folded view/time (3-D) attention of
// MultiviewTransformerBlock (reference attention.py:68-83): the fold "(b t) hw c -> b (t hw) c" is
// free because activations are token-major, so the kernel just sees batch = B/F, L = F*HW.
//
// Work decomposition: a workgroup = 8 waves, each wave owns QB blocks of 32 query rows (QB = 1: 128 VGPRs,
// 4 waves per SIMD, so other waves' softmax VALU work overlaps this wave's MFMAs); K/V tiles of 64 keys are
// staged register-prefetch -> LDS, double buffered, one barrier per tile, both row-major:
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
// The O rescale is lazy: O and l are only rescaled when some row's running max grows by more than
// 2^8 (exact arithmetic, P stays <= 256 in bf16); on typical data that is the first tiles only.
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
  float c;  // scale * log2(e)
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

template <int QB, bool MSUM>
__global__ __launch_bounds__(NW * 64) void attn_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) u16 smem[2 * KV * LDS_LD + 2 * KV * LDS_LDV];
  u16* Ks = smem;                    // [2][64 keys][72]
  u16* Vs = smem + 2 * KV * LDS_LD;  // [2][64 keys][96]  row-major, transposed on the way OUT (ds_read_b64_tr_b16)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int L = p.L, Lk = p.Lk;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = lid % p.nqt, bh = lid / p.nqt;
  const int head = bh % p.heads, batch = bh / p.heads;
  const int q_tile0 = qt * (NW * 32 * QB) + wave * (32 * QB);

  const u16* Qb = p.Q + (int64_t)batch * L * p.ldq + head * 64;
  const u16* Kb = p.K + (int64_t)batch * Lk * p.ldk + head * 64;
  const u16* Vb = p.V + (int64_t)batch * Lk * p.ldv + head * 64;
  u16* Ob = p.O + (int64_t)batch * L * p.ldo + head * 64;

  // Q fragments (B operand of S^T = K Q^T): lane holds Q[q][j*16 + lh*8 .. +7]
  bf16x8_t qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int q = q_tile0 + qb * 32 + l31;
    if (q > L - 1) q = L - 1;
    const u16* qp = Qb + (int64_t)q * p.ldq + lh * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      U4 v = ldg16(qp + j * 16);
      qf[qb][j] = *reinterpret_cast<bf16x8_t*>(&v);
    }
  }

  // per-lane base of the V tr-reads: 16-lane group g -> d0 = 16*(g&1), key offset 4*(g>>1); lane i -> row i>>2, chunk i&3
  const u16* v_lane = Vs + (4 * (lane >> 5) + ((lane & 15) >> 2)) * LDS_LDV + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  f32x16_t o[QB][2];
  float m_run[QB], l_run[QB];
  // MSUM: row sums come out of the matrix pipe (P^T times a block of ones) instead of 16 VALU adds per block --
  // the kernel is VALU-bound on the softmax, the MFMA pipe has slack
  f32x16_t lacc[QB];
  const U4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const bf16x8_t ones = *reinterpret_cast<const bf16x8_t*>(&ones_u);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -1e30f;
    l_run[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[qb][r] = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
  }

  // staging: every thread moves one 16-byte chunk of K and one of V per tile
  U4 rk, rv;
  const int s_key = tid >> 3, s_c = tid & 7;  // key 0..63, d-chunk 0..7
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
    const bool tail = (t == nt - 1) && (Lk % KV) != 0;

    // ---- S^T = K Q^T for both 32-key blocks: two independent accumulator chains, interleaved, so the
    //      matrix pipe stays busy while block 0's softmax runs on the VALU (software pipelining in-wave)
    f32x16_t s[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qb][kb][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (buf * KV + kb * 32 + l31) * LDS_LD + j * 16 + lh * 8);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][j], s[qb][kb], 0, 0, 0);
      }
    }
    if (tail) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int key0 = t * KV + kb * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (key0 + (r & 3) + 8 * (r >> 2) >= Lk) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) s[qb][kb][r] = -1e30f;
          }
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // ---- online softmax (lazy rescale), P^T fragments ---------------------------------------
      bf16x8_t pf[QB][2];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float mx = fmaxf(s[qb][kb][0], s[qb][kb][1]);
#pragma unroll
        for (int r = 2; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__any((mx - m_run[qb]) * p.c > RESCALE_THR)) {
          const float m_new = fmaxf(m_run[qb], mx);
          const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * p.c);
          m_run[qb] = m_new;
          l_run[qb] *= alpha;
          if (MSUM) {
#pragma unroll
            for (int r = 0; r < 16; ++r) lacc[qb][r] *= alpha;
          }
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
        }
        const float mc = m_run[qb] * p.c;
        float pv[16];
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = __builtin_amdgcn_exp2f(s[qb][kb][r] * p.c - mc);
          if (!MSUM) sum += pv[r];
        }
        l_run[qb] += sum;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          U4 w;
          w.x = cvt_pk_bf16(pv[jj * 8 + 0], pv[jj * 8 + 1]);
          w.y = cvt_pk_bf16(pv[jj * 8 + 2], pv[jj * 8 + 3]);
          w.z = cvt_pk_bf16(pv[jj * 8 + 4], pv[jj * 8 + 5]);
          w.w = cvt_pk_bf16(pv[jj * 8 + 6], pv[jj * 8 + 7]);
          pf[qb][jj] = *reinterpret_cast<bf16x8_t*>(&w);
          if (MSUM) lacc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[qb][jj], lacc[qb], 0, 0, 0);
        }
      }
      // ---- O^T += V^T P^T ------------------------------------------------------------------------
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          // A operand = V^T[d][8 keys in P's k-slot order]: two hardware-transposing reads of a [4 keys][16 d] block
          // each (lane i of a 16-lane group supplies row i>>2, d-chunk i&3 and receives column i)
          const u16* vp = v_lane + (buf * KV + kb * 32 + jj * 16) * LDS_LDV + db * 32;
          s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)vp);
          s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vp + 8 * LDS_LDV));
          s16x8_t v01 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
          bf16x8_t vf = *reinterpret_cast<bf16x8_t*>(&v01);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][jj], o[qb][db], 0, 0, 0);
        }
    }

    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[q = lane&31][d = db*32 + (r&3) + 8*(r>>2) + 4*lh] ----
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q_tile0 + qb * 32 + l31;
    const float l_tot = MSUM ? lacc[qb][0] : l_run[qb] + __shfl_xor(l_run[qb], 32);
    const float inv = 1.0f / l_tot;
    if (q < L) {
      u16* op = Ob + (int64_t)q * p.ldo;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 w;
          w.x = cvt_pk_bf16(o[qb][db][4 * g + 0] * inv, o[qb][db][4 * g + 1] * inv);
          w.y = cvt_pk_bf16(o[qb][db][4 * g + 2] * inv, o[qb][db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + db * 32 + 8 * g + 4 * lh) = w;
        }
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
  AttnParams p{(const u16*)Q, (const u16*)K, (const u16*)V, (u16*)O, ldq, ldk, ldv, ldo, Lq, Lk, heads, 0,
               scale * 1.4426950408889634f};
  hipStream_t st = (hipStream_t)stream;
  static const int force_qb = [] { const char* e = getenv("DM4D_ATTN_QB"); return e ? atoi(e) : 0; }();  // tuning aid
  // 32 query rows per wave (4 waves/SIMD) measured faster than 64 (2 waves/SIMD) on every UNet shape
  // (profiles/r01_*); the 64-row variant is kept for tuning (DM4D_ATTN_QB=2)
  const bool use2 = force_qb == 2;
  const int rows = NW * 32 * (use2 ? 2 : 1);
  p.nqt = (Lq + rows - 1) / rows;
  const long nwg = (long)p.nqt * heads * batch;
  if (nwg > 0x7fffffffL) return dm4d_set_error(DM4D_ERR_ARG, "attention: grid too large");
  static const int msum = [] { const char* e = getenv("DM4D_ATTN_MSUM"); return e ? atoi(e) : 0; }();  // tuning aid
  if (use2) {
    hipLaunchKernelGGL((attn_kernel<2, false>), dim3((unsigned)nwg), dim3(NW * 64), 0, st, p);
  } else if (msum) {
    hipLaunchKernelGGL((attn_kernel<1, true>), dim3((unsigned)nwg), dim3(NW * 64), 0, st, p);
  } else {
    hipLaunchKernelGGL((attn_kernel<1, false>), dim3((unsigned)nwg), dim3(NW * 64), 0, st, p);
  }
  return dm4d_check_launch("attn_kernel");
}

extern "C" int dm4d_attention_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq,
                                   int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads, int L, float scale) {
  return dm4d_attention_kv_bf16(stream, Q, K, V, O, ldq, ldk, ldv, ldo, batch, heads, L, L, scale);
}
