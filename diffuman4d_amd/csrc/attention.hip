// Flash-style self-attention for gfx950, head_dim 64, bf16 in/out, fp32 online softmax.
//
// Serves both the per-frame (2-D) and the frame-folded view/time (3-D) attention of
// MultiviewTransformerBlock (reference attention.py:68-83): the fold "(b t) hw c -> b (t hw) c" is
// free because activations are token-major, so the kernel just sees batch = B/F, L = F*HW.
//
// Work decomposition: grid = (q tiles, heads, batch); a workgroup = 4 waves, each wave owns QB
// blocks of 32 query rows; K/V tiles of 64 keys are staged (register prefetch -> LDS, double
// buffered, one barrier per tile):
//   Ks[key][d]   row-major, 144-byte rows          -> conflict-free ds_read_b128 A-fragments
//   Vt[d][key']  TRANSPOSED, 144-byte rows, key' = key with bits 2 and 3 swapped, so that the
//                8 keys a lane needs for one PV MFMA are one contiguous 16-byte read.
// Math (v_mfma_f32_32x32x16_bf16):
//   S^T = K Q^T   ("swapped" QK^T): lane (q = lane&31) holds 32 of the 64 scores of its query row,
//                  so the row max / sum need a single cross-half exchange;
//   O^T = V^T P^T: the P^T B-operand is exactly the packed bf16 of the S^T accumulator registers
//                  (the k-index permutation is shared with the Vt read, so no lane shuffles).
#include "common.h"
#include "dm4d.h"
#include "errors.h"
#include <stdlib.h>

namespace {

struct AttnParams {
  const u16 *Q, *K, *V;
  u16* O;
  int64_t ldq, ldk, ldv, ldo;
  int L;
  float c;  // scale * log2(e)
};

constexpr int KV = 64;    // keys per tile
constexpr int LDS_LD = 72;  // bf16 per LDS row (64 + 8 pad) = 144 B

template <int QB>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) u16 smem[2 * 2 * KV * LDS_LD];
  u16* Ks = smem;                    // [2][64 keys][72]
  u16* Vt = smem + 2 * KV * LDS_LD;  // [2][64 d][72]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, batch = blockIdx.z;
  const int L = p.L;
  const int q_tile0 = blockIdx.x * (4 * 32 * QB) + wave * (32 * QB);

  const u16* Qb = p.Q + (int64_t)batch * L * p.ldq + head * 64;
  const u16* Kb = p.K + (int64_t)batch * L * p.ldk + head * 64;
  const u16* Vb = p.V + (int64_t)batch * L * p.ldv + head * 64;
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

  f32x16_t o[QB][2];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -1e30f;
    l_run[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.f;
  }

  // staging registers: K two 16-B chunks per thread, V one key pair x one 8-wide d chunk
  U4 rk[2], rv[2];
  const int k_key = tid >> 3, k_c = tid & 7;  // + 32 for the second chunk
  const int v_kp = tid & 31, v_c = tid >> 5;
  auto load_tile = [&](int t) {
    const int key0 = t * KV;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = key0 + k_key + 32 * i;
      if (key > L - 1) key = L - 1;
      rk[i] = ldg16(Kb + (int64_t)key * p.ldk + k_c * 8);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = key0 + 2 * v_kp + i;
      if (key > L - 1) key = L - 1;
      rv[i] = ldg16(Vb + (int64_t)key * p.ldv + v_c * 8);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<U4*>(Ks + (buf * KV + k_key + 32 * i) * LDS_LD + k_c * 8) = rk[i];
    // transpose: dword {V[2kp][d], V[2kp+1][d]} -> Vt[d][perm(2kp)], perm swaps key bits 2 and 3
    const int k2 = 2 * v_kp;
    const int kperm = (k2 & ~12) | ((k2 & 4) << 1) | ((k2 & 8) >> 1);
    const uint32_t* a = reinterpret_cast<const uint32_t*>(&rv[0]);
    const uint32_t* b = reinterpret_cast<const uint32_t*>(&rv[1]);
    uint32_t* dst = reinterpret_cast<uint32_t*>(Vt + (buf * 64 + v_c * 8) * LDS_LD + kperm);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t lo = (a[e] & 0xffffu) | (b[e] << 16);          // d = 8c + 2e
      uint32_t hi = (a[e] >> 16) | (b[e] & 0xffff0000u);      // d = 8c + 2e + 1
      dst[(2 * e) * (LDS_LD / 2)] = lo;
      dst[(2 * e + 1) * (LDS_LD / 2)] = hi;
    }
  };

  const int nt = (L + KV - 1) / KV;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) load_tile(t + 1);

    // ---- S^T = K Q^T -------------------------------------------------------------------
    f32x16_t s[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qb][kb][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (buf * KV + kb * 32 + l31) * LDS_LD + j * 16 + lh * 8);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][j], s[qb][kb], 0, 0, 0);
      }
    }
    // mask the key tail (only the last tile can be partial)
    if (t == nt - 1 && (L % KV) != 0) {
      const int key0 = t * KV;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= L) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) s[qb][kb][r] = -1e30f;
          }
        }
    }

    // ---- online softmax, P^T fragments ---------------------------------------------------
    bf16x8_t pf[QB][2][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = s[qb][0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[qb], mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * p.c);
      m_run[qb] = m_new;
      const float mc = m_new * p.c;
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = __builtin_amdgcn_exp2f(s[qb][kb][r] * p.c - mc);
          sum += pv[r];
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          U4 w;
          w.x = pack_bf2(pv[jj * 8 + 0], pv[jj * 8 + 1]);
          w.y = pack_bf2(pv[jj * 8 + 2], pv[jj * 8 + 3]);
          w.z = pack_bf2(pv[jj * 8 + 4], pv[jj * 8 + 5]);
          w.w = pack_bf2(pv[jj * 8 + 6], pv[jj * 8 + 7]);
          pf[qb][kb][jj] = *reinterpret_cast<bf16x8_t*>(&w);
        }
      }
      l_run[qb] = l_run[qb] * alpha + sum;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
    }

    // ---- O^T += V^T P^T ------------------------------------------------------------------
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vt + (buf * 64 + db * 32 + l31) * LDS_LD + kb * 32 + jj * 16 + lh * 8);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][kb][jj], o[qb][db], 0, 0, 0);
        }

    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane holds O[q = lane&31][d = db*32 + (r&3) + 8*(r>>2) + 4*lh] ----
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int q = q_tile0 + qb * 32 + l31;
    float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
    const float inv = 1.0f / l_tot;
    if (q < L) {
      u16* op = Ob + (int64_t)q * p.ldo;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 w;
          w.x = pack_bf2(o[qb][db][4 * g + 0] * inv, o[qb][db][4 * g + 1] * inv);
          w.y = pack_bf2(o[qb][db][4 * g + 2] * inv, o[qb][db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + db * 32 + 8 * g + 4 * lh) = w;
        }
    }
  }
}

}  // namespace

extern "C" int dm4d_attention_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq,
                                   int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads, int L, float scale) {
  if (!Q || !K || !V || !O || batch <= 0 || heads <= 0 || L <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "attention: null pointer or empty shape");
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3))
    return dm4d_set_error(DM4D_ERR_ARG, "attention: row strides must be multiples of 8 elements");
  if (heads > 65535 || batch > 65535) return dm4d_set_error(DM4D_ERR_ARG, "attention: grid too large");
  AttnParams p{(const u16*)Q, (const u16*)K, (const u16*)V, (u16*)O, ldq, ldk, ldv, ldo, L,
               scale * 1.4426950408889634f};
  hipStream_t st = (hipStream_t)stream;
  // 64 query rows per wave when there is enough work to fill the chip, else 32
  const long wgs2 = (long)((L + 255) / 256) * heads * batch;
  static const int force_qb = [] { const char* e = getenv("DM4D_ATTN_QB"); return e ? atoi(e) : 0; }();  // tuning aid
  const bool use2 = force_qb ? (force_qb == 2) : (wgs2 >= 512);
  if (use2) {
    hipLaunchKernelGGL((attn_kernel<2>), dim3((L + 255) / 256, heads, batch), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL((attn_kernel<1>), dim3((L + 127) / 128, heads, batch), dim3(256), 0, st, p);
  }
  return dm4d_check_launch("attn_kernel");
}
