// fp8 (OCP e4m3) attention for head_dim 64 on gfx950: an OPT-IN extension (BASELINE.json configs[4], "fp8 MFMA attention").
// The reference has no fp8 path, so this entry has no reference parity target: its tolerance is its own
// (tests/opcheck.py, attn_fp8_* cases) and nothing on the default path calls it.
//
// Both contractions run on v_mfma_scale_f32_32x32x64_f8f6f4 (block scales fixed at 2^0 / 2^-QSHIFT), i.e. 4 MFMAs of 64
// cycles per 64-key tile and wave against 16 of 32 cycles for the bf16 kernel:
//   S^T[kb] (32 keys x 32 queries) = K8 rows (A: lane = key row, 32 consecutive bytes of d per half-wave)
//                                    x Q8^T     (B: lane = query, same 32 bytes of d)
//   O^T[db] (32 d x 32 queries)   += Vt8 rows (A: lane = d row, 32 consecutive KEY slots per half-wave)
//                                    x P8^T     (B: lane = query, its own 32 probabilities of the tile as 32 bytes)
// The second line needs no data movement for P: a lane's 32 scores of a tile (two 32-key blocks x 16 accumulator
// registers) become its 32 B-operand bytes in register order, and the pack kernel writes V transposed with the keys of
// each tile permuted into exactly that order (slot = 32 h + 16 kb + r  <->  key = 32 kb + (r & 3) + 8 (r >> 2) + 4 h).
//
// Softmax: running max with LAZY rescaling -- the accumulators are rescaled only when a row's max grows by more than
// 2^8 over the value in use, so every probability handed to the fp8 conversion is <= 256 < 448 (e4m3 max): the P operand
// cannot saturate by construction.  Q, K, V are converted by the pack kernels, which CLAMP to +-448 and COUNT what they had
// to clamp (`saturated`, the check VERDICT r01 item 9 asks for); Q carries scale * log2(e) * 2^QSHIFT so that the bulk of
// its values sit in e4m3's normal range, undone by the B-operand block scale of the first contraction.
#include "common.h"
#include "dm4d.h"
#include "errors.h"
#include <stdlib.h>

namespace {

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v_t __attribute__((ext_vector_type(4)));

constexpr int QSHIFT = 3;              // Q8 = fp8(q * scale * log2e * 2^QSHIFT)
constexpr float FP8_MAX = 448.0f;
constexpr float RESCALE_THR8 = 8.0f;   // log2 units: P = exp2(s - m_used) <= 2^8

struct Fp8Params {
  const uint8_t* Q8;
  const uint8_t* K8;
  const uint8_t* Vt8;
  u16* O;
  int64_t ldo;
  int L, Lk, heads, nqt, ntiles;
};

__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}

// Q / K rows: [rows][heads*64] bf16 (row stride ld) -> [rows][heads*64] fp8, x * mul, clamped; 8 elements per thread
__global__ __launch_bounds__(256) void fp8_pack_rows_kernel(const u16* X, int64_t ld, uint8_t* Y, int64_t rows, int cols, float mul,
                                                            int* saturated) {
  const int cv = cols / 8;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= rows * cv) return;
  const int64_t row = id / cv;
  const int c = (int)(id % cv) * 8;
  float v[8];
  unpack8(ldg16(X + row * ld + c), v);
  int sat = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] *= mul;
    sat += fabsf(v[e]) > FP8_MAX ? 1 : 0;
    v[e] = fminf(fmaxf(v[e], -FP8_MAX), FP8_MAX);
  }
  uint2 w;
  w.x = pack4_fp8(v[0], v[1], v[2], v[3]);
  w.y = pack4_fp8(v[4], v[5], v[6], v[7]);
  *reinterpret_cast<uint2*>(Y + row * cols + c) = w;
  if (sat && saturated) atomicAdd(saturated, sat);
}

// V [batch*Lk][heads*64] bf16 -> Vt8 [batch][head][tile][64 d][64 key slots] fp8; one workgroup per (batch, head, tile)
__global__ __launch_bounds__(256) void fp8_pack_vt_kernel(const u16* V, int64_t ldv, uint8_t* Vt8, int Lk, int heads, int ntiles,
                                                          int* saturated) {
  __shared__ u16 tile[64][72];  // [key][d], padded
  const int t = blockIdx.x % ntiles, bh = blockIdx.x / ntiles;
  const int head = bh % heads, batch = bh / heads;
  const int tid = threadIdx.x;
  const u16* Vb = V + (int64_t)batch * Lk * ldv + head * 64;
  {  // coalesced load: 64 keys x 128 B = 512 16-byte vectors, two per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, key = idx >> 3, ch = idx & 7;
      const int gk = t * 64 + key;
      U4 v = U4{0u, 0u, 0u, 0u};
      if (gk < Lk) v = ldg16(Vb + (int64_t)gk * ldv + ch * 8);
      *reinterpret_cast<U4*>(&tile[key][ch * 8]) = v;
    }
  }
  __syncthreads();
  // thread -> (d = tid >> 2, slot group g = tid & 3: slots 16 g .. 16 g + 15, i.e. h = g >> 1, kb = g & 1, r = 0..15)
  const int d = tid >> 2, g = tid & 3, h = g >> 1, kb = g & 1;
  float v[16];
  int sat = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    float x = bf2f(tile[key][d]);
    sat += fabsf(x) > FP8_MAX ? 1 : 0;
    v[r] = fminf(fmaxf(x, -FP8_MAX), FP8_MAX);
  }
  u32x4v_t w;
  w[0] = pack4_fp8(v[0], v[1], v[2], v[3]);
  w[1] = pack4_fp8(v[4], v[5], v[6], v[7]);
  w[2] = pack4_fp8(v[8], v[9], v[10], v[11]);
  w[3] = pack4_fp8(v[12], v[13], v[14], v[15]);
  uint8_t* dst = Vt8 + ((int64_t)bh * ntiles + t) * 4096 + d * 64 + g * 16;
  *reinterpret_cast<u32x4v_t*>(dst) = w;
  if (sat && saturated) atomicAdd(saturated, sat);
}

__device__ __forceinline__ void dma16_fp8(const void* ptr, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(ptr), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ uint32_t cvt_pk_bf16_u(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<uint32_t*>(&b);
}

// NW waves x 32 query rows per workgroup; per 64-key tile: K8 (64 x 64 B) and Vt8 (64 x 64 B) through a 2-stage LDS ring
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fp8_kernel(Fp8Params p) {
  constexpr int LDS_LDO = 72;
  constexpr int RING_BYTES = 2 * 2 * 4096;          // [stage][K | V][4096]
  constexpr int OUT_BYTES = NW * 32 * LDS_LDO * 2;  // O staging (after the loop)
  __shared__ __attribute__((aligned(16))) char smem[RING_BYTES > OUT_BYTES ? RING_BYTES : OUT_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = lid % p.nqt, bh = lid / p.nqt;
  const int head = bh % p.heads, batch = bh / p.heads;
  const int C = p.heads * 64;
  const int q_tile0 = qt * (NW * 32) + wave * 32;

  // Q fragment: B operand of the first contraction, 32 bytes of d per half-wave
  v8i_t qf;
  {
    int q = q_tile0 + l31;
    if (q > p.L - 1) q = p.L - 1;
    const uint8_t* qp = p.Q8 + ((int64_t)batch * p.L + q) * C + head * 64 + lh * 32;
    const u32x4v_t a = *reinterpret_cast<const u32x4v_t*>(qp), b = *reinterpret_cast<const u32x4v_t*>(qp + 16);
    qf = v8i_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  }
  const uint8_t* Kb = p.K8 + (int64_t)batch * p.Lk * C + head * 64;
  const uint8_t* Vb = p.Vt8 + (int64_t)bh * p.ntiles * 4096;

  // DMA: one 1-KiB piece per wave and tile (waves 0-3: K rows 16 w .. 16 w + 15, waves 4-7: the same rows of Vt); a
  // 64-byte row has four 16-byte chunks, LDS position (row, pos) holds chunk pos ^ ((row >> 2) & 3)  (conflict-free
  // ds_read_b128 of one chunk column over the instruction's 16-lane groups)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const int d_row = (wave & 3) * 16 + (lane >> 2), d_pos = lane & 3, d_chunk = d_pos ^ ((d_row >> 2) & 3);
  auto issue = [&](int t, int stage) {
    if (NW == 4 || wave < 4) {
      int key = t * 64 + d_row;
      key = key > p.Lk - 1 ? p.Lk - 1 : key;  // rows past the end: any valid row (their scores are masked)
      dma16_fp8(Kb + (int64_t)key * C + d_chunk * 16, lds0 + stage * 8192 + (wave & 3) * 1024);
    }
    if (NW == 4 || wave >= 4)
      dma16_fp8(Vb + (int64_t)t * 4096 + d_row * 64 + d_chunk * 16, lds0 + stage * 8192 + 4096 + (wave & 3) * 1024);
  };
  // fragment reads: row l31 (+ 32 per block), chunks 2 lh and 2 lh + 1 under the row's key
  const int fkey = (l31 >> 2) & 3;
  const int f_off0 = l31 * 64 + (((2 * lh) ^ fkey) * 16), f_off1 = l31 * 64 + (((2 * lh + 1) ^ fkey) * 16);
  auto read_frag = [&](int base) {
    const u32x4v_t a = *reinterpret_cast<const u32x4v_t*>(smem + base + f_off0);
    const u32x4v_t b = *reinterpret_cast<const u32x4v_t*>(smem + base + f_off1);
    return v8i_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  };

  f32x16_t o[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  // -m_use in every register: the C operand of the first contraction, so the scores come out of the MFMA already relative
  // to the max in use (no per-score subtraction, no accumulator zeroing per tile)
  f32x16_t negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  float l_run = 0.f;
  bool first = true;

  // One barrier per tile, tile t+1 in flight under the work on tile t.  (Issuing the MFMAs of S(t+1) ahead of the softmax
  // of S(t), as the bf16 kernel does, was measured and made this kernel 15 % slower: profiles/r02_attn_fp8.log.)
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < p.ntiles; ++t) {
    const int stage = t & 1;
    if (t + 1 < p.ntiles) issue(t + 1, stage ^ 1);
    const int kbase = stage * 8192, vbase = kbase + 4096;
    f32x16_t s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const v8i_t kf = read_frag(kbase + kb * 32 * 64);
      s[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf, negm, 0, 0, 0, 127, 0, 127 - QSHIFT);
    }
    if (t == p.ntiles - 1 && (p.Lk & 63) != 0) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= p.Lk) s[kb][r] = -1e30f;
    }
    // running max with LAZY rescaling: s is relative to the max in use; only when some row of the wave exceeds it by more
    // than 2^8 (always on the first tile) are that row's accumulators, sum and scores moved to the new max
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const bool grow = first || mx > RESCALE_THR8;
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {
      const float delta = grow ? mx : 0.f;  // new max in use = old + delta
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
      const float nm = negm[0] - delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[r] = nm;
      first = false;
    }
    v8i_t pf;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(s[kb][r]);
      float sum = pv[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) sum += pv[r];
      l_run += sum;
#pragma unroll
      for (int j = 0; j < 4; ++j) pf[kb * 4 + j] = (int)pack4_fp8(pv[4 * j], pv[4 * j + 1], pv[4 * j + 2], pv[4 * j + 3]);
    }
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const v8i_t vf = read_frag(vbase + db * 32 * 64);
      o[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, o[db], 0, 0, 0, 127, 0, 127);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  // O through a wave-private LDS tile (32 rows x 144 B), out as whole 128-byte rows (same as the bf16 kernel)
  u16* Os = reinterpret_cast<u16*>(smem) + wave * (32 * LDS_LDO);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w;
      w.x = cvt_pk_bf16_u(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv);
      w.y = cvt_pk_bf16_u(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(Os + l31 * LDS_LDO + db * 32 + 8 * g + 4 * lh) = w;
    }
  u16* Ob = p.O + (int64_t)batch * p.L * p.ldo + head * 64;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = 8 * k + (lane >> 3), ch = lane & 7;
    const u32x4v_t v = *reinterpret_cast<const u32x4v_t*>(Os + row * LDS_LDO + ch * 8);
    const int q = q_tile0 + row;
    if (q < p.L) *reinterpret_cast<u32x4v_t*>(Ob + (int64_t)q * p.ldo + ch * 8) = v;
  }
}

}  // namespace

extern "C" size_t dm4d_attention_fp8_ws_bytes(int batch, int heads, int Lq, int Lk) {
  if (batch <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0) return 0;
  const size_t C = (size_t)heads * 64, nt = ((size_t)Lk + 63) / 64;
  return (size_t)batch * Lq * C + (size_t)batch * Lk * C + (size_t)batch * heads * nt * 4096 + 256;
}

extern "C" int dm4d_attention_fp8_kv_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq,
                                          int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads, int Lq, int Lk,
                                          float scale, int q_scaled, void* ws, size_t ws_bytes, int* saturated) {
  if (!Q || !K || !V || !O || !ws || batch <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "attention_fp8: null pointer or empty shape");
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7))
    return dm4d_set_error(DM4D_ERR_ARG, "attention_fp8: row strides must be multiples of 8 elements");
  if ((((uintptr_t)Q) | ((uintptr_t)K) | ((uintptr_t)V) | ((uintptr_t)O) | ((uintptr_t)ws)) & 15)
    return dm4d_set_error(DM4D_ERR_ARG, "attention_fp8: Q, K, V, O and the workspace must be 16-byte aligned");
  if (ws_bytes < dm4d_attention_fp8_ws_bytes(batch, heads, Lq, Lk))
    return dm4d_set_error(DM4D_ERR_ARG, "attention_fp8: workspace too small (dm4d_attention_fp8_ws_bytes)");
  hipStream_t st = (hipStream_t)stream;
  const int C = heads * 64, ntiles = (Lk + 63) / 64;
  uint8_t* Q8 = (uint8_t*)ws;
  uint8_t* K8 = Q8 + (((size_t)batch * Lq * C + 15) & ~(size_t)15);
  uint8_t* Vt8 = K8 + (((size_t)batch * Lk * C + 15) & ~(size_t)15);
  const float qmul = (q_scaled ? 1.0f : scale * 1.4426950408889634f) * (float)(1 << QSHIFT);
  {
    const int64_t n = (int64_t)batch * Lq * (C / 8);
    hipLaunchKernelGGL(fp8_pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const u16*)Q, ldq, Q8,
                       (int64_t)batch * Lq, C, qmul, saturated);
    int rc = dm4d_check_launch("fp8_pack_rows_kernel(Q)");
    if (rc) return rc;
  }
  {
    const int64_t n = (int64_t)batch * Lk * (C / 8);
    hipLaunchKernelGGL(fp8_pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const u16*)K, ldk, K8,
                       (int64_t)batch * Lk, C, 1.0f, saturated);
    int rc = dm4d_check_launch("fp8_pack_rows_kernel(K)");
    if (rc) return rc;
  }
  {
    hipLaunchKernelGGL(fp8_pack_vt_kernel, dim3((unsigned)(batch * heads * ntiles)), dim3(256), 0, st, (const u16*)V, ldv, Vt8, Lk,
                       heads, ntiles, saturated);
    int rc = dm4d_check_launch("fp8_pack_vt_kernel");
    if (rc) return rc;
  }
  Fp8Params p{Q8, K8, Vt8, (u16*)O, ldo, Lq, Lk, heads, 0, ntiles};
  // 4 waves (128 query rows) per workgroup: several small workgroups per CU drift out of phase and cover each other's
  // softmax with MFMAs (+4..+10 % over 8 waves on the long sequences, profiles/r02_attn_fp8.log); DM4D_FP8_NW=8 selects 8
  static const int nw = [] { const char* e = getenv("DM4D_FP8_NW"); return (e && atoi(e) == 8) ? 8 : 4; }();
  p.nqt = (Lq + nw * 32 - 1) / (nw * 32);
  const long nwg = (long)p.nqt * heads * batch;
  if (nwg > 0x7fffffffL) return dm4d_set_error(DM4D_ERR_ARG, "attention_fp8: grid too large");
  if (nw == 4) hipLaunchKernelGGL(attn_fp8_kernel<4>, dim3((unsigned)nwg), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(attn_fp8_kernel<8>, dim3((unsigned)nwg), dim3(512), 0, st, p);
  return dm4d_check_launch("attn_fp8_kernel");
}
