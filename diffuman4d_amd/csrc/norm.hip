// HBM-bound normalisation kernels for gfx950: GroupNorm(+SiLU) over NHWC (two-source channel
// concat fused in), LayerNorm, and a row softmax (VAE mid-block attention).  All statistics fp32,
// all global traffic in 16-byte bf16x8 vectors.
#include "common.h"
#include "dm4d.h"
#include "errors.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAXC = 4096;

__host__ __device__ inline int gn_nchunk(int B, int HW) {
  int n = (2048 + B - 1) / B;
  int cap = (HW + 15) / 16;
  if (n > cap) n = cap;
  if (n < 1) n = 1;
  return n;
}

// IO (template parameter of the GroupNorm / LayerNorm kernels): what a tensor element is on the way in and out.
//   0  bf16 in, bf16 out, bf16 gamma / beta                    (fast precision)
//   2  fp32 in, ONE fp16 plane out, fp16 gamma / beta          (precision "fp16": dm4d_groupnorm_nhwc_f32_f16 / dm4d_layernorm_f32_f16)
//   3  fp16 in, ONE fp16 plane out, fp16 gamma / beta          (precision "fp16": dm4d_groupnorm_nhwc_f16_f16, GroupNorm only)
// A "vector" is eight channels either way: 16 bytes of bf16, or 32 bytes of fp32 in and 16 bytes of fp16 out.
template <int IO>
struct NormIO;
template <>
struct NormIO<0> {
  typedef u16 in_t;
  typedef U4 raw_t;
  static __device__ __forceinline__ raw_t ldraw(const u16* p) { return ldg16(p); }
  static __device__ __forceinline__ void unpack(const raw_t& r, float* v) { unpack8(r, v); }
  static __device__ __forceinline__ float ld1(const u16* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st8(u16* p, const float* v) { stg16(p, pack8(v)); }
  static __device__ __forceinline__ void par8(const u16* p, float* v) { unpack8(ldg16(p), v); }
};
template <>
struct NormIO<2> {
  typedef float in_t;
  struct raw_t {
    f32x4_t a, b;
  };
  static __device__ __forceinline__ raw_t ldraw(const float* p) {
    raw_t r;
    r.a = *reinterpret_cast<const f32x4_t*>(p);
    r.b = *reinterpret_cast<const f32x4_t*>(p + 4);
    return r;
  }
  static __device__ __forceinline__ void unpack(const raw_t& r, float* v) {
    v[0] = r.a[0]; v[1] = r.a[1]; v[2] = r.a[2]; v[3] = r.a[3];
    v[4] = r.b[0]; v[5] = r.b[1]; v[6] = r.b[2]; v[7] = r.b[3];
  }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st8(u16* p, const float* v) { stg16(p, pack8h_sat(v)); }
  static __device__ __forceinline__ void par8(const u16* p, float* v) { unpack8h(ldg16(p), v); }
};

// IO = 3: fp16 in (an activation a convolution left in fp16: conv1 of a resnet, whose only reader is norm2), ONE fp16 plane out
template <>
struct NormIO<3> {
  typedef u16 in_t;
  typedef U4 raw_t;
  static __device__ __forceinline__ raw_t ldraw(const u16* p) { return ldg16(p); }
  static __device__ __forceinline__ void unpack(const raw_t& r, float* v) { unpack8h(r, v); }
  static __device__ __forceinline__ float ld1(const u16* p) { return h2f(*p); }
  static __device__ __forceinline__ void st8(u16* p, const float* v) { stg16(p, pack8h_sat(v)); }
  static __device__ __forceinline__ void par8(const u16* p, float* v) { unpack8h(ldg16(p), v); }
};

struct GNParams {
  const void* X1;  // NormIO<IO>::in_t
  const void* X2;
  int C1, C2, B, HW, groups, nchunk;
  float eps;
  const u16* gamma;
  const u16* beta;
  u16* Y;
  int silu;
  float* ws;  // [B][nchunk][groups][2]
  // IO = 2 only: a second output, the UN-normalised input rounded to fp16 ([B*HW, C], the channel concat of X1 | X2) -- the operand of a
  // resnet's 1x1 shortcut convolution, which reads the same tensor as its first GroupNorm (unet_multiview_blocks.py:667, resnet.py)
  u16* Yraw;
};

// Shifted statistics.  Sum / sum of squares are taken of (x - s_g), s_g = the sample's value at pixel 0 in the first channel
// of group g (any value near the group's data would do; this one costs one 2-byte load): mean = s_g + S / n and
// var = Q / n - (S / n)^2 then cancel at the scale of the group's SPREAD, not of its mean -- with raw sums a group whose mean
// is 200 standard deviations loses 15 of fp32's 24 bits (1e-2 relative error on the output; 3e-4 at 50).  All channels of a
// group share the shift, so partial sums of different threads, chunks and kernels add up exactly as before.
template <int IO = 0>
__device__ __forceinline__ float gn_shift(const GNParams& p, int b, int g, int gs) {
  typedef typename NormIO<IO>::in_t in_t;
  const int c = g * gs;  // first channel of the group, in the concatenated channel axis
  return c < p.C1 ? NormIO<IO>::ld1(static_cast<const in_t*>(p.X1) + (int64_t)b * p.HW * p.C1 + c)
                  : NormIO<IO>::ld1(static_cast<const in_t*>(p.X2) + (int64_t)b * p.HW * p.C2 + (c - p.C1));
}
// the shifts of the eight channels of vector cv: groups of eight or more channels put at most two groups in a vector
template <int IO = 0>
__device__ __forceinline__ void gn_shift8(const GNParams& p, int b, int cv, int gs, float (&sh)[8]) {
  const int g0 = (cv * 8) / gs, g1 = (cv * 8 + 7) / gs;
  if (g1 - g0 <= 1) {
    const float s0 = gn_shift<IO>(p, b, g0, gs), s1 = g1 != g0 ? gn_shift<IO>(p, b, g1, gs) : s0;
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[e] = (cv * 8 + e) / gs == g0 ? s0 : s1;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[e] = gn_shift<IO>(p, b, (cv * 8 + e) / gs, gs);
  }
}

// pass 1: per (batch, pixel chunk) partial sum / sum of squares of every group, fixed summation order
template <int IO = 0>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(GNParams p) {
  typedef NormIO<IO> io;
  typedef typename io::in_t in_t;
  typedef typename io::raw_t raw_t;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [PPB][C][2]
  const int C = p.C1 + p.C2, CV = C / 8, CV1 = p.C1 / 8;
  const int b = blockIdx.x / p.nchunk, chunk = blockIdx.x % p.nchunk;
  const int per = (p.HW + p.nchunk - 1) / p.nchunk;
  const int p0 = chunk * per, p1 = min(p0 + per, p.HW);
  const int PPB = CV >= GN_THREADS ? 1 : GN_THREADS / CV;  // pixels processed per pass
  const int tid = threadIdx.x;

  for (int slot = tid; slot < CV * PPB; slot += GN_THREADS) {
    const int cv = slot % CV, prow = slot / CV;
    float s[8], q[8], sh[8];
    gn_shift8<IO>(p, b, cv, C / p.groups, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    const in_t* src;
    int ld, c;
    if (cv < CV1) {
      src = static_cast<const in_t*>(p.X1) + (int64_t)b * p.HW * p.C1;
      ld = p.C1;
      c = cv * 8;
    } else {
      src = static_cast<const in_t*>(p.X2) + (int64_t)b * p.HW * p.C2;
      ld = p.C2;
      c = (cv - CV1) * 8;
    }
    // two loads in flight per thread (a chunk is only a few passes long: one load per pass leaves the kernel waiting on a
    // memory round trip per pass; four cost 92 registers and three of the eight resident workgroups); the sums still run
    // pixel by pixel in ascending order
    auto add = [&](const raw_t& r) {
      float v[8];
      io::unpack(r, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[e] - sh[e];
        s[e] += d;
        q[e] += d * d;
      }
    };
    int px = p0 + prow;
#pragma nounroll
    for (; px + PPB < p1; px += 2 * PPB) {
      const in_t* a = src + (int64_t)px * ld + c;
      const raw_t r0 = io::ldraw(a), r1 = io::ldraw(a + (int64_t)PPB * ld);
      add(r0);
      add(r1);
    }
    for (; px < p1; px += PPB) add(io::ldraw(src + (int64_t)px * ld + c));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sm[(prow * C + cv * 8 + e) * 2 + 0] = s[e];
      sm[(prow * C + cv * 8 + e) * 2 + 1] = q[e];
    }
  }
  __syncthreads();
  const int gs = C / p.groups;
  for (int g = tid; g < p.groups; g += GN_THREADS) {
    float s = 0.f, q = 0.f;
    for (int r = 0; r < PPB; ++r)
      for (int c = g * gs; c < (g + 1) * gs; ++c) {
        s += sm[(r * C + c) * 2 + 0];
        q += sm[(r * C + c) * 2 + 1];
      }
    float* w = p.ws + (((int64_t)b * p.nchunk + chunk) * p.groups + g) * 2;
    w[0] = s;
    w[1] = q;
  }
}

// pass 2: y = (x - mean) * rstd * gamma + beta, optional SiLU; writes the concatenated tensor
template <int IO = 0>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(GNParams p) {
  typedef NormIO<IO> io;
  typedef typename io::in_t in_t;
  typedef typename io::raw_t raw_t;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // reduction scratch, then mean[g], rstd[g]
  const int C = p.C1 + p.C2, CV = C / 8, CV1 = p.C1 / 8;
  float* sc = sm;  // reduction scratch
  float* mean = sm + (2 * C > 2 * GN_THREADS ? 2 * C : 2 * GN_THREADS);  // first part: reduction scratch
  float* rstd = mean + p.groups;
  const int b = blockIdx.x / p.nchunk, chunk = blockIdx.x % p.nchunk;
  const int per = (p.HW + p.nchunk - 1) / p.nchunk;
  const int p0 = chunk * per, p1 = min(p0 + per, p.HW);
  const int tid = threadIdx.x;
  const int gs = C / p.groups;
  // reduce the per-chunk partials: all 256 threads fetch (fixed association order => deterministic), then one
  // thread per group combines the GN_THREADS/groups slices -- the serial 64-deep loop this replaces was the
  // latency floor of the kernel
  {
    const int slices = GN_THREADS / p.groups > 0 ? GN_THREADS / p.groups : 1;
    const int g = tid % p.groups, sl = tid / p.groups;
    float s = 0.f, q = 0.f;
    if (sl < slices) {
      for (int k = sl; k < p.nchunk; k += slices) {
        const float* w = p.ws + (((int64_t)b * p.nchunk + k) * p.groups + g) * 2;
        s += w[0];
        q += w[1];
      }
      sc[(sl * p.groups + g) * 2 + 0] = s;  // 2*slices*groups <= 2*GN_THREADS floats
      sc[(sl * p.groups + g) * 2 + 1] = q;
    }
    __syncthreads();
  }
  for (int g = tid; g < p.groups; g += GN_THREADS) {
    const int slices = GN_THREADS / p.groups > 0 ? GN_THREADS / p.groups : 1;
    float s = 0.f, q = 0.f;
    for (int sl = 0; sl < slices; ++sl) {
      s += sc[(sl * p.groups + g) * 2 + 0];
      q += sc[(sl * p.groups + g) * 2 + 1];
    }
    const float n = (float)gs * (float)p.HW;
    const float dm = s / n;  // mean of (x - shift)
    float var = q / n - dm * dm;
    var = var < 0.f ? 0.f : var;
    mean[g] = gn_shift<IO>(p, b, g, gs) + dm;
    rstd[g] = rsqrtf(var + p.eps);
  }
  __syncthreads();
  // A thread keeps one 8-channel vector (its scale / shift in registers) and walks the chunk's pixels: no index
  // division and no LDS traffic in the streaming loop; 256 threads cover GN_THREADS / CV consecutive pixels per pass.
  const int PPB = CV >= GN_THREADS ? 1 : GN_THREADS / CV;
  for (int slot = tid; slot < CV * PPB; slot += GN_THREADS) {
    const int cv = slot % CV, prow = slot / CV;
    float a[8], sft[8];
    {
      float gm[8], bt[8];
      io::par8(p.gamma + cv * 8, gm);
      io::par8(p.beta + cv * 8, bt);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (cv * 8 + e) / gs;
        a[e] = gm[e] * rstd[g];
        sft[e] = bt[e] - mean[g] * a[e];
      }
    }
    const in_t* src;
    int ld;
    if (cv < CV1) {
      src = static_cast<const in_t*>(p.X1) + (int64_t)b * p.HW * p.C1 + cv * 8;
      ld = p.C1;
    } else {
      src = static_cast<const in_t*>(p.X2) + (int64_t)b * p.HW * p.C2 + (cv - CV1) * 8;
      ld = p.C2;
    }
    u16* dst = p.Y + (int64_t)b * p.HW * C + cv * 8;
    u16* dst_raw = (IO == 2 && p.Yraw) ? p.Yraw + (int64_t)b * p.HW * C + cv * 8 : nullptr;
    auto put = [&](const raw_t& r, int px) {
      float v[8];
      io::unpack(r, v);
      if constexpr (IO == 2) {
        if (dst_raw) io::st8(dst_raw + (int64_t)px * C, v);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = v[e] * a[e] + sft[e];
        v[e] = p.silu ? silu_f(y) : y;
      }
      io::st8(dst + (int64_t)px * C, v);
    };
    int px = p0 + prow;
#pragma nounroll
    for (; px + 3 * PPB < p1; px += 4 * PPB) {  // four loads in flight per thread, as in the statistics pass
      const in_t* s0 = src + (int64_t)px * ld;
      const raw_t r0 = io::ldraw(s0), r1 = io::ldraw(s0 + (int64_t)PPB * ld), r2 = io::ldraw(s0 + (int64_t)2 * PPB * ld), r3 = io::ldraw(s0 + (int64_t)3 * PPB * ld);
      put(r0, px);
      put(r1, px + PPB);
      put(r2, px + 2 * PPB);
      put(r3, px + 3 * PPB);
    }
    for (; px < p1; px += PPB) put(io::ldraw(src + (int64_t)px * ld), px);
  }
}

// Single-launch GroupNorm for feature maps small enough to stay in registers between the statistics and the apply step
// (levels 1-3 of the UNet at 72x40 latents): one workgroup owns (sample, slab of `gps` whole groups), every thread
// keeps one 16-byte channel vector of up to NV pixels in VGPRs, so X is read from HBM exactly once and the second
// launch (with its dependent-launch gap) disappears.  Same fp32 sum / sum-of-squares formulas as the two-pass kernels;
// the summation order is fixed by (HW, C, groups) only, never by the batch.
constexpr int GN_RES_MAXGPS = 8;

template <int NV, int IO = 0>
__global__ __launch_bounds__(GN_THREADS) void gn_resident_kernel(GNParams p, int gps, int SV, int PPB, int NS, int xcd_map) {
  typedef NormIO<IO> io;
  typedef typename io::in_t in_t;
  typedef typename io::raw_t raw_t;
  __shared__ float sm[2 * 8 * GN_THREADS];  // [PPB][SC][2]
  __shared__ float sm2[2 * GN_THREADS];     // [PARTS][SC][2]
  __shared__ float stat[2 * GN_RES_MAXGPS];
  const int C = p.C1 + p.C2, CV1 = p.C1 / 8, gs = C / p.groups, SC = SV * 8;
  int b, slab;
  if (xcd_map) {  // workgroup i runs on XCD i % 8: keep all slabs of a sample on one XCD so that the 128-byte lines
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;  // two slabs share are fetched into ONE L2
    b = x + 8 * (j / NS);
    slab = j % NS;
  } else {
    b = blockIdx.x / NS;
    slab = blockIdx.x % NS;
  }
  const int tid = threadIdx.x, sv = tid % SV, prow = tid / SV;
  const bool active = prow < PPB;
  const int cv = slab * gps * gs / 8 + sv;  // vector index in the concatenated channel axis
  const in_t* src;
  int ld;
  if (cv < CV1) {
    src = static_cast<const in_t*>(p.X1) + (int64_t)b * p.HW * p.C1 + cv * 8;
    ld = p.C1;
  } else {
    src = static_cast<const in_t*>(p.X2) + (int64_t)b * p.HW * p.C2 + (cv - CV1) * 8;
    ld = p.C2;
  }
  raw_t r[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int px = prow + k * PPB;
    if (active && px < p.HW) r[k] = io::ldraw(src + (int64_t)px * ld);
  }
  {
    float s[8], q[8], sh[8];
    gn_shift8<IO>(p, b, cv, gs, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int px = prow + k * PPB;
      if (active && px < p.HW) {
        float v[8];
        io::unpack(r[k], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[e] - sh[e];
          s[e] += d;
          q[e] += d * d;
        }
      }
    }
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sm[(prow * SC + sv * 8 + e) * 2 + 0] = s[e];
        sm[(prow * SC + sv * 8 + e) * 2 + 1] = q[e];
      }
    }
  }
  __syncthreads();
  {
    const int PARTS = GN_THREADS / SC, c = tid % SC, part = tid / SC;
    if (part < PARTS) {
      float s = 0.f, q = 0.f;
      for (int rr = part; rr < PPB; rr += PARTS) {
        s += sm[(rr * SC + c) * 2 + 0];
        q += sm[(rr * SC + c) * 2 + 1];
      }
      sm2[(part * SC + c) * 2 + 0] = s;
      sm2[(part * SC + c) * 2 + 1] = q;
    }
    __syncthreads();
    // one WAVE per group: the PARTS x gs partials of a group are summed 64 at a time and folded with a butterfly.  (This
    // was one THREAD per group walking up to 6 x 80 LDS entries serially -- ~10 us of latency in a kernel that moves a few
    // KB at the 18x10 and 9x5 levels.)  The order depends on (HW, C, groups) only, never on the batch.
    const int wv = tid >> 6, ln = tid & 63;
    for (int g = wv; g < gps; g += GN_THREADS / 64) {
      float s = 0.f, q = 0.f;
      for (int e = ln; e < PARTS * gs; e += 64) {
        const int pt = e / gs, cc = g * gs + (e - pt * gs);
        s += sm2[(pt * SC + cc) * 2 + 0];
        q += sm2[(pt * SC + cc) * 2 + 1];
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
      }
      if (ln == 0) {
        const float n = (float)gs * (float)p.HW;
        const float dm = s / n;  // mean of (x - shift)
        float var = q / n - dm * dm;
        var = var < 0.f ? 0.f : var;
        stat[g * 2 + 0] = gn_shift<IO>(p, b, slab * gps + g, gs) + dm;
        stat[g * 2 + 1] = rsqrtf(var + p.eps);
      }
    }
  }
  __syncthreads();
  if (!active) return;
  float a[8], sft[8];
  {
    float gm[8], bt[8];
    io::par8(p.gamma + cv * 8, gm);
    io::par8(p.beta + cv * 8, bt);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (sv * 8 + e) / gs;
      a[e] = gm[e] * stat[g * 2 + 1];
      sft[e] = bt[e] - stat[g * 2 + 0] * a[e];
    }
  }
  u16* dst = p.Y + (int64_t)b * p.HW * C + cv * 8;
  u16* dst_raw = (IO == 2 && p.Yraw) ? p.Yraw + (int64_t)b * p.HW * C + cv * 8 : nullptr;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int px = prow + k * PPB;
    if (px < p.HW) {
      float v[8];
      io::unpack(r[k], v);
      if constexpr (IO == 2) {
        if (dst_raw) io::st8(dst_raw + (int64_t)px * C, v);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = v[e] * a[e] + sft[e];
        v[e] = p.silu ? silu_f(y) : y;
      }
      io::st8(dst + (int64_t)px * C, v);
    }
  }
}

int g_gn_resident = 1;  // tuning hook (dm4d_tune_set_groupnorm_resident)

// slab geometry of the single-launch kernel, or false when the map does not fit in registers (two-pass path then)
bool gn_resident_plan(int C, int HW, int groups, int* gps, int* SV, int* PPB, int* NV) {
  const int gs = C / groups;
  for (int g = 1; g <= GN_RES_MAXGPS && g <= groups; g *= 2) {
    if (groups % g || (g * gs) % 8 || g * gs * 2 < 64) continue;
    const int sv = g * gs / 8;
    if (sv > 32) return false;
    const int ppb = GN_THREADS / sv, nv = (HW + ppb - 1) / ppb;
    if (nv > 16) return false;
    *gps = g, *SV = sv, *PPB = ppb, *NV = nv;
    return true;
  }
  return false;
}

// LayerNorm: one wave per row, row kept in registers (two-pass mean / variance)
template <int NV, int IO = 0>  // eight-channel vectors per lane
__global__ __launch_bounds__(256) void ln_kernel(const typename NormIO<IO>::in_t* X, int64_t ldx, const u16* gamma, const u16* beta, u16* Y,
                                                 int64_t ldy, int M, int C, float eps) {
  typedef NormIO<IO> io;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const int CV = C / 8;
  float v[NV][8];
  bool on[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int cv = lane + 64 * i;
    on[i] = cv < CV;
    if (on[i]) {
      io::unpack(io::ldraw(X + (int64_t)row * ldx + cv * 8), v[i]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  float mu, rs;
  ln_row_stats<NV>(v, on, C, eps, mu, rs);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int cv = lane + 64 * i;
    if (on[i]) {
      float g[8], bt[8], y[8];
      io::par8(gamma + cv * 8, g);
      io::par8(beta + cv * 8, bt);
      ln_row_apply(v[i], mu, rs, g, bt, y);
      io::st8(Y + (int64_t)row * ldy + cv * 8, y);
    }
  }
}

// row softmax over N columns: P = softmax(S * scale); one workgroup per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const u16* S, int64_t lds, u16* P, int64_t ldp, int N,
                                                           float scale) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u16* s = S + (int64_t)row * lds;
  u16* o = P + (int64_t)row * ldp;
  float mx = -1e30f;
  for (int i = tid; i < N; i += 256) mx = fmaxf(mx, bf2f(s[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  const float c = scale * 1.4426950408889634f;
  for (int i = tid; i < N; i += 256) sum += __builtin_amdgcn_exp2f((bf2f(s[i]) - mx) * c);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = red[4] + red[5] + red[6] + red[7];
  const float inv = 1.0f / sum;
  for (int i = tid; i < N; i += 256) o[i] = f2bf(__builtin_amdgcn_exp2f((bf2f(s[i]) - mx) * c) * inv);
}

// the same for fp32 logits (VAE mid-block attention, d = 512: SDPA keeps its logits in fp32; rounding them to bf16 first
// costs up to 2^-9 * |logit| in the exponent).  Rows are read as float4 (16-byte aligned rows: lds % 4 == 0) with a scalar
// tail of N % 4 columns (threads 0..2), so a row may be LONGER than N: the caller pads the key axis to the GEMM's K
// granularity and the columns behind N are neither read nor written.  The second and third pass of a row hit L2.
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(const float* S, int64_t lds, u16* P, int64_t ldp, int N,
                                                               float scale) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* srow = S + (int64_t)row * lds;
  u16* prow = P + (int64_t)row * ldp;
  const f32x4_t* s = reinterpret_cast<const f32x4_t*>(srow);
  uint2* o = reinterpret_cast<uint2*>(prow);
  const int n4 = N >> 2;
  const bool tail = tid < (N & 3);  // this thread also owns column 4 n4 + tid
  const float st = tail ? srow[4 * n4 + tid] : 0.f;
  float mx = tail ? st : -3.0e38f;
  for (int i = tid; i < n4; i += 256) {
    const f32x4_t v = s[i];
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float c = scale * 1.4426950408889634f;
  const float mc = mx * c;
  const float et = tail ? __builtin_amdgcn_exp2f(fmaf(st, c, -mc)) : 0.f;
  float sum = 0.f;
  for (int i = tid; i < n4; i += 256) {
    const f32x4_t v = s[i];
    sum += (__builtin_amdgcn_exp2f(fmaf(v[0], c, -mc)) + __builtin_amdgcn_exp2f(fmaf(v[1], c, -mc))) +
           (__builtin_amdgcn_exp2f(fmaf(v[2], c, -mc)) + __builtin_amdgcn_exp2f(fmaf(v[3], c, -mc)));
  }
  sum += et;  // (after the vector part: rows with N % 4 == 0 sum exactly as before)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  const float inv = 1.0f / sum;
  for (int i = tid; i < n4; i += 256) {
    const f32x4_t v = s[i];
    uint2 w;
    w.x = pack_bf2(__builtin_amdgcn_exp2f(fmaf(v[0], c, -mc)) * inv, __builtin_amdgcn_exp2f(fmaf(v[1], c, -mc)) * inv);
    w.y = pack_bf2(__builtin_amdgcn_exp2f(fmaf(v[2], c, -mc)) * inv, __builtin_amdgcn_exp2f(fmaf(v[3], c, -mc)) * inv);
    o[i] = w;
  }
  if (tail) prow[4 * n4 + tid] = f2bf(et * inv);
}

}  // namespace

extern "C" size_t dm4d_groupnorm_ws_bytes(int B, int HW, int groups) {
  return (size_t)B * gn_nchunk(B, HW) * groups * 2 * sizeof(float);
}

template <int IO>
static int groupnorm_impl(void* stream, const void* X1, int C1, const void* X2, int C2, int B, int HW, int groups, float eps,
                          const void* gamma, const void* beta, void* Y, int apply_silu, void* ws, void* Yraw = nullptr) {
  if (!X1 || !gamma || !beta || !Y || !ws || B <= 0 || HW <= 0 || groups <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "groupnorm: null pointer or empty shape");
  if (!X2) C2 = 0;
  const int C = C1 + C2;
  if ((C1 & 7) || (C2 & 7) || C % groups != 0 || C > GN_MAXC)
    return dm4d_set_error(DM4D_ERR_ARG, "groupnorm: channels must be multiples of 8, divisible by groups, <= 4096");
  GNParams p{X1, X2, C1, C2, B, HW, groups, gn_nchunk(B, HW), eps,
             (const u16*)gamma, (const u16*)beta, (u16*)Y, apply_silu, (float*)ws, (u16*)Yraw};
  hipStream_t st = (hipStream_t)stream;
  {
    int gps, SV, RPB, NV;
    if (g_gn_resident && gn_resident_plan(C, HW, groups, &gps, &SV, &RPB, &NV)) {
      const int NS = groups / gps, xcd = (B % 8 == 0);
      const dim3 grid(B * NS), blk(GN_THREADS);
      if (NV <= 2) hipLaunchKernelGGL((gn_resident_kernel<2, IO>), grid, blk, 0, st, p, gps, SV, RPB, NS, xcd);
      else if (NV <= 4) hipLaunchKernelGGL((gn_resident_kernel<4, IO>), grid, blk, 0, st, p, gps, SV, RPB, NS, xcd);
      else if (NV <= 8) hipLaunchKernelGGL((gn_resident_kernel<8, IO>), grid, blk, 0, st, p, gps, SV, RPB, NS, xcd);
      else hipLaunchKernelGGL((gn_resident_kernel<16, IO>), grid, blk, 0, st, p, gps, SV, RPB, NS, xcd);
      return dm4d_check_launch("gn_resident_kernel");
    }
  }
  const int CV = C / 8;
  const int PPB = CV >= GN_THREADS ? 1 : GN_THREADS / CV;
  const size_t sm1 = (size_t)PPB * C * 2 * sizeof(float);
  const size_t sm2 = ((size_t)(2 * C > 2 * GN_THREADS ? 2 * C : 2 * GN_THREADS) + 2 * groups) * sizeof(float);
  hipLaunchKernelGGL(gn_stats_kernel<IO>, dim3(B * p.nchunk), dim3(GN_THREADS), sm1, st, p);
  int rc = dm4d_check_launch("gn_stats_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_apply_kernel<IO>, dim3(B * p.nchunk), dim3(GN_THREADS), sm2, st, p);
  return dm4d_check_launch("gn_apply_kernel");
}

extern "C" int dm4d_groupnorm_nhwc_bf16(void* stream, const void* X1, int C1, const void* X2, int C2, int B, int HW,
                                        int groups, float eps, const void* gamma, const void* beta, void* Y,
                                        int apply_silu, void* ws) {
  return groupnorm_impl<0>(stream, X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, Y, apply_silu, ws);
}

// precision "fp16": fp32 NHWC in (shifted fp32 statistics as above), one fp16 plane out.  Channel counts that are not multiples
// of 8 take the general kernels of parity.hip (dm4d_groupnorm_f32_f16_general: fp64 statistics, one / four channels per thread).
extern "C" int dm4d_groupnorm_f32_f16_general(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW, int groups,
                                              float eps, const void* gamma, const void* beta, void* Y, int apply_silu, void* ws);
extern "C" int dm4d_groupnorm_nhwc_f32_f16(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW, int groups,
                                           float eps, const void* gamma, const void* beta, void* Y, int apply_silu, void* ws) {
  if (!X2) C2 = 0;
  if ((C1 & 7) || (C2 & 7) || (((uintptr_t)X1) & 15) || (X2 && (((uintptr_t)X2) & 15)) || (((uintptr_t)Y) & 15))
    return dm4d_groupnorm_f32_f16_general(stream, X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, Y, apply_silu, ws);
  return groupnorm_impl<2>(stream, X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, Y, apply_silu, ws);
}

// precision "fp16", fp16 in: the input is an activation its producer already rounded to fp16 (same statistics, same output rounding;
// half the bytes of the fp32 form on the way in).  Channel counts must be multiples of 8, tensors 16-byte aligned.
extern "C" int dm4d_groupnorm_nhwc_f16_f16(void* stream, const void* X1, int C1, const void* X2, int C2, int B, int HW, int groups,
                                           float eps, const void* gamma, const void* beta, void* Y, int apply_silu, void* ws) {
  if (!X2) C2 = 0;
  if ((C1 & 7) || (C2 & 7) || (((uintptr_t)X1) & 15) || (X2 && (((uintptr_t)X2) & 15)) || (((uintptr_t)Y) & 15))
    return dm4d_set_error(DM4D_ERR_ARG, "groupnorm_f16_f16: channel counts must be multiples of 8, tensors 16-byte aligned");
  return groupnorm_impl<3>(stream, X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, Y, apply_silu, ws);
}

extern "C" int dm4d_groupnorm_nhwc_f32_f16_raw(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW, int groups,
                                               float eps, const void* gamma, const void* beta, void* Y, void* Yraw, int apply_silu,
                                               void* ws) {
  if (!X2) C2 = 0;
  if (!Yraw || (C1 & 7) || (C2 & 7) || (((uintptr_t)X1) & 15) || (X2 && (((uintptr_t)X2) & 15)) || ((((uintptr_t)Y) | ((uintptr_t)Yraw)) & 15))
    return dm4d_set_error(DM4D_ERR_ARG, "groupnorm_f32_f16_raw: needs Yraw, channel counts that are multiples of 8 and 16-byte aligned tensors");
  return groupnorm_impl<2>(stream, X1, C1, X2, C2, B, HW, groups, eps, gamma, beta, Y, apply_silu, ws, Yraw);
}

template <int IO>
static int layernorm_impl(void* stream, const void* X, int64_t ldx, const void* gamma, const void* beta, void* Y, int64_t ldy, int M,
                          int C, float eps) {
  typedef typename NormIO<IO>::in_t in_t;
  hipStream_t st = (hipStream_t)stream;
  const int nv = (C / 8 + 63) / 64;
  dim3 grid((M + 3) / 4), block(256);
#define LN_LAUNCH(NV) \
  hipLaunchKernelGGL((ln_kernel<NV, IO>), grid, block, 0, st, (const in_t*)X, ldx, (const u16*)gamma, (const u16*)beta, (u16*)Y, ldy, M, C, eps)
  if (nv <= 1) LN_LAUNCH(1);
  else if (nv <= 2) LN_LAUNCH(2);
  else if (nv <= 3) LN_LAUNCH(3);
  else if (nv <= 4) LN_LAUNCH(4);
  else LN_LAUNCH(8);
#undef LN_LAUNCH
  return dm4d_check_launch("ln_kernel");
}

extern "C" int dm4d_layernorm_bf16(void* stream, const void* X, int64_t ldx, const void* gamma, const void* beta,
                                   void* Y, int64_t ldy, int M, int C, float eps) {
  if (!X || !gamma || !beta || !Y || M <= 0 || C <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "layernorm: null pointer or empty shape");
  if ((C & 7) || (ldx & 7) || (ldy & 7) || C > 64 * 8 * 8)
    return dm4d_set_error(DM4D_ERR_ARG, "layernorm: C must be a multiple of 8 and <= 4096");
  return layernorm_impl<0>(stream, X, ldx, gamma, beta, Y, ldy, M, C, eps);
}

// precision "fp16": fp32 rows in (kept in registers: read once), one fp16 plane out.  Rows that are not 16-byte vectors of eight
// channels take the general kernel of parity.hip.
extern "C" int dm4d_layernorm_f32_f16_general(void* stream, const float* X, int64_t ldx, const void* gamma, const void* beta, void* Y,
                                              int64_t ldy, int M, int C, float eps);
extern "C" int dm4d_layernorm_f32_f16(void* stream, const float* X, int64_t ldx, const void* gamma, const void* beta, void* Y,
                                      int64_t ldy, int M, int C, float eps) {
  if (!X || !gamma || !beta || !Y || M <= 0 || C <= 0 || ldx < C || ldy < (int64_t)C)
    return dm4d_set_error(DM4D_ERR_ARG, "layernorm_f32_f16: bad arguments");
  if ((C & 7) || (ldx & 3) || (ldy & 7) || C > 64 * 8 * 8 || (((uintptr_t)X) & 15) || (((uintptr_t)Y) & 15) || (((uintptr_t)gamma) & 15) ||
      (((uintptr_t)beta) & 15))
    return dm4d_layernorm_f32_f16_general(stream, X, ldx, gamma, beta, Y, ldy, M, C, eps);
  return layernorm_impl<2>(stream, X, ldx, gamma, beta, Y, ldy, M, C, eps);
}

extern "C" int dm4d_softmax_rows_bf16(void* stream, const void* S, int64_t lds, void* P, int64_t ldp, int M, int N,
                                      float scale) {
  if (!S || !P || M <= 0 || N <= 0) return dm4d_set_error(DM4D_ERR_ARG, "softmax: null pointer or empty shape");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const u16*)S, lds, (u16*)P, ldp,
                     N, scale);
  return dm4d_check_launch("softmax_rows_kernel");
}

extern "C" int dm4d_softmax_rows_f32in_bf16(void* stream, const float* S, int64_t lds, void* P, int64_t ldp, int M, int N,
                                            float scale) {
  if (!S || !P || M <= 0 || N <= 0) return dm4d_set_error(DM4D_ERR_ARG, "softmax: null pointer or empty shape");
  if ((lds & 3) || (ldp & 3) || lds < N || ldp < N || (((uintptr_t)S) & 15) || (((uintptr_t)P) & 7))
    return dm4d_set_error(DM4D_ERR_ARG, "softmax (fp32 logits): row strides must be multiples of 4 and >= N, rows 16-byte aligned");
  hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, S, lds, (u16*)P, ldp, N, scale);
  return dm4d_check_launch("softmax_rows_f32_kernel");
}

extern "C" int dm4d_tune_set_groupnorm_resident(int on) {
  g_gn_resident = on;
  return DM4D_OK;
}
