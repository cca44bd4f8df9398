// Direct convolution for the PoseEncoder's thin layers (pose_encoder.py:14-31: 3->3->16->16->32->32->64->64->128
// channels, 3x3 stride 1 and 4x4 stride 2, SiLU after each).  With 3..64 input channels the implicit-GEMM MFMA
// kernel would run at a fraction of a tile, so these layers use fp32 FMAs: one thread owns one output pixel and COB
// output channels, the weights of the block's COB output channels sit in LDS as fp32 [tap][ci][COB] (every lane
// reads the same address: a broadcast ds_read_b128 per 4 FMAs), the input pixel is read as packed bf16 channels.
// The whole encoder is ~1.5 GFLOP per 576x320 image and runs once per task, not once per denoising step.
#include "common.h"
#include "dm4d.h"
#include "errors.h"

namespace {

// T = u16: bf16 tensors (fast precision).  T = float: fp32 image, weights, bias and result (parity precision: the same fp32 FMA
// chain, nothing rounded to bf16 on the way).
template <typename T>
struct DirectConvParams {
  const T* X;
  const T* Wt;    // [Cout][ks*ks][Cin]
  const T* bias;  // [Cout] or null
  T* Y;
  int B, H, W, Cin, Ho, Wo, Cout, ks, stride, pad, silu;
};

__device__ __forceinline__ float ld1(const u16* p) { return bf2f(*p); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ void ld4(const u16* p, float* xv) {
  const uint2 raw = *reinterpret_cast<const uint2*>(p);
  xv[0] = __uint_as_float(raw.x << 16), xv[1] = __uint_as_float(raw.x & 0xFFFF0000u);
  xv[2] = __uint_as_float(raw.y << 16), xv[3] = __uint_as_float(raw.y & 0xFFFF0000u);
}
__device__ __forceinline__ void ld4(const float* p, float* xv) {
  const f32x4_t v = *reinterpret_cast<const f32x4_t*>(p);
  xv[0] = v[0], xv[1] = v[1], xv[2] = v[2], xv[3] = v[3];
}
__device__ __forceinline__ void st4(u16* p, const float* v) {
  uint2 pk;
  pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = pk;
}
__device__ __forceinline__ void st4(float* p, const float* v) {
  f32x4_t o = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4_t*>(p) = o;
}

template <int COB, typename T>
__global__ __launch_bounds__(256) void conv_direct_kernel(DirectConvParams<T> p) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [ks*ks*Cin][COB]
  const int co0 = blockIdx.y * COB;
  const int taps = p.ks * p.ks;
  const int kc = taps * p.Cin;
  for (int i = threadIdx.x; i < kc * COB; i += 256) {
    const int co = i % COB, k = i / COB;
    wsm[i] = ld1(p.Wt + (int64_t)(co0 + co) * kc + k);
  }
  __syncthreads();
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)p.B * p.Ho * p.Wo;
  if (o >= total) return;
  const int ox = (int)(o % p.Wo);
  const int64_t t = o / p.Wo;
  const int oy = (int)(t % p.Ho);
  const int b = (int)(t / p.Ho);
  float acc[COB];
#pragma unroll
  for (int c = 0; c < COB; ++c) acc[c] = p.bias ? ld1(p.bias + co0 + c) : 0.f;
  const T* img = p.X + (int64_t)b * p.H * p.W * p.Cin;
  for (int ky = 0; ky < p.ks; ++ky) {
    const int iy = oy * p.stride - p.pad + ky;
    if ((unsigned)iy >= (unsigned)p.H) continue;
    for (int kx = 0; kx < p.ks; ++kx) {
      const int ix = ox * p.stride - p.pad + kx;
      if ((unsigned)ix >= (unsigned)p.W) continue;
      const T* px = img + ((int64_t)iy * p.W + ix) * p.Cin;
      const float* wt = wsm + (ky * p.ks + kx) * p.Cin * COB;
      for (int c4 = 0; c4 < p.Cin; c4 += 4) {
        float xv[4];
        ld4(px + c4, xv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* w = wt + (c4 + e) * COB;
#pragma unroll
          for (int c = 0; c < COB; c += 4) {
            const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(w + c);
            acc[c + 0] = fmaf(xv[e], wv[0], acc[c + 0]);
            acc[c + 1] = fmaf(xv[e], wv[1], acc[c + 1]);
            acc[c + 2] = fmaf(xv[e], wv[2], acc[c + 2]);
            acc[c + 3] = fmaf(xv[e], wv[3], acc[c + 3]);
          }
        }
      }
    }
  }
  T* dst = p.Y + o * p.Cout + co0;
#pragma unroll
  for (int c = 0; c < COB; c += 4) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = p.silu ? silu_f(acc[c + e]) : acc[c + e];
    st4(dst + c, v);
  }
}

template <typename T>
int conv_direct_impl(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, const void* bias, void* Y, int Ho, int Wo,
                     int Cout, int ksize, int stride, int pad, int silu) {

  if (!X || !Wt || !Y || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "conv2d_direct: null pointer or empty shape");
  if (Cin <= 0 || Cin % 4 != 0 || Cout <= 0 || Cout % 4 != 0)
    return dm4d_set_error(DM4D_ERR_ARG, "conv2d_direct: Cin and Cout must be multiples of 4 (zero-pad)");
  if (ksize < 1 || ksize > 4 || stride < 1 || pad < 0) return dm4d_set_error(DM4D_ERR_ARG, "conv2d_direct: bad kernel geometry");
  if (Ho != (H + 2 * pad - ksize) / stride + 1 || Wo != (W + 2 * pad - ksize) / stride + 1)
    return dm4d_set_error(DM4D_ERR_ARG, "conv2d_direct: output size does not match the geometry");
  const int cob = (Cout % 16 == 0) ? 16 : 4;
  const size_t smem = (size_t)ksize * ksize * Cin * cob * sizeof(float);
  if (smem > 64 * 1024) return dm4d_set_error(DM4D_ERR_ARG, "conv2d_direct: weights of one channel block exceed 64 KiB of LDS");
  DirectConvParams<T> p{(const T*)X, (const T*)Wt, (const T*)bias, (T*)Y, B, H, W, Cin, Ho, Wo, Cout, ksize, stride, pad, silu};
  const int64_t total = (int64_t)B * Ho * Wo;
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)(Cout / cob));
  if (cob == 16) {
    hipLaunchKernelGGL((conv_direct_kernel<16, T>), grid, dim3(256), smem, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL((conv_direct_kernel<4, T>), grid, dim3(256), smem, (hipStream_t)stream, p);
  }
  return dm4d_check_launch("conv_direct_kernel");
}

}  // namespace

extern "C" int dm4d_conv2d_direct_nhwc_bf16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt,
                                            const void* bias, void* Y, int Ho, int Wo, int Cout, int ksize, int stride,
                                            int pad, int silu) {
  return conv_direct_impl<u16>(stream, X, B, H, W, Cin, Wt, bias, Y, Ho, Wo, Cout, ksize, stride, pad, silu);
}

extern "C" int dm4d_conv2d_direct_nhwc_f32(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt,
                                           const void* bias, void* Y, int Ho, int Wo, int Cout, int ksize, int stride,
                                           int pad, int silu) {
  return conv_direct_impl<float>(stream, X, B, H, W, Cin, Wt, bias, Y, Ho, Wo, Cout, ksize, stride, pad, silu);
}
