// Small HBM-bound kernels around the UNet call: timestep embedding, SiLU, model-input packing,
// fused CFG + batched per-latent DDIM step, NCHW<->NHWC converters at the pipeline boundary.
#include "common.h"
#include "dm4d.h"
#include "errors.h"

namespace {

// Storage type of a tensor: u16 = bf16 (the fast path), float = the parity-precision path (host/ops.py precision "parity"); the
// arithmetic between load and store is fp32 either way, so the bf16 instantiations are the kernels they always were.
__device__ __forceinline__ float ldv(const u16* p) { return bf2f(*p); }
__device__ __forceinline__ float ldv(const float* p) { return *p; }
__device__ __forceinline__ void stv(u16* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void stv(float* p, float v) { *p = v; }

template <typename T>
__global__ void timestep_embedding_kernel(const float* t, T* out, int B, int dim, int flip, float freq_shift) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i % half;
  // diffusers get_timestep_embedding: exponent = -ln(10000) * k / (half - shift), all in fp32
  const float exponent = -9.210340371976184f * (float)k / ((float)half - freq_shift);
  const float arg = t[b] * expf(exponent);
  const float s = sinf(arg), c = cosf(arg);
  T* o = out + (int64_t)b * dim;
  if (flip) {
    stv(o + k, c);
    stv(o + half + k, s);
  } else {
    stv(o + k, s);
    stv(o + half + k, c);
  }
  if ((dim & 1) && k == 0) stv(o + dim - 1, 0.f);
}

__global__ void silu_kernel(const u16* X, u16* Y, int64_t n) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (; i + 8 <= n; i += stride) {
    float v[8];
    unpack8(ldg16(X + i), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    stg16(Y + i, pack8(v));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t j = n & ~(int64_t)7; j < n; ++j) Y[j] = f2bf(silu_f(bf2f(X[j])));
}

// one thread per (frame, pixel): reads the 4+6+4+1 conditioning channels, writes both CFG halves.
// T = u16: `out` rows are cpad bf16 channels.  T = float (parity precision): the task tensors are fp32 and `out` rows are the
// two-term operand of conv_in, [hi(cpad) | lo(cpad)].
// OUT: 0 = bf16 row, 1 = two-term operand [hi(cpad) | lo(cpad)] (parity precision), 2 = fp16 row (precision "fp16")
template <int OUT>
__device__ __forceinline__ void put_in(u16* row, int c, int cpad, float v) {  // bf16 values survive the float round trip bit for bit
  if constexpr (OUT == 2) {
    row[c] = f2h_sat(v);
  } else {
    const u16 hi = f2bf(v);
    row[c] = hi;
    if constexpr (OUT == 1) row[cpad + c] = f2bf(v - bf2f(hi));
  }
}
template <typename T, bool H16 = false>
__global__ void pack_kernel(T* latents, const T* pv, const T* pl, const T* sk, const T* mask, const int32_t* is_cond,
                            const int32_t* frame_idx, u16* out, int F, int HW, int cpad, int use_cfg) {
  constexpr bool SPLIT = sizeof(T) == 4 && !H16;
  constexpr int OUT = H16 ? 2 : (SPLIT ? 1 : 0);
  const int ldo = SPLIT ? 2 * cpad : cpad;
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // window-local (frame, pixel)
  if (o >= (int64_t)F * HW) return;
  const int f = (int)(o / HW);
  const bool cond = is_cond[f] != 0;
  // i = (frame, pixel) inside the task-level tensors the window is gathered from
  const int64_t i = frame_idx ? (int64_t)frame_idx[f] * HW + (o - (int64_t)f * HW) : o;
  auto put = [&](u16* row, int c, float v) { put_in<OUT>(row, c, cpad, v); };
  float lat[4];
  if (cond) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      lat[c] = ldv(pv + i * 4 + c);
      latents[i * 4 + c] = pv[i * 4 + c];  // reference aliasing side effect (pipeline_diffuman4d.py:375-379)
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) lat[c] = ldv(latents + i * 4 + c);
  }
  const float m = ldv(mask + i);
  const int nsk = sk ? 4 : 0;
  u16* pos = out + ((use_cfg ? (int64_t)F * HW : 0) + o) * ldo;
  int c = 0;
  for (int k = 0; k < 4; ++k) put(pos, c++, lat[k]);
  for (int k = 0; k < 6; ++k) put(pos, c++, ldv(pl + i * 6 + k));
  for (int k = 0; k < nsk; ++k) put(pos, c++, ldv(sk + i * 4 + k));
  put(pos, c++, m);
  for (; c < cpad; ++c) put(pos, c, 0.f);
  if (use_cfg) {
    u16* neg = out + o * ldo;
    c = 0;
    for (int k = 0; k < 4; ++k) put(neg, c++, cond ? 1.0f : lat[k]);
    for (int k = 0; k < 6; ++k) put(neg, c++, 0.f);
    for (int k = 0; k < nsk; ++k) put(neg, c++, -1.0f);
    put(neg, c++, m);
    for (; c < cpad; ++c) put(neg, c, 0.f);
  }
}

// latents [F,HW,4], noise_pred [cfg*F, HW, ldn] (first 4 channels used)
template <typename T>
__global__ void cfg_ddim_kernel(T* latents, const T* np, int64_t ldn, const float* coef, const int32_t* is_cond,
                                const int32_t* frame_idx, int F, int HW, int use_cfg, float gs, int vpred) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // window-local (frame, pixel)
  if (o >= (int64_t)F * HW) return;
  const int f = (int)(o / HW);
  if (is_cond[f] != 0) return;  // "only denoise target latents" (pipeline_diffuman4d.py:418-420)
  const int64_t i = frame_idx ? (int64_t)frame_idx[f] * HW + (o - (int64_t)f * HW) : o;
  const float sa = coef[f * 4 + 0], sb = coef[f * 4 + 1], sap = coef[f * 4 + 2], sbp = coef[f * 4 + 3];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float e;
    if (use_cfg) {
      const float u = ldv(np + o * ldn + c);
      const float cc = ldv(np + ((int64_t)F * HW + o) * ldn + c);
      e = u + gs * (cc - u);
    } else {
      e = ldv(np + o * ldn + c);
    }
    const float x = ldv(latents + i * 4 + c);
    float x0, eps;
    if (vpred) {
      x0 = sa * x - sb * e;
      eps = sa * e + sb * x;
    } else {
      x0 = (x - sb * e) / sa;
      eps = e;
    }
    stv(latents + i * 4 + c, sap * x0 + sbp * eps);
  }
}

// Linear multistep update (DPM-Solver++ and every other solver whose update is linear in the sample, the model output and one
// stored prediction): x' = a x + b m + c p,  p' = d x + e m with per-frame rows (a, b, c, d, e, -, -, -) planned on the host
// (host/scheduler.py); m = the guided model output, p = x0_prev (the latent's previous x0 prediction, updated in place).
template <typename T>
__global__ void cfg_linear_step_kernel(T* latents, T* x0_prev, const T* np, int64_t ldn, const float* coef,
                                       const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg, float gs) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // window-local (frame, pixel)
  if (o >= (int64_t)F * HW) return;
  const int f = (int)(o / HW);
  if (is_cond[f] != 0) return;  // "only denoise target latents" (pipeline_diffuman4d.py:418-420)
  const int64_t i = frame_idx ? (int64_t)frame_idx[f] * HW + (o - (int64_t)f * HW) : o;
  const float a = coef[f * 8 + 0], b = coef[f * 8 + 1], c = coef[f * 8 + 2], d = coef[f * 8 + 3], e = coef[f * 8 + 4];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    float m;
    if (use_cfg) {
      const float u = ldv(np + o * ldn + ch);
      const float cc = ldv(np + ((int64_t)F * HW + o) * ldn + ch);
      m = u + gs * (cc - u);
    } else {
      m = ldv(np + o * ldn + ch);
    }
    const float x = ldv(latents + i * 4 + ch);
    const float p = c != 0.f ? ldv(x0_prev + i * 4 + ch) : 0.f;  // first step of a latent in a call: the slot holds nothing yet
    stv(latents + i * 4 + ch, a * x + b * m + c * p);
    stv(x0_prev + i * 4 + ch, d * x + e * m);
  }
}

// General linear multistep update with up to three stored tensors per latent (UniPC with its corrector, DEIS of order <= 3): rows of
// 16 floats planned on the host (host/scheduler.py, block comment above UniPCConfig):
//   m = guided model output;  conv = k0 x + k1 m;  xc = k2 x + k3 s3 + k4 s1 + k5 s2 + k6 conv;  x' = k7 xc + k8 conv + k9 s1 + k10 s2
//   s3' = xc, s2' = s1, s1' = conv.   s2 / s3 may be NULL (order-1 / DEIS schedulers keep fewer tensors).  The stored tensors start a
// call as ZEROS (the host allocates them so): a latent's first steps meet zero coefficients on them.
// Two more fields serve PNDM's linear multistep form (PLMS, four model outputs deep): k11 = weight of s3 in x', and k12 = what the stored
// tensors become -- 0: as above;  1: kept as they are (the repeated second step of PLMS re-does the first from the stored sample and must
// not enter the history);  2: shifted as a history, s3' = s2, s2' = s1, s1' = conv.
template <typename T>
__global__ void cfg_multistep_kernel(T* latents, T* s1, T* s2, T* s3, const T* np, int64_t ldn, const float* coef, const int32_t* is_cond,
                                     const int32_t* frame_idx, int F, int HW, int use_cfg, float gs) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // window-local (frame, pixel)
  if (o >= (int64_t)F * HW) return;
  const int f = (int)(o / HW);
  if (is_cond[f] != 0) return;  // "only denoise target latents" (pipeline_diffuman4d.py:418-420)
  const int64_t i = frame_idx ? (int64_t)frame_idx[f] * HW + (o - (int64_t)f * HW) : o;
  const float* k = coef + f * 16;
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    float m;
    if (use_cfg) {
      const float u = ldv(np + o * ldn + ch);
      const float cc = ldv(np + ((int64_t)F * HW + o) * ldn + ch);
      m = u + gs * (cc - u);
    } else {
      m = ldv(np + o * ldn + ch);
    }
    const float x = ldv(latents + i * 4 + ch);
    const float p1 = ldv(s1 + i * 4 + ch), p2 = s2 ? ldv(s2 + i * 4 + ch) : 0.f, p3 = s3 ? ldv(s3 + i * 4 + ch) : 0.f;
    const float conv = k[0] * x + k[1] * m;
    const float xc = k[2] * x + k[3] * p3 + k[4] * p1 + k[5] * p2 + k[6] * conv;
    stv(latents + i * 4 + ch, k[7] * xc + k[8] * conv + k[9] * p1 + k[10] * p2 + k[11] * p3);
    const int keep = (int)k[12];
    if (keep == 1) continue;
    if (s3) stv(s3 + i * 4 + ch, keep == 2 ? p2 : xc);
    if (s2) stv(s2 + i * 4 + ch, p1);
    stv(s1 + i * 4 + ch, conv);
  }
}

__global__ void nchw_to_nhwc_kernel(const u16* X, u16* Y, int B, int C, int HW, int cpad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*HW*cpad
  if (i >= (int64_t)B * HW * cpad) return;
  const int c = (int)(i % cpad);
  const int64_t bp = i / cpad;
  const int px = (int)(bp % HW), b = (int)(bp / HW);
  Y[i] = c < C ? X[((int64_t)b * C + c) * HW + px] : (u16)0;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* X, T* Y, int B, int C, int HW, int ldx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*C*HW
  if (i >= (int64_t)B * C * HW) return;
  const int px = (int)(i % HW);
  const int64_t bc = i / HW;
  const int c = (int)(bc % C), b = (int)(bc / C);
  Y[i] = X[((int64_t)b * HW + px) * ldx + c];
}

// VAE posterior sample (DiagonalGaussianDistribution.sample) * scaling_factor
template <typename T>
__global__ void vae_sample_kernel(const T* mom, int64_t ldm, const T* noise, T* out, int64_t M, int C, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const int64_t m = i / C;
  const int c = (int)(i % C);
  const float mean = ldv(mom + m * ldm + c);
  float logvar = ldv(mom + m * ldm + C + c);
  logvar = fminf(fmaxf(logvar, -30.f), 20.f);
  stv(out + i, (mean + expf(0.5f * logvar) * ldv(noise + i)) * scale);
}

__global__ void scale_pad_kernel(const u16* X, int64_t ldx, u16* Y, int cpad, int64_t M, int C, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * cpad) return;
  const int64_t m = i / cpad;
  const int c = (int)(i % cpad);
  Y[i] = c < C ? f2bf(bf2f(X[m * ldx + c]) * scale) : (u16)0;
}

// F.interpolate(x, size=(h,w), mode="bilinear"|"nearest") (align_corners=False, no antialias), fp32 NCHW in,
// bf16 NHWC out -- pipeline_diffuman4d.py:90-100 computes this in fp32 and casts afterwards.
template <typename T>
__global__ void resize_kernel(const float* X, T* Y, int B, int C, int H, int W, int h, int w, int bilinear) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*h*w*C (C fastest)
  if (i >= (int64_t)B * h * w * C) return;
  const int c = (int)(i % C);
  int64_t r = i / C;
  const int ox = (int)(r % w);
  r /= w;
  const int oy = (int)(r % h);
  const int b = (int)(r / h);
  const float* src = X + ((int64_t)b * C + c) * H * W;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  float v;
  if (bilinear) {
    float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    v = hy * (hx * src[(int64_t)y0 * W + x0] + lx * src[(int64_t)y0 * W + x1]) +
        ly * (hx * src[(int64_t)y1 * W + x0] + lx * src[(int64_t)y1 * W + x1]);
  } else {
    int y0 = (int)floorf((float)oy * sy), x0 = (int)floorf((float)ox * sx);
    y0 = y0 > H - 1 ? H - 1 : y0;
    x0 = x0 > W - 1 ? W - 1 : x0;
    v = src[(int64_t)y0 * W + x0];
  }
  stv(Y + i, v);
}

// VaeImageProcessor.postprocess(denormalize): (x / 2 + 0.5).clamp(0, 1); NHWC (ld >= C) -> NCHW
template <typename T>
__global__ void postprocess_kernel(const T* X, T* Y, int B, int C, int HW, int ldx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*C*HW
  if (i >= (int64_t)B * C * HW) return;
  const int px = (int)(i % HW);
  const int64_t bc = i / HW;
  const int c = (int)(bc % C), b = (int)(bc / C);
  float v = ldv(X + ((int64_t)b * HW + px) * ldx + c) * 0.5f + 0.5f;
  v = fminf(fmaxf(v, 0.f), 1.f);
  stv(Y + i, v);
}

// Pluecker ray maps at LATENT resolution, straight from the cameras: what calc_plucker_embeds (ray_utils.py:101-112: rays
// through the pixel centres of the H x W image, d = normalize(R^T (K^-1 [x+.5, y+.5, 1]^T - T) - o), o = -R^T T,
// embed = [d | o x d]) followed by F.interpolate(size=(h, w), mode="bilinear") (pipeline_diffuman4d.py:90-100) produces,
// without ever building the [N, 6, H, W] fp32 map (25 MB per 1024^2 frame, 7.5 GB per temporal task of 300 frames) on the
// host: a latent pixel is the bilinear blend of the 2 x 2 full-resolution pixels around its sample point, so the kernel
// evaluates those four rays and blends them with F.interpolate's weights.  cam[n] = {invK (9, row major), R (9), T (3),
// o (3)} fp32, prepared on the host from K and the camera-to-world pose exactly as the reference does (torch.inverse).
// One thread per latent pixel; out [N, h*w, 6] bf16.
struct Ray6 {
  float v[6];
};
__device__ __forceinline__ Ray6 plucker_ray(const float* cam, float px, float py) {
  const float* iK = cam;
  const float* R = cam + 9;
  const float* T = cam + 18;
  const float* o = cam + 21;
  // pixel_camera = invK @ [x, y, 1]
  const float c0 = iK[0] * px + iK[1] * py + iK[2], c1 = iK[3] * px + iK[4] * py + iK[5], c2 = iK[6] * px + iK[7] * py + iK[8];
  const float t0 = c0 - T[0], t1 = c1 - T[1], t2 = c2 - T[2];
  // pixel_world = R^T @ (pixel_camera - T);  ray_d = pixel_world - ray_o
  float d0 = (R[0] * t0 + R[3] * t1 + R[6] * t2) - o[0];
  float d1 = (R[1] * t0 + R[4] * t1 + R[7] * t2) - o[1];
  float d2 = (R[2] * t0 + R[5] * t1 + R[8] * t2) - o[2];
  const float inv = 1.0f / (sqrtf(d0 * d0 + d1 * d1 + d2 * d2) + 1e-8f);  // normalize(): x / (|x| + eps)
  d0 *= inv;
  d1 *= inv;
  d2 *= inv;
  Ray6 r;
  r.v[0] = d0;
  r.v[1] = d1;
  r.v[2] = d2;
  r.v[3] = o[1] * d2 - o[2] * d1;  // o x d
  r.v[4] = o[2] * d0 - o[0] * d2;
  r.v[5] = o[0] * d1 - o[1] * d0;
  return r;
}

template <typename T>
__global__ void plucker_latent_kernel(const float* cams, T* Y, int N, int H, int W, int h, int w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over N*h*w
  if (i >= (int64_t)N * h * w) return;
  const int ox = (int)(i % w);
  const int64_t r = i / w;
  const int oy = (int)(r % h), n = (int)(r / h);
  const float* cam = cams + (int64_t)n * 24;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  // pixel centres (correct_pix): column x + 0.5, row y + 0.5
  const Ray6 a = plucker_ray(cam, (float)x0 + 0.5f, (float)y0 + 0.5f), b = plucker_ray(cam, (float)x1 + 0.5f, (float)y0 + 0.5f);
  const Ray6 c = plucker_ray(cam, (float)x0 + 0.5f, (float)y1 + 0.5f), d = plucker_ray(cam, (float)x1 + 0.5f, (float)y1 + 0.5f);
  T* y = Y + i * 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) stv(y + k, hy * (hx * a.v[k] + lx * b.v[k]) + ly * (hx * c.v[k] + lx * d.v[k]));
}

// F.interpolate(x, size=(h, w), mode="bilinear", antialias=True, align_corners=False) on fp32 NCHW planes: the down-scale of the
// result writer's snapshot mosaic (sampling_utils.py:70-93 resizes the grid with torchvision's antialiased `resize`).  PIL's /
// ATen's separable triangle filter: output i covers [c - s, c + s) around c = scale (i + 0.5) with support s = max(scale, 1);
// tap j has weight max(0, 1 - |(j + 0.5 - c) / max(scale, 1)|), normalised per axis.  One thread per output pixel.
__device__ __forceinline__ void aa_span(int i, float scale, int in_size, int& lo, int& n, float& center, float& inv) {
  const float support = scale >= 1.0f ? scale : 1.0f;
  inv = scale >= 1.0f ? 1.0f / scale : 1.0f;
  center = scale * ((float)i + 0.5f);
  lo = max((int)(center - support + 0.5f), 0);
  n = min((int)(center + support + 0.5f), in_size) - lo;
}
__device__ __forceinline__ float aa_weight(int j, int lo, float center, float inv) {
  const float x = fabsf(((float)(j + lo) - center + 0.5f) * inv);
  return x < 1.0f ? 1.0f - x : 0.0f;
}
__global__ void resize_aa_kernel(const float* X, float* Y, int64_t planes, int H, int W, int h, int w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over planes * h * w
  if (i >= planes * h * w) return;
  const int ox = (int)(i % w);
  const int64_t r = i / w;
  const int oy = (int)(r % h);
  const float* src = X + (r / h) * (int64_t)H * W;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  int y0, ny, x0, nx;
  float cy, iy, cx, ix;
  aa_span(oy, sy, H, y0, ny, cy, iy);
  aa_span(ox, sx, W, x0, nx, cx, ix);
  float wy_sum = 0.f, wx_sum = 0.f;
  for (int j = 0; j < ny; ++j) wy_sum += aa_weight(j, y0, cy, iy);
  for (int k = 0; k < nx; ++k) wx_sum += aa_weight(k, x0, cx, ix);
  float acc = 0.f;
  for (int j = 0; j < ny; ++j) {
    const float wy = aa_weight(j, y0, cy, iy) / wy_sum;
    const float* row = src + (int64_t)(y0 + j) * W + x0;
    float racc = 0.f;
    for (int k = 0; k < nx; ++k) racc += (aa_weight(k, x0, cx, ix) / wx_sum) * row[k];
    acc += wy * racc;
  }
  Y[i] = acc;
}

inline dim3 grid1d(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

// One implementation per operator, instantiated for bf16 tensors (the fast path) and fp32 tensors (parity precision).
template <typename T>
static int timestep_embedding_impl(void* stream, const float* t, void* out, int B, int dim, int flip, float freq_shift) {
  if (!t || !out || B <= 0 || dim < 2) return dm4d_set_error(DM4D_ERR_ARG, "timestep_embedding: bad arguments");
  hipLaunchKernelGGL(timestep_embedding_kernel<T>, grid1d((int64_t)B * (dim / 2), 256), dim3(256), 0, (hipStream_t)stream, t,
                     (T*)out, B, dim, flip, freq_shift);
  return dm4d_check_launch("timestep_embedding_kernel");
}
extern "C" int dm4d_timestep_embedding_bf16(void* stream, const float* t, void* out, int B, int dim, int flip, float freq_shift) {
  return timestep_embedding_impl<u16>(stream, t, out, B, dim, flip, freq_shift);
}
extern "C" int dm4d_timestep_embedding_f32(void* stream, const float* t, float* out, int B, int dim, int flip, float freq_shift) {
  return timestep_embedding_impl<float>(stream, t, out, B, dim, flip, freq_shift);
}

extern "C" int dm4d_silu_bf16(void* stream, const void* X, void* Y, int64_t n) {
  if (!X || !Y || n <= 0) return dm4d_set_error(DM4D_ERR_ARG, "silu: bad arguments");
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(silu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)X, (u16*)Y, n);
  return dm4d_check_launch("silu_kernel");
}

template <typename T, bool H16 = false>
static int pack_impl(void* stream, void* latents, const void* pv_lat, const void* plucker, const void* skel, const void* mask,
                     const int32_t* is_cond, const int32_t* frame_idx, void* out, int F, int HW, int cpad, int use_cfg) {
  if (!latents || !pv_lat || !plucker || !mask || !is_cond || !out || F <= 0 || HW <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "pack_model_input: null pointer or empty shape");
  if (cpad < 11 + (skel ? 4 : 0)) return dm4d_set_error(DM4D_ERR_ARG, "pack_model_input: cpad too small");
  hipLaunchKernelGGL((pack_kernel<T, H16>), grid1d((int64_t)F * HW, 256), dim3(256), 0, (hipStream_t)stream, (T*)latents,
                     (const T*)pv_lat, (const T*)plucker, (const T*)skel, (const T*)mask, is_cond, frame_idx,
                     (u16*)out, F, HW, cpad, use_cfg);
  return dm4d_check_launch("pack_kernel");
}
extern "C" int dm4d_pack_model_input_bf16(void* stream, void* latents, const void* pv_lat, const void* plucker,
                                          const void* skel, const void* mask, const int32_t* is_cond,
                                          const int32_t* frame_idx, void* out, int F, int HW, int cpad, int use_cfg) {
  return pack_impl<u16>(stream, latents, pv_lat, plucker, skel, mask, is_cond, frame_idx, out, F, HW, cpad, use_cfg);
}
extern "C" int dm4d_pack_model_input_f32_split(void* stream, float* latents, const float* pv_lat, const float* plucker,
                                               const float* skel, const float* mask, const int32_t* is_cond,
                                               const int32_t* frame_idx, void* out, int F, int HW, int cpad, int use_cfg) {
  return pack_impl<float>(stream, latents, pv_lat, plucker, skel, mask, is_cond, frame_idx, out, F, HW, cpad, use_cfg);
}
extern "C" int dm4d_pack_model_input_f32_f16(void* stream, float* latents, const float* pv_lat, const float* plucker,
                                             const float* skel, const float* mask, const int32_t* is_cond,
                                             const int32_t* frame_idx, void* out, int F, int HW, int cpad, int use_cfg) {
  return pack_impl<float, true>(stream, latents, pv_lat, plucker, skel, mask, is_cond, frame_idx, out, F, HW, cpad, use_cfg);
}

template <typename T>
static int cfg_ddim_impl(void* stream, void* latents, const void* noise_pred, int64_t ldn, const float* coef,
                         const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg, float guidance_scale,
                         int v_prediction) {
  if (!latents || !noise_pred || !coef || !is_cond || F <= 0 || HW <= 0 || ldn < 4)
    return dm4d_set_error(DM4D_ERR_ARG, "cfg_ddim_step: bad arguments");
  hipLaunchKernelGGL(cfg_ddim_kernel<T>, grid1d((int64_t)F * HW, 256), dim3(256), 0, (hipStream_t)stream, (T*)latents,
                     (const T*)noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale, v_prediction);
  return dm4d_check_launch("cfg_ddim_kernel");
}
extern "C" int dm4d_cfg_ddim_step_bf16(void* stream, void* latents, const void* noise_pred, int64_t ldn,
                                       const float* coef, const int32_t* is_cond, const int32_t* frame_idx, int F,
                                       int HW, int use_cfg, float guidance_scale, int v_prediction) {
  return cfg_ddim_impl<u16>(stream, latents, noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale, v_prediction);
}
extern "C" int dm4d_cfg_ddim_step_f32(void* stream, float* latents, const float* noise_pred, int64_t ldn,
                                      const float* coef, const int32_t* is_cond, const int32_t* frame_idx, int F,
                                      int HW, int use_cfg, float guidance_scale, int v_prediction) {
  return cfg_ddim_impl<float>(stream, latents, noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale, v_prediction);
}

template <typename T>
static int cfg_linear_impl(void* stream, void* latents, void* x0_prev, const void* noise_pred, int64_t ldn, const float* coef,
                           const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg, float guidance_scale) {
  if (!latents || !x0_prev || !noise_pred || !coef || !is_cond || F <= 0 || HW <= 0 || ldn < 4)
    return dm4d_set_error(DM4D_ERR_ARG, "cfg_linear_step: bad arguments");
  hipLaunchKernelGGL(cfg_linear_step_kernel<T>, grid1d((int64_t)F * HW, 256), dim3(256), 0, (hipStream_t)stream, (T*)latents,
                     (T*)x0_prev, (const T*)noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale);
  return dm4d_check_launch("cfg_linear_step_kernel");
}
extern "C" int dm4d_cfg_linear_step_bf16(void* stream, void* latents, void* x0_prev, const void* noise_pred, int64_t ldn,
                                         const float* coef, const int32_t* is_cond, const int32_t* frame_idx, int F, int HW,
                                         int use_cfg, float guidance_scale) {
  return cfg_linear_impl<u16>(stream, latents, x0_prev, noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale);
}
extern "C" int dm4d_cfg_linear_step_f32(void* stream, float* latents, float* x0_prev, const float* noise_pred, int64_t ldn,
                                        const float* coef, const int32_t* is_cond, const int32_t* frame_idx, int F, int HW,
                                        int use_cfg, float guidance_scale) {
  return cfg_linear_impl<float>(stream, latents, x0_prev, noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale);
}

extern "C" int dm4d_nchw_to_nhwc_bf16(void* stream, const void* X, void* Y, int B, int C, int HW, int cpad) {
  if (!X || !Y || B <= 0 || C <= 0 || HW <= 0 || cpad < C) return dm4d_set_error(DM4D_ERR_ARG, "nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid1d((int64_t)B * HW * cpad, 256), dim3(256), 0, (hipStream_t)stream,
                     (const u16*)X, (u16*)Y, B, C, HW, cpad);
  return dm4d_check_launch("nchw_to_nhwc_kernel");
}

template <typename T>
static int nhwc_to_nchw_impl(void* stream, const void* X, void* Y, int B, int C, int HW, int ldx) {
  if (!X || !Y || B <= 0 || C <= 0 || HW <= 0 || ldx < C) return dm4d_set_error(DM4D_ERR_ARG, "nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, grid1d((int64_t)B * C * HW, 256), dim3(256), 0, (hipStream_t)stream,
                     (const T*)X, (T*)Y, B, C, HW, ldx);
  return dm4d_check_launch("nhwc_to_nchw_kernel");
}
extern "C" int dm4d_nhwc_to_nchw_bf16(void* stream, const void* X, void* Y, int B, int C, int HW, int ldx) {
  return nhwc_to_nchw_impl<u16>(stream, X, Y, B, C, HW, ldx);
}
extern "C" int dm4d_nhwc_to_nchw_f32(void* stream, const float* X, float* Y, int B, int C, int HW, int ldx) {
  return nhwc_to_nchw_impl<float>(stream, X, Y, B, C, HW, ldx);
}

template <typename T>
static int vae_sample_impl(void* stream, const void* moments, int64_t ldm, const void* noise, void* out, int64_t M, int C, float scale) {
  if (!moments || !noise || !out || M <= 0 || C <= 0 || ldm < 2 * C) return dm4d_set_error(DM4D_ERR_ARG, "vae_sample: bad arguments");
  hipLaunchKernelGGL(vae_sample_kernel<T>, grid1d(M * C, 256), dim3(256), 0, (hipStream_t)stream, (const T*)moments, ldm,
                     (const T*)noise, (T*)out, M, C, scale);
  return dm4d_check_launch("vae_sample_kernel");
}
extern "C" int dm4d_vae_sample_bf16(void* stream, const void* moments, int64_t ldm, const void* noise, void* out, int64_t M,
                                    int C, float scale) {
  return vae_sample_impl<u16>(stream, moments, ldm, noise, out, M, C, scale);
}
extern "C" int dm4d_vae_sample_f32(void* stream, const float* moments, int64_t ldm, const float* noise, float* out, int64_t M,
                                   int C, float scale) {
  return vae_sample_impl<float>(stream, moments, ldm, noise, out, M, C, scale);
}

extern "C" int dm4d_scale_pad_bf16(void* stream, const void* X, int64_t ldx, void* Y, int cpad, int64_t M, int C, float scale) {
  if (!X || !Y || M <= 0 || C <= 0 || cpad < C || ldx < C) return dm4d_set_error(DM4D_ERR_ARG, "scale_pad: bad arguments");
  hipLaunchKernelGGL(scale_pad_kernel, grid1d(M * cpad, 256), dim3(256), 0, (hipStream_t)stream, (const u16*)X, ldx,
                     (u16*)Y, cpad, M, C, scale);
  return dm4d_check_launch("scale_pad_kernel");
}

template <typename T>
static int resize_impl(void* stream, const float* X, void* Y, int B, int C, int H, int W, int h, int w, int bilinear) {
  if (!X || !Y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0)
    return dm4d_set_error(DM4D_ERR_ARG, "resize: bad arguments");
  hipLaunchKernelGGL(resize_kernel<T>, grid1d((int64_t)B * h * w * C, 256), dim3(256), 0, (hipStream_t)stream, X, (T*)Y, B,
                     C, H, W, h, w, bilinear);
  return dm4d_check_launch("resize_kernel");
}
extern "C" int dm4d_resize_nchw_f32_to_nhwc_bf16(void* stream, const float* X, void* Y, int B, int C, int H, int W, int h,
                                                 int w, int bilinear) {
  return resize_impl<u16>(stream, X, Y, B, C, H, W, h, w, bilinear);
}
extern "C" int dm4d_resize_nchw_f32_to_nhwc_f32(void* stream, const float* X, float* Y, int B, int C, int H, int W, int h,
                                                int w, int bilinear) {
  return resize_impl<float>(stream, X, Y, B, C, H, W, h, w, bilinear);
}

template <typename T>
static int postprocess_impl(void* stream, const void* X, void* Y, int B, int C, int HW, int ldx) {
  if (!X || !Y || B <= 0 || C <= 0 || HW <= 0 || ldx < C) return dm4d_set_error(DM4D_ERR_ARG, "postprocess: bad arguments");
  hipLaunchKernelGGL(postprocess_kernel<T>, grid1d((int64_t)B * C * HW, 256), dim3(256), 0, (hipStream_t)stream,
                     (const T*)X, (T*)Y, B, C, HW, ldx);
  return dm4d_check_launch("postprocess_kernel");
}
extern "C" int dm4d_postprocess_images_bf16(void* stream, const void* X, void* Y, int B, int C, int HW, int ldx) {
  return postprocess_impl<u16>(stream, X, Y, B, C, HW, ldx);
}
extern "C" int dm4d_postprocess_images_f32(void* stream, const float* X, float* Y, int B, int C, int HW, int ldx) {
  return postprocess_impl<float>(stream, X, Y, B, C, HW, ldx);
}

template <typename T>
static int plucker_impl(void* stream, const float* cams, void* Y, int N, int H, int W, int h, int w) {
  if (!cams || !Y || N <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return dm4d_set_error(DM4D_ERR_ARG, "plucker: bad arguments");
  hipLaunchKernelGGL(plucker_latent_kernel<T>, grid1d((int64_t)N * h * w, 256), dim3(256), 0, (hipStream_t)stream, cams, (T*)Y, N,
                     H, W, h, w);
  return dm4d_check_launch("plucker_latent_kernel");
}
extern "C" int dm4d_plucker_latent_bf16(void* stream, const float* cams, void* Y, int N, int H, int W, int h, int w) {
  return plucker_impl<u16>(stream, cams, Y, N, H, W, h, w);
}
extern "C" int dm4d_plucker_latent_f32(void* stream, const float* cams, float* Y, int N, int H, int W, int h, int w) {
  return plucker_impl<float>(stream, cams, Y, N, H, W, h, w);
}

extern "C" int dm4d_resize_aa_nchw_f32(void* stream, const float* X, float* Y, int64_t planes, int H, int W, int h, int w) {
  if (!X || !Y || planes <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return dm4d_set_error(DM4D_ERR_ARG, "resize_aa: bad arguments");
  hipLaunchKernelGGL(resize_aa_kernel, grid1d(planes * h * w, 256), dim3(256), 0, (hipStream_t)stream, X, Y, planes, H, W, h, w);
  return dm4d_check_launch("resize_aa_kernel");
}

template <typename T>
static int cfg_multistep_impl(void* stream, void* latents, void* s1, void* s2, void* s3, const void* noise_pred, int64_t ldn, const float* coef,
                              const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg, float guidance_scale) {
  if (!latents || !s1 || !noise_pred || !coef || !is_cond || F <= 0 || HW <= 0 || ldn < 4)
    return dm4d_set_error(DM4D_ERR_ARG, "cfg_multistep_step: bad arguments");
  hipLaunchKernelGGL(cfg_multistep_kernel<T>, grid1d((int64_t)F * HW, 256), dim3(256), 0, (hipStream_t)stream, (T*)latents, (T*)s1, (T*)s2,
                     (T*)s3, (const T*)noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale);
  return dm4d_check_launch("cfg_multistep_kernel");
}
extern "C" int dm4d_cfg_multistep_step_bf16(void* stream, void* latents, void* s1, void* s2, void* s3, const void* noise_pred, int64_t ldn,
                                            const float* coef, const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg,
                                            float guidance_scale) {
  return cfg_multistep_impl<u16>(stream, latents, s1, s2, s3, noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale);
}
extern "C" int dm4d_cfg_multistep_step_f32(void* stream, float* latents, float* s1, float* s2, float* s3, const float* noise_pred, int64_t ldn,
                                           const float* coef, const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg,
                                           float guidance_scale) {
  return cfg_multistep_impl<float>(stream, latents, s1, s2, s3, noise_pred, ldn, coef, is_cond, frame_idx, F, HW, use_cfg, guidance_scale);
}
