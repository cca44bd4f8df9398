// Probe: do MFMA and VALU instructions of two waves that share a SIMD overlap, and what do plain / packed /
// transcendental VALU instructions cost on gfx950?  One workgroup of 8 waves per CU (waves w and w+4 share a SIMD);
// waves 0-3 run op A, waves 4-7 run op B, each a loop of independent instructions.  Prints microseconds per
// configuration; "A|B together" ~ max(A, B) means the pipes overlap, ~ A + B means they serialise.
//   hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip && ./issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { NONE = 0, MFMA = 1, FMA = 2, PKFMA = 3, EXP = 4, PKADD = 5, CVT = 6, MAX3 = 7, ADD = 8, MOV = 9, MAX2 = 10, DOT2 = 11, DOT2C = 12, EXPH = 13, DSREAD = 14, DSTR = 15 };

template <int OP>
__device__ __forceinline__ float run_op(int iters, float seed) {
  if constexpr (OP == MFMA) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = seed;
    bf16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (__bf16)seed; b[r] = (__bf16)(seed + 1.f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    return acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else if constexpr (OP == FMA) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(seed));
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i]; return s;
  } else if constexpr (OP == PKFMA) {
    f32x2 x[16];
    f32x2 c = {seed, seed};
    for (int i = 0; i < 16; ++i) x[i] = f32x2{seed + i, seed - i};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(c));
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i][0] + x[i][1]; return s;
  } else if constexpr (OP == PKADD) {
    f32x2 x[16];
    f32x2 c = {seed, seed};
    for (int i = 0; i < 16; ++i) x[i] = f32x2{seed + i, seed - i};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i][0] + x[i][1]; return s;
  } else if constexpr (OP == EXP) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed * 0.001f + i * 0.01f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i]; return s;
  } else if constexpr (OP == CVT) {
    float x[16]; uint32_t y[16];
    for (int i = 0; i < 16; ++i) { x[i] = seed + i; y[i] = 0; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(y[i]) : "v"(x[i]));
    uint32_t s = 0; for (int i = 0; i < 16; ++i) s += y[i]; return (float)s;
  } else if constexpr (OP == MAX3) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(seed));
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i]; return s;
  } else if constexpr (OP == ADD || OP == MOV || OP == MAX2 || OP == DOT2 || OP == DOT2C || OP == EXPH) {
    float x[16];
    const float y = seed * 0.5f;
    for (int i = 0; i < 16; ++i) x[i] = seed * 0.001f + i * 0.01f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if constexpr (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(y));
        if constexpr (OP == MAX2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        if constexpr (OP == DOT2) asm volatile("v_dot2_f32_bf16 %0, %1, %1, %0" : "+v"(x[i]) : "v"(y));
        if constexpr (OP == DOT2C) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(x[i]) : "v"(y));
        if constexpr (OP == EXPH) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
      }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i]; return s;
  } else if constexpr (OP == DSREAD || OP == DSTR) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 4096; i += 64) lds[(threadIdx.x >> 6) * 4096 + i] = seed;
    const unsigned base = (unsigned)(size_t)(lds + (threadIdx.x >> 6) * 4096) + lane * (OP == DSREAD ? 16 : 8);
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      f32x4 v[16];
      s16x4 w[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if constexpr (OP == DSREAD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(base), "n"(i * 1024));
        else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(w[i]) : "v"(base), "n"(i * 512));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if constexpr (OP == DSREAD) asm volatile("" :: "v"(v[i])); else asm volatile("" :: "v"(w[i]));
      }
    }
    return acc[0];
  } else {
    return seed;
  }
}

// one wave per SIMD; per loop iteration 4 x { 1 MFMA ; KV independent VALU instructions (v_fma or v_exp) }
template <int KV, bool EXPOP>
__global__ __launch_bounds__(256) void mix_probe(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = seed;
  bf16x8 a, b;
  for (int r = 0; r < 8; ++r) { a[r] = (__bf16)seed; b[r] = (__bf16)(seed + 1.f); }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed * 0.001f + i * 0.01f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        if constexpr (EXPOP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[k & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[k & 7]) : "v"(seed));
      }
    }
  }
  float r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  for (int i = 0; i < 8; ++i) r += x[i];
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KV, bool EXPOP>
float time_mix(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mix_probe<KV, EXPOP>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((mix_probe<KV, EXPOP>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}

template <int A, int B>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  float r;
  if (wave < 4) r = run_op<A>(iters, seed); else r = run_op<B>(iters, seed);
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int A, int B>
float time_it(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<A, B>), dim3(256), dim3(512), 8 * 4096 * 4, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<A, B>), dim3(256), dim3(512), 8 * 4096 * 4, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  const int it = 20000;  // 16 instructions per iteration per wave
  const double n = 16.0 * it;
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("device %s, %d CUs; waves 0-3 of a 512-thread workgroup run op A, waves 4-7 op B (w and w+4 share a SIMD)\n", pr.name, pr.multiProcessorCount);
  // the clock is calibrated on the MFMA stream itself: one wave per SIMD issuing independent 32x32x16 MFMAs back to
  // back retires one per 32 cycles (MI355X_MICROARCH.md), which also keeps the part at its loaded clock
  time_it<MFMA, MFMA>(out, it);
  const double cyc_us = 32.0 * n / time_it<MFMA, NONE>(out, it);
  printf("calibrated clock %.0f MHz\n", cyc_us);
#define ONE(name, A) { float us = time_it<A, A>(out, it); printf("%-22s two waves / SIMD  %9.1f us  %6.2f cycles/instr per SIMD\n", name, us, us * cyc_us / (2 * n)); }
  ONE("mfma 32x32x16 bf16", MFMA) ONE("v_fma_f32", FMA) ONE("v_pk_fma_f32", PKFMA) ONE("v_pk_add_f32", PKADD)
  ONE("v_exp_f32", EXP) ONE("v_cvt_pk_bf16_f32", CVT) ONE("v_max3_f32", MAX3) ONE("v_add_f32", ADD) ONE("v_mov_b32", MOV)
  ONE("ds_read_b128", DSREAD) ONE("ds_read_b64_tr_b16", DSTR)
  ONE("v_max_f32", MAX2) ONE("v_dot2_f32_bf16", DOT2) ONE("v_dot2c_f32_bf16", DOT2C) ONE("v_exp_f16", EXPH)
#define TWO(name, A, B) { float us = time_it<A, B>(out, it); printf("%-22s one wave each     %9.1f us  %6.2f cycles per (A,B) instruction pair per SIMD\n", name, us, us * cyc_us / n); }
  TWO("mfma | none", MFMA, NONE) TWO("mfma | v_fma", MFMA, FMA) TWO("mfma | v_exp", MFMA, EXP) TWO("mfma | v_pk_fma", MFMA, PKFMA)
  TWO("mfma | v_cvt_pk", MFMA, CVT) TWO("mfma | v_max3", MFMA, MAX3) TWO("v_fma | v_exp", FMA, EXP)
  TWO("mfma | v_add", MFMA, ADD) TWO("mfma | v_mov", MFMA, MOV) TWO("mfma | v_max", MFMA, MAX2) TWO("mfma | v_dot2", MFMA, DOT2)
  TWO("mfma | ds_read_b128", MFMA, DSREAD) TWO("mfma | ds_read_tr", MFMA, DSTR)
  TWO("mfma | v_dot2c", MFMA, DOT2C) TWO("mfma | v_exp_f16", MFMA, EXPH) TWO("mfma | v_pk_add", MFMA, PKADD)
  // VALU work issued by the SAME wave between its MFMAs ("MFMA shadow"): cycles per {1 MFMA + KV VALU}
#define MIX(KV) { float f = time_mix<KV, false>(out, it * 4); float e = time_mix<KV, true>(out, it * 4); \
    printf("same wave: 1 mfma + %2d v_fma  %6.2f cycles   | 1 mfma + %2d v_exp  %6.2f cycles\n", KV, f * cyc_us / n, KV, e * cyc_us / n); }
  MIX(0) MIX(2) MIX(4) MIX(6) MIX(8) MIX(12) MIX(16)
  return 0;
}
