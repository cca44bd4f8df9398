// Tuning translation unit of tools/conv64/cab.py: conv64_kernel alone behind one C entry (the product compiles it inside gemm.hip).
#include "gemm_common.h"
namespace {
#include "conv64.h"
}
extern "C" int dm4d_conv64_test(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y, int Cout, const void* bias) {
  GemmParams p{};
  p.A = (const u16*)X; p.H = H; p.W = W; p.Cin = Cin; p.Ho = H; p.Wo = W; p.stride = 1; p.pad = 1; p.upsample = 0;
  p.Wt = (const u16*)Wt; p.ldw = (int64_t)9 * Cin; p.C = (u16*)Y; p.ldc = Cout;
  p.M = B * H * W; p.N = Cout; p.K = 9 * Cin;
  p.bias = (const u16*)bias; p.rows_per_rb = H * W; p.flags = 0; p.out_scale = 1.0f; p.splits = 1;
  if (Cout % 128 == 0) return launch_conv64<128, 2, 2, 0>((hipStream_t)stream, p);
  return launch_conv64<160, 4, 1, 0>((hipStream_t)stream, p);
}
#ifdef CONV64_TIMING
extern "C" void dm4d_conv64_set_debug(void* stream, void* ptr) {
  hipLaunchKernelGGL(conv64_set_dbg, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)ptr);
}
#endif
