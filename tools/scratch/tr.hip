#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void trtest(const int* addr_bytes, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (unsigned short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4* lp;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)((__attribute__((address_space(3))) char*)sm + addr_bytes[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
