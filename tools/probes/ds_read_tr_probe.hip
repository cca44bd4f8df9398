#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void trtest(const int* addr_bytes, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (unsigned short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4* lp;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)((__attribute__((address_space(3))) char*)sm + addr_bytes[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int *d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int test = 0; test < 2; ++test) {
    // test 0: lane l -> row (l&15)>>2 of a [4][16] block (row stride 16 elems = 32 B), chunk (l&3); group g=l>>4 -> block base g*64 elems
    // test 1: row stride 72 elements (144 B), block for group g at d0 = 16*(g&1), key0 = 4*(g>>1)
    for (int l = 0; l < 64; ++l) {
      int i = l & 15, g = l >> 4;
      if (test == 0) h_addr[l] = 2 * (g * 64 + (i >> 2) * 16 + (i & 3) * 4);
      else h_addr[l] = 2 * ((4 * (g >> 1) + (i >> 2)) * 72 + 16 * (g & 1) + (i & 3) * 4);
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(trtest, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("test %d\n", test);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr_elem %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
  }
  return 0;
}
