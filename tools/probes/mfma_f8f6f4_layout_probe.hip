#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A[i][k], B[k][j] as fp8 e4m3; we give each lane 32 bytes for A and B and discover which (i,k) / (k,j) they are:
// set A bytes = 1.0 (0x38) only at one (lane, byte) position and B = all ones -> D[i][*] = 1 reveals row i; similarly for k via B selective.
__global__ void probe(const uint8_t* a_bytes, const uint8_t* b_bytes, float* out) {
  int lane = threadIdx.x;
  v8i a, b;
  for (int r = 0; r < 8; ++r) { a[r] = ((const int*)a_bytes)[lane * 8 + r]; b[r] = ((const int*)b_bytes)[lane * 8 + r]; }
  f32x16 c = {0};
  // cbsz = 0 (A fp8 e4m3), blgp = 0 (B fp8 e4m3); scales = 127 (2^0) in e8m0
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
int main() {
  uint8_t *da, *db; float* dout;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dout, 64 * 16 * 4);
  uint8_t ha[64 * 32], hb[64 * 32]; float ho[64 * 16];
  // Test 1: B all ones; A one-hot at (lane la, byte ba): which output rows light up, and with what value (1 => that k slot exists)
  int tests[][2] = {{0, 0}, {0, 1}, {0, 15}, {0, 16}, {0, 31}, {1, 0}, {31, 5}, {32, 0}, {32, 31}, {63, 17}};
  for (auto& t : tests) {
    for (int i = 0; i < 64 * 32; ++i) { ha[i] = 0; hb[i] = 0x38; }
    ha[t[0] * 32 + t[1]] = 0x38;
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout); hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    // D layout: lane -> col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    int rows[32] = {0}; float val = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) if (ho[l * 16 + r] != 0) { rows[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)]++; val = ho[l * 16 + r]; }
    printf("A one-hot lane %2d byte %2d -> rows:", t[0], t[1]); for (int i = 0; i < 32; ++i) if (rows[i]) printf(" %d(x%d)", i, rows[i]); printf("  val %.2f\n", val);
  }
  // Test 2: A one-hot (lane 0, byte ba) and B one-hot (lane lb, byte bb): nonzero iff same k -> find k mapping of B relative to A
  int ab[] = {0, 5, 16, 31};
  for (int ba : ab) for (int la : {0, 32}) {
    printf("A(lane %d, byte %d) matches B at:", la, ba);
    for (int lb : {0, 32}) for (int bb = 0; bb < 32; ++bb) {
      for (int i = 0; i < 64 * 32; ++i) { ha[i] = 0; hb[i] = 0; }
      ha[la * 32 + ba] = 0x38; hb[lb * 32 + bb] = 0x38;
      hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout); hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
      float s = 0; for (int i = 0; i < 64 * 16; ++i) s += ho[i];
      if (s != 0) printf(" (lane %d, byte %d)", lb, bb);
    }
    printf("\n");
  }
  return 0;
}
