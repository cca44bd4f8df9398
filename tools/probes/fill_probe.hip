// Probe: how many bytes per cycle can ONE CU pull from L2 / Infinity Cache / HBM into LDS (global_load_lds_dwordx4) or
// into registers (global_load_dwordx4), as a function of (a) waves per CU, (b) loads kept in flight per wave, (c) the
// footprint (per-XCD-L2 resident, Infinity-Cache resident, HBM), (d) the shape of one wave-instruction's 1 KiB: one
// contiguous KiB vs 8 rows x 128 B vs 16 rows x 64 B at a 640-byte row stride (what the GEMM / conv kernels of
// libdm4d.so issue), and (e) with every CU reading the SAME bytes at the same time (weight tiles) vs its own.
// The GEMM / conv kernels of this repo all land on 10-18 B/cycle/CU of L2->LDS traffic whatever else is changed
// (DESIGN.md section 4); this measures the ceiling of that path and what moves it.
//   hipcc --offload-arch=gfx950 -O3 -o fill_probe fill_probe.hip && ./fill_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void dma16(const void* ptr, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(ptr), "s"(lds_dst)
               : "memory");
}

struct Args {
  const char* src;
  uint64_t region;     // bytes each XCD-group of workgroups walks through (power of two)
  int iters;           // DMA instructions per wave
  int rows, rowbytes;  // one wave-instruction = rows x rowbytes (rows * rowbytes == 1024), row stride = stride
  int stride;
  int shared;          // 1: every workgroup reads the same addresses (a weight tile); 0: its own slice
  unsigned* sink;
};

// DEPTH loads in flight per wave (counted vmcnt), LDS = NW waves x DEPTH slots x 1 KiB
template <int NW, int DEPTH, bool TO_LDS>
__global__ __launch_bounds__(NW * 64) void fill_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) char lds[NW * DEPTH * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + wave * DEPTH * 1024;
  const int lanes_per_row = a.rowbytes / 16;
  const int r = lane / lanes_per_row, c = lane % lanes_per_row;
  const uint64_t lane_off = (uint64_t)r * a.stride + (uint64_t)c * 16;
  const uint64_t tile_bytes = (uint64_t)a.rows * a.stride;  // address span of one instruction
  // workgroup b of the launch runs on XCD b % 8 (observed placement; used for footprint control only)
  const uint64_t wg = a.shared ? 0 : blockIdx.x;
  uint64_t pos = (wg * NW + wave) * tile_bytes;
  const uint64_t step = (a.shared ? (uint64_t)NW : (uint64_t)gridDim.x * NW) * tile_bytes;
  uint4 acc = {0, 0, 0, 0};
  for (int it = 0; it < a.iters; ++it) {
    const char* p = a.src + ((pos + lane_off) & (a.region - 1));
    if constexpr (TO_LDS) {
      dma16(p, lds_base + (it % DEPTH) * 1024);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    } else {  // register destination: load + wait in ONE statement (an un-waited asm load's destination may be reused)
      uint4 v;
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
      acc.x ^= v.x;
      acc.y ^= v.y;
    }
    pos += step;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) a.sink[threadIdx.x] = acc.x;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); fflush(stdout); exit(1); } } while (0)

template <int NW, int DEPTH, bool TO_LDS>
double run(Args a, int wgs_per_cu, double clock_mhz) {
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL((fill_kernel<NW, DEPTH, TO_LDS>), dim3(grid), dim3(NW * 64), 0, 0, a);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill_kernel<NW, DEPTH, TO_LDS>), dim3(grid), dim3(NW * 64), 0, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 3.0 * grid * NW * (double)a.iters * 1024.0;
  const double cycles = ms * 1e-3 * clock_mhz * 1e6;
  return bytes / cycles / 256.0;  // bytes per cycle per CU
}

int main() {
  const uint64_t cap = 1ull << 30;
  char* buf = nullptr;
  CHECK(hipMalloc((void**)&buf, cap));
  CHECK(hipMemset(buf, 1, cap));
  unsigned* sink = nullptr;
  CHECK(hipMalloc((void**)&sink, 4096));
  CHECK(hipDeviceSynchronize());
  printf("buffer %p, sink %p\n", (void*)buf, (void*)sink);
  fflush(stdout);
  const double clk = 2400.0;  // nominal; B/cycle figures scale with the real clock (DVFS), TB/s printed beside
  printf("bytes/cycle/CU at a nominal 2.4 GHz  (x 256 CUs x 2.4e9 = chip bytes/s; 10 B/cycle/CU = 6.1 TB/s)\n");
  struct Shape { const char* name; int rows, rowbytes, stride; } shapes[] = {
      {"1x1024 contiguous", 1, 1024, 1024}, {"8 rows x 128 B, stride 640", 8, 128, 640}, {"16 rows x 64 B, stride 640", 16, 64, 640},
      {"8 rows x 128 B, stride 2560", 8, 128, 2560}, {"16 rows x 64 B, stride 2560", 16, 64, 2560}};
  struct Foot { const char* name; uint64_t region; int shared; } foots[] = {
      {"own slice, 2 MB footprint (resident in every XCD's L2)", 2ull << 20, 0},
      {"own slice, 16 MB footprint (4x an XCD's L2)", 16ull << 20, 0},
      {"own slice, 128 MB footprint (Infinity Cache)", 128ull << 20, 0},
      {"own slice, 1 GB footprint (HBM)", 1ull << 30, 0},
      {"ALL workgroups read the same 256 KB (weight tile)", 256ull << 10, 1}};
  for (auto& f : foots) {
    printf("--- %s\n", f.name);
    for (auto& s : shapes) {
      Args a{buf, f.region, 4096, s.rows, s.rowbytes, s.stride, f.shared, sink};
#define ROW(NW, D, L, W) { double b = run<NW, D, L>(a, W, clk); printf("   %-28s %s waves/CU=%2d in-flight/wave=%d : %6.2f B/cyc/CU  (%5.2f TB/s)\n", s.name, L ? "->LDS" : "->VGPR", NW * W, D, b, b * 256 * 2.4e-3); fflush(stdout); }
      ROW(4, 4, true, 1) ROW(8, 4, true, 1) ROW(16, 4, true, 1) ROW(8, 4, true, 2) ROW(16, 4, true, 2)
      ROW(8, 2, true, 1) ROW(8, 8, true, 1) ROW(16, 8, true, 1) ROW(16, 2, true, 1)
    }
  }
  return 0;
}
