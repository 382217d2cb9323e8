// Micro-benchmark: the store pattern of an accumulator-layout epilogue (8 bytes per lane: 16 rows x 32 contiguous bytes
// per wave instruction, rows 768 B apart) against row-contiguous 16-byte stores, one 512-thread workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// tile = 128 rows x 768 bytes; wave w owns byte columns [96 w, 96 w + 96); fragments (a = 16-row block, b = 32-byte block)
__global__ __launch_bounds__(512) void wr_frag(char* p, int tiles_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x2 v = {threadIdx.x, blockIdx.x};
  for (int t = 0; t < tiles_per_wg; ++t) {
    char* base = p + ((long)blockIdx.x * tiles_per_wg + t) * (128 * 768);
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        *reinterpret_cast<u32x2*>(base + (a * 16 + (lane & 15)) * 768 + wave * 96 + b * 32 + (lane >> 4) * 8) = v;
  }
}
__global__ __launch_bounds__(512) void wr_rows(char* p, int tiles_per_wg) {
  u32x4 v = {threadIdx.x, blockIdx.x, 1u, 2u};
  for (int t = 0; t < tiles_per_wg; ++t) {
    char* base = p + ((long)blockIdx.x * tiles_per_wg + t) * (128 * 768);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int c = threadIdx.x + 512 * i;  // 48 chunks of 16 B per row
      *reinterpret_cast<u32x4*>(base + (c / 48) * 768 + (c % 48) * 16) = v;
    }
  }
}
__global__ __launch_bounds__(512) void rd_frag(const char* p, int tiles_per_wg, u32x2* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x2 acc = {0, 0};
  for (int t = 0; t < tiles_per_wg; ++t) {
    const char* base = p + ((long)blockIdx.x * tiles_per_wg + t) * (128 * 768);
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        acc ^= *reinterpret_cast<const u32x2*>(base + (a * 16 + (lane & 15)) * 768 + wave * 96 + b * 32 + (lane >> 4) * 8);
  }
  if (acc.x == 0x1234567) out[0] = acc;
}
int main() {
  const long maxb = 1L << 30;
  char* a; u32x2* o;
  (void)hipMalloc(&a, maxb); (void)hipMalloc(&o, 64); (void)hipMemset(a, 1, maxb);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int grids[] = {1, 160, 256};
  for (int g : grids) for (int tiles : {2, 10}) for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0);
      if (mode == 0) wr_rows<<<g, 512>>>(a, tiles);
      else if (mode == 1) wr_frag<<<g, 512>>>(a, tiles);
      else rd_frag<<<g, 512>>>(a, tiles, o);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const char* nm[] = {"write rows16", "write frag8 ", "read  frag8 "};
    double bytes = (double)g * tiles * 128 * 768;
    printf("%s grid %4d x %2d tiles: %8.1f us  %7.1f GB/s  %6.2f B/clk/WG\n", nm[mode], g, tiles, best * 1e3, bytes / best * 1e-6, bytes / g / (best * 1e-3 * 2.4e9));
  }
  return 0;
}
