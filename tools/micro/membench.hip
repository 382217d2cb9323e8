// Micro-benchmark: per-CU global store / load throughput of one 512-thread workgroup per CU (the shape of the fused chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void wr(u32x4* p, long per_wg16, int active_lanes) {
  u32x4* base = p + (long)blockIdx.x * per_wg16;
  const int lane = threadIdx.x & 63;
  if (lane >= active_lanes) return;
  u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  for (long i = threadIdx.x; i < per_wg16; i += 512) base[i] = v;
}
__global__ __launch_bounds__(512) void rd(const u32x4* p, long per_wg16, u32x4* out) {
  const u32x4* base = p + (long)blockIdx.x * per_wg16;
  u32x4 a = {0, 0, 0, 0};
#pragma unroll 4
  for (long i = threadIdx.x; i < per_wg16; i += 512) a ^= __builtin_nontemporal_load(base + i);
  if (a.x == 0x12345) out[0] = a;
}
__global__ __launch_bounds__(512) void rw(const u32x4* p, u32x4* q, long per_wg16) {
  const u32x4* base = p + (long)blockIdx.x * per_wg16;
  u32x4* ob = q + (long)blockIdx.x * per_wg16;
#pragma unroll 4
  for (long i = threadIdx.x; i < per_wg16; i += 512) ob[i] = base[i] + 1u;
}
int main() {
  const long maxb = 1L << 30;
  u32x4 *a, *b;
  hipMalloc(&a, maxb); hipMalloc(&b, maxb);
  hipMemset(a, 1, maxb); hipMemset(b, 1, maxb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grids[] = {1, 32, 160, 256, 512, 2048};
  long sizes[] = {256 << 10, 1 << 20};
  for (long sz : sizes) for (int g : grids) {
    if ((long)g * sz > maxb) continue;
    const long n16 = sz / 16;
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) wr<<<g, 512>>>(a, n16, 64);
        else if (mode == 1) wr<<<g, 512>>>(a, n16, 48);
        else if (mode == 2) rd<<<g, 512>>>(a, n16, b);
        else rw<<<g, 512>>>(a, b, n16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const char* nm[] = {"write", "write48", "read", "copy"};
      double bytes = (double)g * sz * (mode == 3 ? 2 : 1) * (mode == 1 ? 0.75 : 1.0);
      printf("%-8s grid %5d x %5ld KB: %8.1f us  %7.1f GB/s  %6.2f B/clk/WG(2.4GHz)\n", nm[mode], g, sz >> 10, best * 1e3, bytes / best * 1e-6,
             bytes / g / (best * 1e-3 * 2.4e9));
    }
  }
  return 0;
}
