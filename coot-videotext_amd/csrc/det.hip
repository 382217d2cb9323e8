// Deterministic accumulation: configuration and flush (det.h).
#include "det.h"

#include "../../include/coot_hip.h"

namespace coot {
namespace {
DetTable g_host_table = {0, {}, nullptr};

__global__ __launch_bounds__(256) void det_flush_kernel(float* p, long long* shadow, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long long v = shadow[i];
    if (v != 0) {
      shadow[i] = 0;
      p[i] += (float)((double)v * (1.0 / kDetScale));
    }
  }
}
}  // namespace

bool det_on() { return g_host_table.n > 0; }

// addends that bypassed the fixed-point shadow since coot_det_configure (NaN / Inf / |v| >= 2^22); synchronises the device; -1: mode off
int det_bypass_count() {
  if (g_host_table.n <= 0 || !g_host_table.bypass) return -1;
  unsigned v = 0;
  if (check_hip(hipDeviceSynchronize(), "det bypass count") || check_hip(hipMemcpy(&v, g_host_table.bypass, 4, hipMemcpyDeviceToHost), "det bypass count")) return -1;
  return (int)(v > 0x7fffffffu ? 0x7fffffffu : v);
}

int det_flush_range(const void* base, size_t bytes, hipStream_t st) {
  const char* b0 = (const char*)base;
  for (int i = 0; i < g_host_table.n; ++i) {
    const DetRange& R = g_host_table.r[i];
    const char* lo = b0 > R.base ? b0 : R.base;
    const char* hi = (b0 + bytes) < (R.base + R.bytes) ? (b0 + bytes) : (R.base + R.bytes);
    if (lo >= hi) continue;
    const long n = (hi - lo) / 4, first = (lo - R.base) / 4;
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(det_flush_kernel, dim3(blocks), dim3(256), 0, st, (float*)const_cast<char*>(lo), R.shadow + first, n);
    COOT_CHECK_LAUNCH("det_flush");
  }
  return 0;
}

}  // namespace coot

using namespace coot;

extern "C" {

size_t coot_det_shadow_bytes(int n, const size_t* bytes) {
  size_t tot = 0;
  for (int i = 0; i < n; ++i) tot += ((bytes[i] + 3) / 4) * 8 + 256;
  return tot;
}

int coot_det_configure(int n, void* const* bases, const size_t* bytes, void* shadow, size_t shadow_bytes, coot_stream_t stream) {
  COOT_REQUIRE(n >= 0 && n <= 8, "det_configure: at most 8 ranges (%d)", n);
  DetTable t = {0, {}, nullptr};
  if (n > 0) {
    COOT_REQUIRE(bases && bytes && shadow && shadow_bytes >= coot_det_shadow_bytes(n, bytes), "det_configure: shadow too small (%zu < %zu)",
                 shadow_bytes, n > 0 && bytes ? coot_det_shadow_bytes(n, bytes) : (size_t)0);
    char* s = (char*)shadow;
    for (int i = 0; i < n; ++i) {
      COOT_REQUIRE(bases[i] && ((size_t)bases[i] & 3) == 0, "det_configure: range %d", i);
      t.r[i].base = (const char*)bases[i]; t.r[i].bytes = bytes[i] & ~(size_t)3; t.r[i].shadow = (long long*)s;
      s += ((bytes[i] + 3) / 4) * 8 + 256;
      if (i == 0) t.bypass = (unsigned*)(s - 256);  // the 256 padding bytes behind range 0's accumulators (zeroed below, never flushed)
    }
    t.n = n;
    if (check_hip(hipMemsetAsync(shadow, 0, shadow_bytes, (hipStream_t)stream), "det shadow")) return -1;
    if (check_hip(hipStreamSynchronize((hipStream_t)stream), "det shadow")) return -1;
  }
  // (hipMemcpyToSymbol synchronises with the device: no kernel of any stream sees half a table)
  if (det_set_table_fused(t) || det_set_table_gemm(t) || det_set_table_loss(t) || det_set_table_pool(t) || det_set_table_rowops(t)) return -1;
  g_host_table = t;
  return 0;
}

int coot_det_flush(const void* base, size_t bytes, coot_stream_t stream) {
  COOT_REQUIRE(base || bytes == 0, "det_flush: null range");
  return det_flush_range(base, bytes, (hipStream_t)stream);
}

}  // extern "C"
