// What slows a workgroup down when more of its kind share an XCD?  (Round 5: the fused chains' tile latency follows the tiles per XCD —
// profiles/r05_chain_probe_xcd.txt — in EVERY phase, also in phases that touch no global memory.)  One 512-thread workgroup per CU (150 KB
// of dynamic LDS, as the chain kernels), fixed work per workgroup, G workgroups (G / 8 per XCD: workgroup b runs on XCD b % 8):
//   mode 0  VALU only: 8 independent v_fma_f32 chains per lane
//   mode 1  LDS only: ds_read_b128 / ds_write_b128 rounds over a 96 KB tile (conflict free)
//   mode 2  L2 stream: every workgroup reads the same 2.4 MB (the weights of a chain pass) with 16-byte loads, 3 in flight per wave
//   mode 3  straight-line VALU: the mode-0 work as ~48 KB of unrolled code (instruction fetch)
//   mode 6 / 7  the mode-3 code executed ONCE per launch, the two (identical, separately compiled) kernels launched alternately so
//           that each launch finds the instruction cache (64 KB per CU pair) holding the other one: cold straight-line code
//   mode 4  MFMA only: 12 independent v_mfma_f32_16x16x32_bf16 accumulators per wave, back to back
//   mode 5  a chain-like mix per repetition: an MFMA pass (96 MFMAs per wave), then an "epilogue" of 1 536 VALU ops and 48 LDS round trips
// Reported per G: median / max workgroup time in shader clocks (s_memtime) and in us (100 MHz counter).
//   hipcc --offload-arch=gfx950 -O3 -o occupancy occupancy.hip && ./occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamp, const f32x4* wts, int reps, float seed) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
  for (int i = threadIdx.x; i < 96 * 256; i += 512) lds[i] = seed + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 0) {
    for (int r = 0; r < reps; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
  } else if (MODE == 1) {
    f32x4* t = reinterpret_cast<f32x4*>(lds);
    f32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int idx = (threadIdx.x + 512 * j) % (96 * 64);
        f32x4 v = t[idx];
        acc += v;
        t[idx] = acc;
      }
    a[0] += acc[0] + acc[1] + acc[2] + acc[3];
  } else if (MODE == 2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n16 = (2400 * 1024) / 16;  // 16-byte elements; wave w reads chunks w, w + 8, ... of 64 elements
    f32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
      for (int c = wave * 64; c + 192 <= n16; c += 8 * 64 * 3) {
        const f32x4 v0 = wts[c + lane], v1 = wts[c + 512 + lane], v2 = wts[c + 1024 + lane];
        acc += v0 + v1 + v2;
      }
    a[0] += acc[0] + acc[1] + acc[2] + acc[3];
  } else if (MODE == 4 || MODE == 5) {
    bf16x8 wa, xb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { wa[i] = (__bf16)(seed + i); xb[i] = (__bf16)(seed - i); }
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f32x4* t = reinterpret_cast<f32x4*>(lds);
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[i], 0, 0, 0);
      if (MODE == 5) {
#pragma unroll
        for (int j = 0; j < 48; ++j) {
          const int idx = (threadIdx.x + 512 * j) % (96 * 64);
          f32x4 v = t[idx];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(v[q & 3]));
          v[0] += a[j & 7];
          t[idx] = v;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i & 7] += acc[i][0] + acc[i][3];
  } else {  // modes 3, 6, 7
    for (int r = 0; r < reps; ++r) {
#pragma unroll
      for (int j = 0; j < 768; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { stamp[2 * blockIdx.x] = t1 - t0; stamp[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int reps, float* out, unsigned long long* stamp, const f32x4* wts) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  printf("%s\n", name);
  const int Gs[] = {8, 64, 96, 128, 160, 200, 256};
  for (int G : Gs) {
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(512), 150 * 1024, 0, out, stamp, wts, reps, 1.0001f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * G);
    hipMemcpy(h.data(), stamp, 2 * G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc, us;
    for (int b = 0; b < G; ++b) { cyc.push_back((double)h[2 * b]); us.push_back(h[2 * b + 1] / 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(us.begin(), us.end());
    printf("  G = %3d (%4.1f per XCD): clocks median %9.0f max %9.0f | us median %7.1f max %7.1f | clocks / us %.0f\n", G, G / 8.0, cyc[G / 2], cyc[G - 1],
           us[G / 2], us[G - 1], cyc[G / 2] / us[G / 2]);
  }
}

void run_cold(float* out, unsigned long long* stamp, const f32x4* wts) {
  hipFuncSetAttribute((const void*)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipFuncSetAttribute((const void*)k<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  printf("mode 6 / 7: ~48 KB of straight-line VALU executed ONCE per launch, two copies of the kernel launched alternately (cold instruction cache)\n");
  const int Gs[] = {8, 64, 96, 128, 160, 200, 256};
  for (int G : Gs) {
    std::vector<double> med;
    for (int it = 0; it < 6; ++it) {
      if (it & 1) hipLaunchKernelGGL(k<7>, dim3(G), dim3(512), 150 * 1024, 0, out, stamp, wts, 1, 1.0001f);
      else hipLaunchKernelGGL(k<6>, dim3(G), dim3(512), 150 * 1024, 0, out, stamp, wts, 1, 1.0001f);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(2 * G);
      hipMemcpy(h.data(), stamp, 2 * G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::vector<double> cyc;
      for (int b = 0; b < G; ++b) cyc.push_back((double)h[2 * b]);
      std::sort(cyc.begin(), cyc.end());
      if (it >= 2) med.push_back(cyc[G / 2]);
    }
    std::sort(med.begin(), med.end());
    printf("  G = %3d (%4.1f per XCD): clocks for the one pass, median workgroup, over 4 launches: min %8.0f max %8.0f   (warm, mode 3: ~29 500)\n", G, G / 8.0,
           med.front(), med.back());
  }
}

int main() {
  float* out; unsigned long long* stamp; f32x4* wts;
  hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&stamp, 2 * 256 * sizeof(unsigned long long)); hipMalloc(&wts, 2400 * 1024);
  hipMemset(wts, 0, 2400 * 1024);
  run<0>("mode 0: VALU only (v_fma_f32 chains in a loop)", 4096, out, stamp, wts);
  run<1>("mode 1: LDS only (ds_read_b128 + ds_write_b128 rounds)", 1024, out, stamp, wts);
  run<2>("mode 2: L2 stream (every workgroup reads the same 2.4 MB, 16 B per lane, 3 loads in flight per wave)", 24, out, stamp, wts);
  run<3>("mode 3: straight-line VALU (~48 KB of unrolled v_fma_f32)", 24, out, stamp, wts);
  run_cold(out, stamp, wts);
  run<4>("mode 4: MFMA only (12 accumulators per wave, 2 waves per SIMD)", 2048, out, stamp, wts);
  run<5>("mode 5: chain-like mix (96 MFMAs, then 1 536 VALU ops + 48 LDS round trips, per repetition)", 256, out, stamp, wts);
  return 0;
}
