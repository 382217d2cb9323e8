// VALU issue rates on gfx950 at the occupancy of the fused chains (one 512-thread workgroup per CU = 2 waves per SIMD):
// cycles per wave64 instruction per SIMD for the instruction classes an epilogue is made of.  Each test is a loop of 8 independent
// chains x 16 unrolled ops (no memory), timed with s_memtime by every wave; the slowest wave of workgroup 0 is reported.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 256
template <int OP>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, float seed) {
  float a[8]; unsigned u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 2654435761u + i; }
  const unsigned seedu = __float_as_uint(seed);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
        if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        if (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 3) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[i]));
        if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "+v"(u[i]) : "v"(a[i]));
        if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        if (OP == 10) asm volatile("v_cmp_le_u32 vcc, %0, %1" :: "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc");
        if (OP == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 12) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 13) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 14) asm volatile("v_cmp_le_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]), "v"(seedu) : "vcc");
        if (OP == 15) asm volatile("v_sub_u32 %0, %1, %0\n\tv_ashrrev_i32 %0, 31, %0\n\tv_and_b32 %0, %0, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(seedu));
        if (OP == 16) asm volatile("v_bfe_u32 %0, %0, 3, 16" : "+v"(u[i]));
        if (OP == 17) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == 18) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(u[i]));
        if (OP == 19) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(seedu));
        if (OP == 20) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(seedu));
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; unsigned x = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { s += a[i]; x ^= u[i]; }
  out[blockIdx.x * 512 + threadIdx.x] = s + (float)x;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
// packed: 2 floats per op
__global__ __launch_bounds__(512) void kpk(float* out, unsigned long long* cyc, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f2{seed + threadIdx.x * 0.001f + i, seed + i};
  const f2 c = {seed, seed * 0.5f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const char* names[] = {"v_fma_f32", "v_mul_f32", "v_and_b32", "v_lshrrev_b32", "v_mad_u32_u24", "v_exp_f32", "v_rcp_f32", "v_cvt_pk_bf16_f32",
                         "v_cndmask_b32 (dst also a source, other source another chain)", "v_max_f32", "v_cmp_le_u32", "v_mul_lo_u32", "v_xor_b32",
                         "v_min_u32", "v_cmp + v_cndmask (2 instr)", "v_sub + v_ashr + v_and (3 instr)", "v_bfe_u32", "v_sub_u32",
                         "v_ashrrev_i32", "v_and_or_b32", "v_cndmask_b32 (uniform other source)", "v_pk_fma_f32"};
  unsigned long long h[8];
  for (int op = 0; op <= 21; ++op) {
    for (int it = 0; it < 2; ++it) {
      switch (op) {
#define C(N) case N: hipLaunchKernelGGL(k<N>, dim3(256), dim3(512), 0, 0, out, cyc, 1.0001f); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20)
        default: hipLaunchKernelGGL(kpk, dim3(256), dim3(512), 0, 0, out, cyc, 1.0001f);
      }
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < 8; ++w) if (h[w] > mx) mx = h[w];
    // each wave issued REP * 16 instructions; two waves share a SIMD: 2 * REP * 16 instructions per SIMD in mx memtime ticks
    printf("%-20s %8llu shader cycles (s_memtime) = %.2f cycles per wave-instruction per SIMD (2 waves/SIMD)\n", names[op], mx, (double)mx / (2.0 * REP * 16));
  }
  return 0;
}
