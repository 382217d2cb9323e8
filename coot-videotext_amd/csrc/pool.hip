// GenPool softmax-over-sequence pooling, avg_special pooling, clip packing (gfx950).
// All HBM-bound and tiny next to the GEMMs: one workgroup per sequence, a thread owns two
// adjacent channels (4-byte bf16x2 loads, fully coalesced rows), the L loop runs in registers.
#include "pool.h"

namespace coot {

// GenPool.forward tail (nntrainer/models/poolers.py:188-206): padded rows -> -INF, softmax over
// the sequence axis per channel, pooled = sum_l z * w.
__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolArgs p) {
  const int n = blockIdx.x;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 2;
  if (c >= p.D) return;
  const int len = (int)p.lens[n];
  const long r0 = (long)n * p.L;
  float m0 = -INFINITY, m1 = -INFINITY;
  // rows >= len hold -INF(=-32752) after masked_fill; they only matter if len == 0
  for (int l = 0; l < len; ++l) {
    unsigned u = *reinterpret_cast<const unsigned*>(p.s + (r0 + l) * p.lds + c);
    m0 = fmaxf(m0, bflo(u)); m1 = fmaxf(m1, bfhi(u));
  }
  if (len < p.L) { m0 = fmaxf(m0, kMaskFill); m1 = fmaxf(m1, kMaskFill); }
  float z0 = 0.f, z1 = 0.f, a0 = 0.f, a1 = 0.f;
  for (int l = 0; l < len; ++l) {
    unsigned u = *reinterpret_cast<const unsigned*>(p.s + (r0 + l) * p.lds + c);
    unsigned f = *reinterpret_cast<const unsigned*>(p.z + (r0 + l) * p.ldz + c);
    float e0 = __expf(bflo(u) - m0), e1 = __expf(bfhi(u) - m1);
    z0 += e0; z1 += e1;
    if (p.drop_w.thr) {
      unsigned long long idx = (unsigned long long)(r0 + l) * p.D + c;
      float sc[2];
      drop_scales<2>(eff_seed(p.drop_w.seed, p.drop_w.seed_ptr), p.drop_w.site, idx, p.drop_w.thr, p.drop_w.inv_keep, sc);
      e0 *= sc[0]; e1 *= sc[1];
    }
    a0 += e0 * bflo(f); a1 += e1 * bfhi(f);
  }
  // masked rows contribute exp(-32752 - m) == 0 in fp32 unless every row is masked
  if (len < p.L) { float e0 = __expf(kMaskFill - m0), e1 = __expf(kMaskFill - m1); z0 += (p.L - len) * e0; z1 += (p.L - len) * e1; }
  p.pooled[(long)n * p.ldp + c] = a0 / z0;
  p.pooled[(long)n * p.ldp + c + 1] = a1 / z1;
  if (p.pooled_copy) { p.pooled_copy[(long)n * p.D + c] = a0 / z0; p.pooled_copy[(long)n * p.D + c + 1] = a1 / z1; }
  if (p.smax) {
    p.smax[(long)n * p.D + c] = m0; p.smax[(long)n * p.D + c + 1] = m1;
    p.ssum[(long)n * p.D + c] = z0; p.ssum[(long)n * p.D + c + 1] = z1;
  }
}

// dw = dpooled*z; ds = w*(dw - sum_l w*dw) = w*(dw*drop3 - dpooled*pooled); dz = dpooled*w*drop3
__global__ __launch_bounds__(256) void pool_bwd_kernel(PoolArgs p) {
  const int n = blockIdx.x;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 2;
  if (c >= p.D) return;
  const int len = (int)p.lens[n];
  const long r0 = (long)n * p.L;
  const float m0 = p.smax[(long)n * p.D + c], m1 = p.smax[(long)n * p.D + c + 1];
  const float iz0 = 1.f / p.ssum[(long)n * p.D + c], iz1 = 1.f / p.ssum[(long)n * p.D + c + 1];
  const float g0 = p.dpooled[(long)n * p.lddp + c], g1 = p.dpooled[(long)n * p.lddp + c + 1];
  const float gp0 = g0 * p.pooled[(long)n * p.ldp + c], gp1 = g1 * p.pooled[(long)n * p.ldp + c + 1];
  float cs0 = 0.f, cs1 = 0.f;
  for (int l = 0; l < p.L; ++l) {
    float ds0 = 0.f, ds1 = 0.f, dz0 = 0.f, dz1 = 0.f;
    if (l < len) {
      unsigned u = *reinterpret_cast<const unsigned*>(p.s + (r0 + l) * p.lds + c);
      unsigned f = *reinterpret_cast<const unsigned*>(p.z + (r0 + l) * p.ldz + c);
      float w0 = __expf(bflo(u) - m0) * iz0, w1 = __expf(bfhi(u) - m1) * iz1;
      float d30 = 1.f, d31 = 1.f;
      if (p.drop_w.thr) {
        unsigned long long idx = (unsigned long long)(r0 + l) * p.D + c;
        float sc[2];
        drop_scales<2>(eff_seed(p.drop_w.seed, p.drop_w.seed_ptr), p.drop_w.site, idx, p.drop_w.thr, p.drop_w.inv_keep, sc);
        d30 = sc[0]; d31 = sc[1];
      }
      ds0 = w0 * (g0 * bflo(f) * d30 - gp0);
      ds1 = w1 * (g1 * bfhi(f) * d31 - gp1);
      dz0 = g0 * w0 * d30; dz1 = g1 * w1 * d31;
      if (p.drop_s.thr) {
        unsigned long long idx = (unsigned long long)(p.drop_s_row0 + r0 + l) * p.drop_s_ld + c;
        float sc[2];
        drop_scales<2>(eff_seed(p.drop_s.seed, p.drop_s.seed_ptr), p.drop_s.site, idx, p.drop_s.thr, p.drop_s.inv_keep, sc);
        ds0 *= sc[0]; ds1 *= sc[1];
      }
      cs0 += ds0; cs1 += ds1;
    }
    *reinterpret_cast<unsigned*>(p.ds + (r0 + l) * p.ldds + c) = pack2bf(ds0, ds1);
    *reinterpret_cast<unsigned*>(p.dz + (r0 + l) * p.lddz + c) = pack2bf(dz0, dz1);
  }
  if (p.part_ws) { p.part_ws[(long)n * p.D + c] = cs0; p.part_ws[(long)n * p.D + c + 1] = cs1; }
  else if (p.ds_colsum) { atomicAdd(p.ds_colsum + c, cs0); atomicAdd(p.ds_colsum + c + 1, cs1); }
}

static int pool_check(const PoolArgs& p) {
  COOT_REQUIRE(p.s && p.z && p.lens && p.pooled, "pool: null pointer");
  COOT_REQUIRE(p.D % 2 == 0 && p.lds % 2 == 0 && p.ldz % 2 == 0, "pool: D must be even");
  return 0;
}
int launch_pool_fwd(const PoolArgs& p, hipStream_t st) {
  if (int rc = pool_check(p)) return rc;
  if (p.N <= 0) return 0;
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(p.N, (p.D / 2 + 255) / 256), dim3(256), 0, st, p);
  COOT_CHECK_LAUNCH("pool_fwd");
  return 0;
}
int launch_pool_bwd(const PoolArgs& p_in, hipStream_t st) {
  PoolArgs p = p_in;
  if (int rc = pool_check(p)) return rc;
  COOT_REQUIRE(p.dpooled && p.ds && p.dz && p.smax && p.ssum, "pool bwd: null pointer");
  if (p.N <= 0) return 0;
  p.part_ws = p.ds_colsum ? partials_workspace((size_t)p.N * p.D) : nullptr;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(p.N, (p.D / 2 + 255) / 256), dim3(256), 0, st, p);
  COOT_CHECK_LAUNCH("pool_bwd");
  if (p.part_ws) return launch_reduce_partials(p.part_ws, p.N, p.D, p.D, p.ds_colsum, st);
  return 0;
}

// ---- avg_special ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16_t* z, long ldz, const long long* lens, int L, int D, float* out, long ldo) {
  const int n = blockIdx.x, c = (blockIdx.y * 256 + threadIdx.x) * 2;
  if (c >= D) return;
  float a0 = 0.f, a1 = 0.f;
  for (int l = 0; l < L; ++l) {
    unsigned f = *reinterpret_cast<const unsigned*>(z + ((long)n * L + l) * ldz + c);
    a0 += bflo(f); a1 += bfhi(f);
  }
  const float inv = 1.0f / (float)lens[n];
  out[(long)n * ldo + c] = a0 * inv; out[(long)n * ldo + c + 1] = a1 * inv;
}
int launch_avgpool_fwd(const bf16_t* z, long ldz, const long long* lens, int N, int L, int D, float* out, long ldo, hipStream_t st) {
  COOT_REQUIRE(z && lens && out && D % 2 == 0, "avgpool: bad args");
  if (N <= 0) return 0;
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(N, (D / 2 + 255) / 256), dim3(256), 0, st, z, ldz, lens, L, D, out, ldo);
  COOT_CHECK_LAUNCH("avgpool_fwd");
  return 0;
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* dp, long lddp, const long long* lens, int L, int D, bf16_t* dz, long lddz) {
  const int n = blockIdx.x, c = (blockIdx.y * 256 + threadIdx.x) * 2;
  if (c >= D) return;
  const float inv = 1.0f / (float)lens[n];
  const unsigned v = pack2bf(dp[(long)n * lddp + c] * inv, dp[(long)n * lddp + c + 1] * inv);
  for (int l = 0; l < L; ++l) *reinterpret_cast<unsigned*>(dz + ((long)n * L + l) * lddz + c) = v;
}
int launch_avgpool_bwd(const float* dp, long lddp, const long long* lens, int N, int L, int D, bf16_t* dz, long lddz, hipStream_t st) {
  COOT_REQUIRE(dp && lens && dz && D % 2 == 0, "avgpool bwd: bad args");
  if (N <= 0) return 0;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(N, (D / 2 + 255) / 256), dim3(256), 0, st, dp, lddp, lens, L, D, dz, lddz);
  COOT_CHECK_LAUNCH("avgpool_bwd");
  return 0;
}

// ---- pack by count --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_fwd_kernel(const float* emb, const long long* counts, int B, int Cmax, int D, float* out,
                                                       unsigned char* mask, long long* lens) {
  const int b = blockIdx.x;
  long ptr = 0;
  for (int i = 0; i < b; ++i) ptr += counts[i];
  const int cnt = (int)counts[b];
  for (int i = threadIdx.x; i < Cmax * D; i += 256) {
    const int c = i / D, d = i % D;
    out[((long)b * Cmax + c) * D + d] = c < cnt ? emb[(ptr + c) * D + d] : 0.f;
  }
  if (mask) for (int c = threadIdx.x; c < Cmax; c += 256) mask[(long)b * Cmax + c] = c < cnt ? 0 : 1;
  if (lens && threadIdx.x == 0) lens[b] = cnt;
}
int launch_pack_fwd(const float* emb, const long long* counts, int B, int Cmax, int D, float* out, unsigned char* mask,
                    long long* lens, hipStream_t st) {
  COOT_REQUIRE(emb && counts && out, "pack: null pointer");
  if (B <= 0) return 0;
  hipLaunchKernelGGL(pack_fwd_kernel, dim3(B), dim3(256), 0, st, emb, counts, B, Cmax, D, out, mask, lens);
  COOT_CHECK_LAUNCH("pack_fwd");
  return 0;
}
__global__ __launch_bounds__(256) void pack_bwd_kernel(const float* dout, const long long* counts, int B, int Cmax, int D, float* demb) {
  const int b = blockIdx.x;
  long ptr = 0;
  for (int i = 0; i < b; ++i) ptr += counts[i];
  const int cnt = (int)counts[b];
  for (int i = threadIdx.x; i < cnt * D; i += 256) {
    const int c = i / D, d = i % D;
    demb[(ptr + c) * D + d] += dout[((long)b * Cmax + c) * D + d];
  }
}
int launch_pack_bwd(const float* dout, const long long* counts, int B, int Cmax, int D, float* demb, hipStream_t st) {
  COOT_REQUIRE(dout && counts && demb, "pack bwd: null pointer");
  if (B <= 0) return 0;
  hipLaunchKernelGGL(pack_bwd_kernel, dim3(B), dim3(256), 0, st, dout, counts, B, Cmax, D, demb);
  COOT_CHECK_LAUNCH("pack_bwd");
  return 0;
}

// ---- casts ----------------------------------------------------------------------------------
__global__ void cast_bf16_f32_kernel(const bf16_t* src, long lds, int R, int C, float* dst, long ldd) {
  const long total = (long)R * (C / 2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (C / 2); const int c = (int)(i % (C / 2)) * 2;
    unsigned u = *reinterpret_cast<const unsigned*>(src + r * lds + c);
    dst[r * ldd + c] = bflo(u); dst[r * ldd + c + 1] = bfhi(u);
  }
}
int launch_cast_bf16_f32(const bf16_t* src, long lds, int R, int C, float* dst, long ldd, hipStream_t st) {
  COOT_REQUIRE(src && dst && C % 2 == 0, "cast: bad args");
  if (R <= 0) return 0;
  long total = (long)R * (C / 2);
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(blocks), dim3(256), 0, st, src, lds, R, C, dst, ldd);
  COOT_CHECK_LAUNCH("cast_bf16_f32");
  return 0;
}
__global__ void cast_f32_bf16_kernel(const float* src, long lds, int R, int C, bf16_t* dst, long ldd) {
  const long total = (long)R * (C / 2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (C / 2); const int c = (int)(i % (C / 2)) * 2;
    *reinterpret_cast<unsigned*>(dst + r * ldd + c) = pack2bf(src[r * lds + c], src[r * lds + c + 1]);
  }
}
int launch_cast_f32_bf16(const float* src, long lds, int R, int C, bf16_t* dst, long ldd, hipStream_t st) {
  COOT_REQUIRE(src && dst && C % 2 == 0, "cast: bad args");
  if (R <= 0) return 0;
  long total = (long)R * (C / 2);
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, st, src, lds, R, C, dst, ldd);
  COOT_CHECK_LAUNCH("cast_f32_bf16");
  return 0;
}
__global__ void add_bf16_to_f32_kernel(const bf16_t* a, long lda, const bf16_t* b, long ldb, int R, int C, float* dst, long ldd, int acc) {
  const long total = (long)R * (C / 2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / (C / 2); const int c = (int)(i % (C / 2)) * 2;
    unsigned u = *reinterpret_cast<const unsigned*>(a + r * lda + c);
    float v0 = bflo(u), v1 = bfhi(u);
    if (b) { unsigned w = *reinterpret_cast<const unsigned*>(b + r * ldb + c); v0 += bflo(w); v1 += bfhi(w); }
    if (acc) { dst[r * ldd + c] += v0; dst[r * ldd + c + 1] += v1; }
    else { dst[r * ldd + c] = v0; dst[r * ldd + c + 1] = v1; }
  }
}
int launch_add_bf16_to_f32(const bf16_t* a, long lda, const bf16_t* b, long ldb, int R, int C, float* dst, long ldd, int accumulate, hipStream_t st) {
  COOT_REQUIRE(a && dst && C % 2 == 0, "add: bad args");
  if (R <= 0) return 0;
  long total = (long)R * (C / 2);
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_bf16_to_f32_kernel, dim3(blocks), dim3(256), 0, st, a, lda, b, ldb, R, C, dst, ldd, accumulate);
  COOT_CHECK_LAUNCH("add_bf16_to_f32");
  return 0;
}

}  // namespace coot
