// GenPool softmax-over-sequence pooling, avg_special pooling, clip packing (gfx950).
// All HBM-bound and tiny next to the GEMMs: 16-byte loads, fully coalesced rows, the L loop runs in registers.
#include "pool.h"
#include "det.h"

namespace coot {

// GenPool.forward tail (nntrainer/models/poolers.py:188-206): padded rows -> -INF, softmax over the sequence axis per
// channel, pooled = sum_l z * w.
//
// The channels are independent, so a workgroup takes one sequence x 128 channels (grid = sequences x D / 128; the whole
// row if D is not a multiple of 128): thread t -> 8-channel chunk t % CPB (16-byte loads), row group t / CPB (RG = 256 /
// CPB = 16 row groups).  A thread walks its rows four at a time — the eight loads of a batch are in flight together —
// with an online softmax; the RG partial states are merged through LDS.  History: two passes of dependent 4-byte loads,
// one workgroup per sequence: 48 us per launch (0.8 TB/s); one pass of 16-byte loads, 4 row groups: 17 us (one 4-wave
// workgroup per CU walking 20 rows one load pair at a time); this version fills the chip.
constexpr int POOL_MAXCH = 64;  // D <= 512
constexpr int POOL_RB = 4;      // rows per load batch

__device__ __forceinline__ void pool_unpack(u32x4_t u, float* v) {
  v[0] = bflo(u[0]); v[1] = bfhi(u[0]); v[2] = bflo(u[1]); v[3] = bfhi(u[1]);
  v[4] = bflo(u[2]); v[5] = bfhi(u[2]); v[6] = bflo(u[3]); v[7] = bfhi(u[3]);
}
// chunks per workgroup / row groups for D channels
__host__ __device__ inline int pool_cpb(int D) { return (D % 128 == 0) ? 16 : D / 8; }

struct PoolArgs2 { PoolArgs seg[2]; int n0; };  // sequences [0, n0) = segment 0, the rest segment 1

// (video b, position c, count of b) of item n: prefix search over counts[0 .. B) by the whole 256-thread workgroup, result through LDS
// (sh: 8 ints).  Its one global load is issued before the kernel's main loop, so the search costs a scan, not a memory round trip.
__device__ __forceinline__ void pool_find_video(const long long* counts, int B, int n, int* sh, int& b, int& c, int& cnt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long carry = 0;
  if (tid == 0) sh[4] = -1;
  for (int base = 0; base < B; base += 256) {
    const int v = (base + tid < B) ? (int)counts[base + tid] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += sh[w];
    const int tot = sh[0] + sh[1] + sh[2] + sh[3];
    const long excl = carry + woff + inc - v;
    if (v > 0 && n >= excl && n < excl + v) { sh[4] = base + tid; sh[5] = (int)(n - excl); sh[6] = v; }
    __syncthreads();
    if (sh[4] >= 0) break;
    carry += tot;
  }
  b = sh[4]; c = sh[5]; cnt = sh[6];
}
// forward hand-over, rows of video vb the items do not cover (cntb .. Cmax - 1) of this workgroup's channels -> 0; channel block 0 also
// writes the video's mask row and length
__device__ __forceinline__ void pool_pack_rest(const PoolArgs& p, int vb, int cntb, int cpb) {
  const int Cm = p.pk_Cmax, nch = cpb * 8;
  for (int i = threadIdx.x; i < (Cm - cntb) * nch; i += 256) {
    const int r = cntb + i / nch, ch = blockIdx.y * nch + i % nch;
    p.pk_out[((long)vb * Cm + r) * p.D + ch] = 0.f;
  }
  if (blockIdx.y == 0) {
    if (p.pk_mask) for (int r = threadIdx.x; r < Cm; r += 256) p.pk_mask[(long)vb * Cm + r] = r < cntb ? 0 : 1;
    if (p.pk_lens && threadIdx.x == 0) p.pk_lens[vb] = cntb;
  }
}

__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolArgs2 pp) {
  __shared__ float red[3][256 * 8 + 8];
  const bool second = (int)blockIdx.x >= pp.n0;
  const PoolArgs& p = pp.seg[second ? 1 : 0];
  const unsigned kw = p.drop_w.thr ? drop_site_key(p.drop_w.seed, p.drop_w.seed_ptr, p.drop_w.site) : 0u;
  const int n = (int)blockIdx.x - (second ? pp.n0 : 0), cpb = pool_cpb(p.D), rgs = 256 / cpb;
  const int cl = threadIdx.x % cpb, rg = threadIdx.x / cpb, c = (blockIdx.y * cpb + cl) * 8;
  const int len = (int)p.lens[n];
  const long r0 = p.cu ? (long)p.cu[n] : (long)n * p.L;
  const int Lp = p.cu ? len : p.L;  // rows the sequence occupies (padded length, or its own when packed)
  __shared__ int pk_sh[8];
  int vb = -1, vc = 0, vcnt = 0;
  if (p.pk_out) pool_find_video(p.pk_counts, p.pk_B, n, pk_sh, vb, vc, vcnt);
  float m[8], z[8], a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; z[j] = 0.f; a[j] = 0.f; }
  if (rg < rgs) {
    for (int l0 = rg; l0 < len; l0 += rgs * POOL_RB) {
      u32x4_t su[POOL_RB], fu[POOL_RB];
#pragma unroll
      for (int b = 0; b < POOL_RB; ++b) {
        const int l = l0 + b * rgs;
        su[b] = u32x4_t{0u, 0u, 0u, 0u}; fu[b] = su[b];
        if (l < len) {
          su[b] = *reinterpret_cast<const u32x4_t*>(p.s + (r0 + l) * p.lds + c);
          fu[b] = *reinterpret_cast<const u32x4_t*>(p.z + (r0 + l) * p.ldz + c);
        }
      }
#pragma unroll
      for (int b = 0; b < POOL_RB; ++b) {
        const int l = l0 + b * rgs;
        if (l < len) {
          float s[8], f[8], sc[8];
          pool_unpack(su[b], s);
          pool_unpack(fu[b], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) sc[j] = 1.f;
          if (p.drop_w.thr) drop_scales_key<8>(kw, (unsigned long long)(r0 + l) * p.D + c, p.drop_w.thr, p.drop_w.inv_keep, sc);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float mn = fmaxf(m[j], s[j]);
            const float r = __expf(m[j] - mn), e = __expf(s[j] - mn);  // first row: exp(-inf) = 0
            z[j] = z[j] * r + e;
            a[j] = a[j] * r + e * sc[j] * f[j];
            m[j] = mn;
          }
        }
      }
    }
    float* w = &red[0][(rg * cpb + cl) * 8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w[j] = m[j]; w[(256 * 8 + 8) + j] = z[j]; w[2 * (256 * 8 + 8) + j] = a[j]; }
  }
  __syncthreads();
  for (int cc = threadIdx.x; cc < cpb * 8; cc += 256) {  // one thread per channel of the block merges the row groups
    float mm = -INFINITY;
    for (int g = 0; g < rgs; ++g) mm = fmaxf(mm, red[0][g * cpb * 8 + cc]);
    // rows >= len hold -INF (= -32752) after masked_fill (poolers.py:190)
    if (len < Lp) mm = fmaxf(mm, kMaskFill);
    float zz = 0.f, aa = 0.f;
    for (int g = 0; g < rgs; ++g) {
      const float mg = red[0][g * cpb * 8 + cc];
      const float r = mg == -INFINITY ? 0.f : __expf(mg - mm);
      zz += red[1][g * cpb * 8 + cc] * r; aa += red[2][g * cpb * 8 + cc] * r;
    }
    if (len < Lp) zz += (float)(Lp - len) * __expf(kMaskFill - mm);  // 0 in fp32 unless every row is masked
    const float pooled = aa / zz;
    const int ch = blockIdx.y * cpb * 8 + cc;
    p.pooled[(long)n * p.ldp + ch] = pooled;
    if (p.pooled_copy) p.pooled_copy[(long)n * p.D + ch] = pooled;
    if (p.smax) { p.smax[(long)n * p.D + ch] = mm; p.ssum[(long)n * p.D + ch] = zz; }
    if (vb >= 0) p.pk_out[((long)vb * p.pk_Cmax + vc) * p.D + ch] = pooled;
  }
  if (vb >= 0) {
    // the rest of the padded layout: a video's padding rows, mask and length by its LAST item; videos without items by the first item
    // of the next video that has some (those in front of it) and by the very last item (those behind it)
    if (vc == vcnt - 1) pool_pack_rest(p, vb, vcnt, cpb);
    if (vc == 0) for (int e = vb - 1; e >= 0 && p.pk_counts[e] == 0; --e) pool_pack_rest(p, e, 0, cpb);
    if (n == p.N - 1) for (int e = vb + 1; e < p.pk_B; ++e) pool_pack_rest(p, e, 0, cpb);
  }
}

// dw = dpooled*z; ds = w*(dw - sum_l w*dw) = w*(dw*drop3 - dpooled*pooled); dz = dpooled*w*drop3.
// Purely elementwise given the saved softmax statistics, plus the column sum of ds (bias gradient of the second pooling
// FC): same thread mapping as the forward, 16-byte loads and stores, the RG partial column sums merged through LDS.
__global__ __launch_bounds__(256) void pool_bwd_kernel(PoolArgs2 pp) {
  __shared__ float red[256 * 8 + 8];
  const bool second = (int)blockIdx.x >= pp.n0;
  const PoolArgs& p = pp.seg[second ? 1 : 0];
  const unsigned kw = p.drop_w.thr ? drop_site_key(p.drop_w.seed, p.drop_w.seed_ptr, p.drop_w.site) : 0u;
  const unsigned ks = p.drop_s.thr ? drop_site_key(p.drop_s.seed, p.drop_s.seed_ptr, p.drop_s.site) : 0u;
  const int n = (int)blockIdx.x - (second ? pp.n0 : 0), cpb = pool_cpb(p.D), rgs = 256 / cpb;
  const int cl = threadIdx.x % cpb, rg = threadIdx.x / cpb, c = (blockIdx.y * cpb + cl) * 8;
  const int len = (int)p.lens[n];
  const long r0 = p.cu ? (long)p.cu[n] : (long)n * p.L;
  const int Lp = p.cu ? len : p.L;
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  if (rg < rgs) {
    float m[8], iz[8], g[8], gp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      m[j] = p.smax[(long)n * p.D + c + j];
      iz[j] = 1.f / p.ssum[(long)n * p.D + c + j];
      g[j] = p.dpooled[(long)n * p.lddp + c + j];
      gp[j] = g[j] * p.pooled[(long)n * p.ldp + c + j];
    }
    for (int l0 = rg; l0 < Lp; l0 += rgs * POOL_RB) {
      u32x4_t su[POOL_RB], fu[POOL_RB];
#pragma unroll
      for (int b = 0; b < POOL_RB; ++b) {
        const int l = l0 + b * rgs;
        su[b] = u32x4_t{0u, 0u, 0u, 0u}; fu[b] = su[b];
        if (l < len) {
          su[b] = *reinterpret_cast<const u32x4_t*>(p.s + (r0 + l) * p.lds + c);
          fu[b] = *reinterpret_cast<const u32x4_t*>(p.z + (r0 + l) * p.ldz + c);
        }
      }
#pragma unroll
      for (int b = 0; b < POOL_RB; ++b) {
        const int l = l0 + b * rgs;
        if (l < Lp) {
          float ds[8], dz[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { ds[j] = 0.f; dz[j] = 0.f; }
          if (l < len) {
            float s[8], f[8], d3[8], d2[8];
            pool_unpack(su[b], s);
            pool_unpack(fu[b], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { d3[j] = 1.f; d2[j] = 1.f; }
            if (p.drop_w.thr) drop_scales_key<8>(kw, (unsigned long long)(r0 + l) * p.D + c, p.drop_w.thr, p.drop_w.inv_keep, d3);
            if (p.drop_s.thr) drop_scales_key<8>(ks, (unsigned long long)(p.drop_s_row0 + r0 + l) * p.drop_s_ld + c, p.drop_s.thr, p.drop_s.inv_keep, d2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float w = __expf(s[j] - m[j]) * iz[j];
              ds[j] = w * (g[j] * f[j] * d3[j] - gp[j]) * d2[j];
              dz[j] = g[j] * w * d3[j];
              cs[j] += ds[j];
            }
          }
          *reinterpret_cast<u32x4_t*>(p.ds + (r0 + l) * p.ldds + c) = u32x4_t{pack2bf(ds[0], ds[1]), pack2bf(ds[2], ds[3]), pack2bf(ds[4], ds[5]), pack2bf(ds[6], ds[7])};
          *reinterpret_cast<u32x4_t*>(p.dz + (r0 + l) * p.lddz + c) = u32x4_t{pack2bf(dz[0], dz[1]), pack2bf(dz[2], dz[3]), pack2bf(dz[4], dz[5]), pack2bf(dz[6], dz[7])};
        }
      }
    }
    float* w = &red[(rg * cpb + cl) * 8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = cs[j];
  }
  __syncthreads();
  for (int cc = threadIdx.x; cc < cpb * 8 && (p.part_ws || p.ds_colsum); cc += 256) {
    const int ch = blockIdx.y * cpb * 8 + cc;
    float v = 0.f;
    for (int g2 = 0; g2 < rgs; ++g2) v += red[g2 * cpb * 8 + cc];
    if (p.part_ws) p.part_ws[(long)n * p.D + ch] = v;
    else acc_add(p.ds_colsum + ch, v);
  }
}

static int pool_check(const PoolArgs& p) {
  COOT_REQUIRE(p.s && p.z && p.lens && p.pooled, "pool: null pointer");
  COOT_REQUIRE(p.D % 8 == 0 && p.D >= 64 && p.D <= 8 * POOL_MAXCH && p.lds % 8 == 0 && p.ldz % 8 == 0 && p.ldds % 8 == 0 && p.lddz % 8 == 0,
               "pool: D = %d unsupported (multiple of 8, 64..512)", p.D);
  return 0;
}
int launch_pool_fwd2(const PoolArgs* segs, int nseg, hipStream_t st) {
  COOT_REQUIRE(nseg >= 1 && nseg <= 2, "pool: one or two segments");
  PoolArgs2 pp; int total = 0;
  for (int i = 0; i < nseg; ++i) {
    if (int rc = pool_check(segs[i])) return rc;
    COOT_REQUIRE(segs[i].D == segs[0].D, "pool: the segments share the channel count");
    pp.seg[i] = segs[i]; total += segs[i].N > 0 ? segs[i].N : 0;
  }
  pp.n0 = segs[0].N > 0 ? segs[0].N : 0;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(total, segs[0].D / 8 / pool_cpb(segs[0].D)), dim3(256), 0, st, pp);
  COOT_CHECK_LAUNCH("pool_fwd");
  return 0;
}
int launch_pool_fwd(const PoolArgs& p, hipStream_t st) { return launch_pool_fwd2(&p, 1, st); }

int launch_pool_bwd2(const PoolArgs* segs, int nseg, hipStream_t st) {
  COOT_REQUIRE(nseg >= 1 && nseg <= 2, "pool: one or two segments");
  PoolArgs2 pp; int total = 0;
  for (int i = 0; i < nseg; ++i) {
    if (int rc = pool_check(segs[i])) return rc;
    COOT_REQUIRE(segs[i].dpooled && segs[i].ds && segs[i].dz && segs[i].smax && segs[i].ssum, "pool bwd: null pointer");
    COOT_REQUIRE(segs[i].D == segs[0].D && segs[i].ds_colsum == segs[0].ds_colsum, "pool bwd: the segments share D and the bias gradient");
    pp.seg[i] = segs[i]; total += segs[i].N > 0 ? segs[i].N : 0;
  }
  pp.n0 = segs[0].N > 0 ? segs[0].N : 0;
  if (total <= 0) return 0;
  const int D = segs[0].D;
  // bias-gradient partials: deferred to the pass's single reduction launch if the caller opened a scope (rowops.h)
  float* part = nullptr; bool deferred = false;
  if (segs[0].ds_colsum) {
    part = partials_workspace_top((size_t)total * D);
    deferred = part && colsum_defer_add(segs[0].ds_colsum, part, D, total, D, 0);
    if (!deferred) part = partials_workspace((size_t)total * D);
  }
  for (int i = 0; i < nseg; ++i) pp.seg[i].part_ws = part ? part + (i == 1 ? (size_t)pp.n0 * D : 0) : nullptr;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(total, D / 8 / pool_cpb(D)), dim3(256), 0, st, pp);
  COOT_CHECK_LAUNCH("pool_bwd");
  if (part && !deferred) return launch_reduce_partials(part, total, D, D, segs[0].ds_colsum, st);
  return 0;
}
int launch_pool_bwd(const PoolArgs& p, hipStream_t st) { return launch_pool_bwd2(&p, 1, st); }

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
// exclusive prefix sum_{i < b} counts[i] by the whole workgroup (a serial loop of dependent loads was ~5 us for block 63)
__device__ __forceinline__ long pack_prefix(const long long* counts, int b) {
  __shared__ long long part[4];
  long long v = 0;
  for (int i = threadIdx.x; i < b; i += 256) v += counts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  return (long)(part[0] + part[1] + part[2] + part[3]);
}
__global__ __launch_bounds__(256) void pack_fwd_kernel(const float* emb, const long long* counts, int B, int Cmax, int D, float* out,
                                                       unsigned char* mask, long long* lens) {
  const int b = blockIdx.x;
  const long ptr = pack_prefix(counts, b);
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
  const long ptr = pack_prefix(counts, b);
  const int cnt = (int)counts[b];
  for (int i = threadIdx.x; i < cnt * D; i += 256) {
    const int c = i / D, d = i % D;
    demb[(ptr + c) * D + d] += dout[((long)b * Cmax + c) * D + d];
  }
}
// the hand-over from a global network's backward to its local network's in ONE launch (was axpy + pack_bwd + pack_bwd, three
// dependent 5 us launches): item rows += unpack(dout1) (+ unpack(dout2)), context rows += dctx
__global__ __launch_bounds__(256) void pack_bwd_join_kernel(const float* dout1, const float* dout2, const float* dctx, const long long* counts,
                                                            int B, int Cmax, int D, float* demb_items, float* demb_ctx) {
  const int b = blockIdx.x;
  const long ptr = pack_prefix(counts, b);
  const int cnt = (int)counts[b];
  for (int i = threadIdx.x; i < cnt * D; i += 256) {
    const int c = i / D, d = i % D;
    const long src = ((long)b * Cmax + c) * D + d;
    float v = dout1[src];
    if (dout2) v += dout2[src];
    demb_items[(ptr + c) * D + d] += v;
  }
  if (dctx) for (int d = threadIdx.x; d < D; d += 256) demb_ctx[(long)b * D + d] += dctx[(long)b * D + d];
}
int launch_pack_bwd_join(const float* dout1, const float* dout2, const float* dctx, const long long* counts, int B, int Cmax, int D,
                         float* demb_items, float* demb_ctx, hipStream_t st) {
  COOT_REQUIRE(dout1 && counts && demb_items && (!dctx || demb_ctx), "pack bwd join: null pointer");
  if (B <= 0) return 0;
  hipLaunchKernelGGL(pack_bwd_join_kernel, dim3(B), dim3(256), 0, st, dout1, dout2, dctx, counts, B, Cmax, D, demb_items, demb_ctx);
  COOT_CHECK_LAUNCH("pack_bwd_join");
  return 0;
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

namespace coot {
COOT_DET_DEFINE_SETTER(pool)
}  // namespace coot
