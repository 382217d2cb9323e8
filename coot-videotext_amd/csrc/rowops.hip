// Row-wise kernels for gfx950: one 64-lane wave per row, the row lives in registers,
// 8/16-byte coalesced loads, wave-shuffle reductions (no LDS, no barriers in the forward).
#include "rowops.h"
#include "det.h"
#include "gemm.h"
#include "fused.h"

namespace coot {

// ---- row load/store helpers: chunk = 4 consecutive elements ---------------------------------
__device__ __forceinline__ f32x4_t load4(const void* base, int is_f32, long off) {
  // fp32 sources are input features, read exactly once per step: streaming loads (no reuse to keep in the caches)
  if (is_f32) return __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(base) + off));
  u32x2_t u = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const bf16_t*>(base) + off);
  return f32x4_t{bflo(u[0]), bfhi(u[0]), bflo(u[1]), bfhi(u[1])};
}
__device__ __forceinline__ void store4_bf(bf16_t* p, f32x4_t v) {
  u32x2_t pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
  *reinterpret_cast<u32x2_t*>(p) = pk;
}

// LayerNormalization 'layernorm_coot' (nntrainer/models/normalizations.py:98-101):
//   y = gain * (x - mean) / (std_unbiased + 1e-6) + bias
// min 8 waves per SIMD: left alone hipcc keeps the 8-chunk row AND all unrolled epilogue temporaries live (239 VGPRs, 2 waves
// per SIMD, 64 KB in flight per CU: the input LayerNorm ran at 2.4 TB/s); 64 VGPRs are enough and fill the CU
// (rows of more than 2 048 elements — NV = 16: 64 row registers per lane — get 128 registers: at 64 the row itself was spilled, 228 B/lane)
template <int NV>
__global__ __launch_bounds__(256, NV <= 8 ? 8 : 4) void ln_fwd_kernel(LnFwd p) {
  const unsigned dkey = p.drop.thr ? drop_site_key(p.drop.seed, p.drop.seed_ptr, p.drop.site) : 0u;
  const int lane = threadIdx.x & 63;
  // (wave-uniform: as a scalar the row's base addresses live in SGPRs — as per-lane 64-bit values one of them was parked in scratch)
  const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (row >= p.R) return;
  const int D = p.D, nch = D >> 2;
  bool second = p.x2 && row >= p.R0;
  long xrow = second ? row - p.R0 : row;
  if (p.cu) {  // packed rows: find the sequence (largest n with cu[n] <= row), gather from the padded source
    int lo = 0, hi = p.nseq;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.cu[mid] <= row) lo = mid; else hi = mid; }
    const int l = row - p.cu[lo];
    second = lo >= p.N0;
    xrow = second ? (long)(lo - p.N0) * p.L1 + l : (long)lo * p.L0 + l;
    if (p.src_packed) { second = false; xrow = row; }  // packed at the source: the row is where it belongs already
    if (p.pos_out && lane == 0) p.pos_out[row] = l;
  }
  const void* xsrc = second ? p.x2 : p.x;
  f32x4_t v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + 64 * i;
    v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (c < nch) v[i] = load4(xsrc, p.x_f32, xrow * p.ldx + c * 4);
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { float d = v[i][j] - mean; v[i][j] = d; q += d * d; }
    }
  }
  const float stdv = sqrtf(wave_sum(q) / (float)(D - 1));
  const float rs = 1.0f / (stdv + kLnEps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + 64 * i;
    if (c >= nch) continue;
    const int col = c * 4;
    f32x4_t y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = v[i][j] * rs;
      if (p.gain) t = t * p.gain[col + j] + p.bias[col + j];
      if (p.pe) t += p.pe[(long)(row % p.pe_L) * D + col + j];
      y[j] = t;
    }
    if (p.drop.thr) {
      float sc[4];
      drop_scales_key<4>(dkey, (unsigned long long)row * D + col, p.drop.thr, p.drop.inv_keep, sc);
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] *= sc[j];
    }
    if (p.y) store4_bf(p.y + (long)row * p.ldy + col, y);
    if (p.y32) *reinterpret_cast<f32x4_t*>(p.y32 + (long)row * p.ldy32 + col) = y;
  }
}

// The prefetched input LayerNorm (p.nt / p.max_wgs: api_step.hip, input stages): the same row code with non-temporal loads of an fp32
// source and stores of the bf16 output, on a capped grid that walks the rows.  A kernel of its own: the plain variant's 64-VGPR budget
// does not survive a row loop around its body (58 VGPRs spilled even for a loop of one trip, 1 800 with the body in a function).
template <int NV>
__global__ __launch_bounds__(256, NV <= 8 ? 4 : 2) void ln_fwd_stream_kernel(LnFwd p) {
  const unsigned dkey = p.drop.thr ? drop_site_key(p.drop.seed, p.drop.seed_ptr, p.drop.site) : 0u;
  const int lane = threadIdx.x & 63;
  for (int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); row < p.R; row += gridDim.x * 4) {
  const int D = p.D, nch = D >> 2;
  bool second = p.x2 && row >= p.R0;
  long xrow = second ? row - p.R0 : row;
  if (p.cu) {  // packed rows: find the sequence (largest n with cu[n] <= row), gather from the padded source
    int lo = 0, hi = p.nseq;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.cu[mid] <= row) lo = mid; else hi = mid; }
    const int l = row - p.cu[lo];
    second = lo >= p.N0;
    xrow = second ? (long)(lo - p.N0) * p.L1 + l : (long)lo * p.L0 + l;
    if (p.src_packed) { second = false; xrow = row; }  // packed at the source: the row is where it belongs already
    if (p.pos_out && lane == 0) p.pos_out[row] = l;
  }
  const void* xsrc = second ? p.x2 : p.x;
  f32x4_t v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + 64 * i;
    v[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (c < nch) {
      if (p.nt && p.x_f32) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>((const float*)xsrc + xrow * p.ldx + c * 4));
      else v[i] = load4(xsrc, p.x_f32, xrow * p.ldx + c * 4);
    }
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { float d = v[i][j] - mean; v[i][j] = d; q += d * d; }
    }
  }
  const float stdv = sqrtf(wave_sum(q) / (float)(D - 1));
  const float rs = 1.0f / (stdv + kLnEps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = lane + 64 * i;
    if (c >= nch) continue;
    const int col = c * 4;
    f32x4_t y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = v[i][j] * rs;
      if (p.gain) t = t * p.gain[col + j] + p.bias[col + j];
      if (p.pe) t += p.pe[(long)(row % p.pe_L) * D + col + j];
      y[j] = t;
    }
    if (p.drop.thr) {
      float sc[4];
      drop_scales_key<4>(dkey, (unsigned long long)row * D + col, p.drop.thr, p.drop.inv_keep, sc);
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] *= sc[j];
    }
    if (p.y && p.nt) {
      const u32x2_t pk = {(unsigned)f2bf(y[0]) | ((unsigned)f2bf(y[1]) << 16), (unsigned)f2bf(y[2]) | ((unsigned)f2bf(y[3]) << 16)};
      __builtin_nontemporal_store(pk, reinterpret_cast<u32x2_t*>(p.y + (long)row * p.ldy + col));
    } else if (p.y) store4_bf(p.y + (long)row * p.ldy + col, y);
    if (p.y32) *reinterpret_cast<f32x4_t*>(p.y32 + (long)row * p.ldy32 + col) = y;
  }
  }
}


int launch_ln_fwd(const LnFwd& p, hipStream_t stream) {
  COOT_REQUIRE(p.x && (p.y || p.y32), "ln_fwd: null pointer");
  COOT_REQUIRE(p.D % 4 == 0 && p.D <= 4096 && p.D >= 8 && p.ldx % 4 == 0, "ln_fwd: D=%d unsupported (need D%%4==0, 8<=D<=4096)", p.D);
  if (p.R <= 0) return 0;
  dim3 grid((p.R + 3) / 4);
  const bool stream_variant = p.nt || p.max_wgs > 0;
  if (p.max_wgs > 0 && (int)grid.x > p.max_wgs) grid.x = p.max_wgs;
  // the input LayerNorm of a local network (the one kernel that reads the feature stream): HIP events around it, with the
  // bytes it must move (input once, bf16 xhat once) in the flops field
  void* ts = (p.x_f32 && !p.gain && p.y) ? timing_begin(TIMING_INLN, (double)p.R * p.D * 6.0, 0, stream) : nullptr;
  if (stream_variant) {
    if (p.D <= 1024) hipLaunchKernelGGL(ln_fwd_stream_kernel<4>, grid, dim3(256), 0, stream, p);
    else if (p.D <= 2048) hipLaunchKernelGGL(ln_fwd_stream_kernel<8>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(ln_fwd_stream_kernel<16>, grid, dim3(256), 0, stream, p);
  } else if (p.D <= 512) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, dim3(256), 0, stream, p);
  else if (p.D <= 1024) hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, dim3(256), 0, stream, p);
  else if (p.D <= 2048) hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(ln_fwd_kernel<16>, grid, dim3(256), 0, stream, p);
  timing_end(ts, stream);
  COOT_CHECK_LAUNCH("ln_fwd");
  return 0;
}

#define RUN_RED(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

static thread_local float* g_part_ws = nullptr;
static thread_local size_t g_part_floats = 0;
static thread_local size_t g_part_top = 0;  // floats reserved at the top of the workspace by deferred reductions
static thread_local size_t g_part_low = 0;   // floats at the start that deferred regions never touch
void set_partials_workspace(float* ws, size_t floats, size_t low_floats) { g_part_ws = ws; g_part_floats = floats; g_part_top = 0; g_part_low = low_floats; }
float* partials_workspace(size_t need) { return (g_part_ws && need + g_part_top <= g_part_floats) ? g_part_ws : nullptr; }

// ---- deferred column-sum reductions (rowops.h) ----------------------------------------------------------------
static thread_local bool g_defer = false;
static thread_local DeferArgs g_defer_args;

void colsum_defer_begin() { g_defer = true; g_defer_args.nseg = 0; g_part_top = 0; }
void colsum_defer_end() { g_defer = false; g_defer_args.nseg = 0; g_part_top = 0; }
float* partials_workspace_top(size_t need) {
  if (!g_defer || !g_part_ws || g_defer_args.nseg >= DEFER_MAX) return nullptr;
  need = (need + 63) & ~(size_t)63;
  if (g_part_low + g_part_top + need > g_part_floats) return nullptr;  // the first g_part_low floats stay with the immediate users
  g_part_top += need;
  return g_part_ws + (g_part_floats - g_part_top);
}
int colsum_defer_room() { return g_defer ? DEFER_MAX - g_defer_args.nseg : 0; }
bool colsum_defer_add(float* dst, const float* src, long ld, int nparts, int n, int overwrite) {
  if (!g_defer || g_defer_args.nseg >= DEFER_MAX) return false;
  g_defer_args.s[g_defer_args.nseg++] = DeferSeg{dst, src, ld, nparts, n, overwrite};
  return true;
}
__global__ __launch_bounds__(256) void colsum_defer_kernel(DeferArgs a) { colsum_defer_block(a, blockIdx.y, blockIdx.x); }
bool colsum_defer_take(DeferArgs* out, int max_cols) {
  const int n = g_defer_args.nseg;
  if (n == 0) return false;
  for (int i = 0; i < n; ++i) if (g_defer_args.s[i].n > max_cols) return false;  // (stays recorded: colsum_defer_flush runs it)
  *out = g_defer_args;
  g_defer_args.nseg = 0; g_part_top = 0;
  return true;
}
int colsum_defer_flush(hipStream_t stream) {
  const int n = g_defer_args.nseg;
  if (n == 0) { g_part_top = 0; return 0; }
  int maxn = 0;
  for (int i = 0; i < n; ++i) if (g_defer_args.s[i].n > maxn) maxn = g_defer_args.s[i].n;
  hipLaunchKernelGGL(colsum_defer_kernel, dim3((maxn + 15) / 16, n), dim3(256), 0, stream, g_defer_args);
  g_defer_args.nseg = 0; g_part_top = 0;
  COOT_CHECK_LAUNCH("colsum_defer");
  return 0;
}

// out_k[c] += sum_p ws[p*ld + k*seg + c] for up to three outputs k (seg columns each).
// grid = (column blocks of 64, part chunks): every block sums <= 64 parts (16 independent loads per thread), the
// few chunks per column are combined with atomics (<= 8 adds per address: no contention to speak of).
struct ReduceOuts { float* out[3]; int nout; int seg; };
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* ws, int nparts, long ld, int C, ReduceOuts ro, int parts_per_chunk) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), pg = threadIdx.x >> 6;
  const int p0 = blockIdx.y * parts_per_chunk, p1 = min(nparts, p0 + parts_per_chunk);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int p = p0 + pg;
    for (; p + 12 < p1; p += 16) {
      s0 += ws[(long)p * ld + c]; s1 += ws[(long)(p + 4) * ld + c]; s2 += ws[(long)(p + 8) * ld + c]; s3 += ws[(long)(p + 12) * ld + c];
    }
    for (; p < p1; p += 4) s0 += ws[(long)p * ld + c];
  }
  red[pg][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pg == 0 && c < C) {
    const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    const int k = c / ro.seg;
    float* o = ro.out[k];
    if (o) {
      if (gridDim.y == 1) o[c - k * ro.seg] += v; else acc_add(o + (c - k * ro.seg), v);
    }
  }
}
static int launch_reduce_multi(const float* ws, int nparts, long ld, ReduceOuts ro, hipStream_t stream) {
  const int C = ro.nout * ro.seg;
  if (C <= 0 || nparts <= 0) return 0;
  int chunks = (nparts + 63) / 64;
  if (chunks > 8) chunks = 8;
  const int ppc = (nparts + chunks - 1) / chunks;
  chunks = (nparts + ppc - 1) / ppc;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + 63) / 64, chunks), dim3(256), 0, stream, ws, nparts, ld, C, ro, ppc);
  COOT_CHECK_LAUNCH("reduce_partials");
  return 0;
}
int launch_reduce_partials(const float* ws, int nparts, long ld, int C, float* out, hipStream_t stream) {
  ReduceOuts ro; ro.out[0] = out; ro.out[1] = nullptr; ro.out[2] = nullptr; ro.nout = 1; ro.seg = C;
  return launch_reduce_multi(ws, nparts, ld, ro, stream);
}

// Backward:  xc = x - mean, s = std + eps, h = dy * gain
//   dx = (h - mean(h))/s - (sum h*xc)/s^2 * xc/((n-1)*std)      [second term 0 where std == 0]
//   dgain += dy * xc/s ; dbias += dy
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwd p) {
  __shared__ float red[3][4][NV * 256];
  const unsigned dkey = p.drop.thr ? drop_site_key(p.drop.seed, p.drop.seed_ptr, p.drop.site) : 0u;
  const unsigned dkey_m = (p.dxm && p.dxm_drop.thr) ? drop_site_key(p.dxm_drop.seed, p.dxm_drop.seed_ptr, p.dxm_drop.site) : 0u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = p.D, nch = D >> 2;
  f32x4_t ag[NV], ab[NV], ac[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { ag[i] = f32x4_t{0, 0, 0, 0}; ab[i] = ag[i]; ac[i] = ag[i]; }
  for (int row = blockIdx.x * 4 + wave; row < p.R; row += gridDim.x * 4) {
    f32x4_t x[NV], dy[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + 64 * i;
      x[i] = f32x4_t{0, 0, 0, 0}; dy[i] = x[i];
      if (c < nch) {
        const int col = c * 4;
        x[i] = load4(p.x, p.x_f32, (long)row * p.ldx + col);
        dy[i] = p.dy32 ? *reinterpret_cast<const f32x4_t*>(p.dy32 + (long)row * p.lddy32 + col)
                       : load4(p.dy, 0, (long)row * p.lddy + col);
        if (p.dy_add) { f32x4_t t = load4(p.dy_add, 0, (long)row * p.lddy_add + col); dy[i] += t; }
        if (p.drop.thr) {
          float sc[4];
          drop_scales_key<4>(dkey, (unsigned long long)row * D + col, p.drop.thr, p.drop.inv_keep, sc);
#pragma unroll
          for (int j = 0; j < 4; ++j) dy[i][j] *= sc[j];
        }
      }
      s += x[i][0] + x[i][1] + x[i][2] + x[i][3];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f, hs = 0.f, hx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + 64 * i;
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float d = x[i][j] - mean; x[i][j] = d; q += d * d;
          float h = dy[i][j] * p.gain[c * 4 + j];
          hs += h; hx += h * d;
        }
      }
    }
    q = wave_sum(q); hs = wave_sum(hs); hx = wave_sum(hx);
    const float stdv = sqrtf(q / (float)(D - 1));
    const float sv = stdv + kLnEps, rs = 1.0f / sv;
    const float hmean = hs / (float)D;
    const float k2 = stdv > 0.f ? hx * rs * rs / ((float)(D - 1) * stdv) : 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = lane + 64 * i;
      if (c >= nch) continue;
      const int col = c * 4;
      f32x4_t dx, dm;
      float msc[4] = {1.f, 1.f, 1.f, 1.f};
      if (p.dxm && p.dxm_drop.thr)
        drop_scales_key<4>(dkey_m, (unsigned long long)row * p.dxm_drop_ld + col, p.dxm_drop.thr,
                       p.dxm_drop.inv_keep, msc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float h = dy[i][j] * p.gain[col + j];
        dx[j] = (h - hmean) * rs - k2 * x[i][j];
        ag[i][j] += dy[i][j] * x[i][j] * rs;
        ab[i][j] += dy[i][j];
        const float m = dx[j] * msc[j];
        dm[j] = m;
        ac[i][j] += m;
      }
      if (p.dx) store4_bf(p.dx + (long)row * p.lddx + col, dx);
      if (p.dx32) *reinterpret_cast<f32x4_t*>(p.dx32 + (long)row * p.lddx32 + col) = dx;
      if (p.dxm) store4_bf(p.dxm + (long)row * p.lddxm + col, dm);
    }
  }
  // block reduce of the column partials (4 waves), then one atomic per column per block
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int idx = (lane + 64 * i) * 4 + j;
      red[0][wave][idx] = ag[i][j]; red[1][wave][idx] = ab[i][j]; red[2][wave][idx] = ac[i][j];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float g = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    float x = red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c];
    if (p.part_ws) {
      float* w = p.part_ws + (long)blockIdx.x * 3 * D;
      w[c] = g; w[D + c] = b; w[2 * D + c] = x;
    } else {
      if (p.dgain) acc_add(p.dgain + c, g);
      if (p.dbias) acc_add(p.dbias + c, b);
      if (p.dxcolsum) acc_add(p.dxcolsum + c, x);
    }
  }
}

int launch_ln_bwd(const LnBwd& p_in, hipStream_t stream) {
  LnBwd p = p_in;
  COOT_REQUIRE((p.dy || p.dy32) && p.x && p.gain, "ln_bwd: null pointer");
  COOT_REQUIRE(p.D % 4 == 0 && p.D <= 1024 && p.D >= 8, "ln_bwd: D=%d unsupported (need D%%4==0, 8<=D<=1024)", p.D);
  if (p.R <= 0) return 0;
  // one wave per row and a dependent load -> reduce -> store chain per row: the kernel is latency bound, so give every
  // wave only a few rows (256 blocks = 25 rows per wave at T = 25600 ran at 0.3 TB/s)
  int blocks = (p.R + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  p.part_ws = partials_workspace((size_t)blocks * 3 * p.D);
  if (!p.part_ws && blocks > 256) { blocks = 256; p.part_ws = partials_workspace((size_t)blocks * 3 * p.D); }
  // few rows (global networks): <= 128 atomics per address are cheaper than a second launch on the step's critical path
  if (p.R <= 512) p.part_ws = nullptr;
  if (p.D <= 512) hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(blocks), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(blocks), dim3(256), 0, stream, p);
  COOT_CHECK_LAUNCH("ln_bwd");
  if (p.part_ws) {  // dgain | dbias | colsum(dx) in one launch
    ReduceOuts ro; ro.out[0] = p.dgain; ro.out[1] = p.dbias; ro.out[2] = p.dxcolsum; ro.nout = 3; ro.seg = p.D;
    RUN_RED(launch_reduce_multi(p.part_ws, blocks, 3L * p.D, ro, stream));
  }
  return 0;
}

// ---- column sums of a bf16 matrix -----------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* x, long ldx, int R, int C, float* out, int rows_per_block, float* part) {
  // thread -> 2 columns (4-byte loads); block covers 512 columns x rows_per_block rows
  const int col = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (col >= C) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float a = 0.f, b = 0.f;
  for (int r = r0; r < r1; ++r) {
    unsigned u = *reinterpret_cast<const unsigned*>(x + (long)r * ldx + col);
    a += bflo(u); b += bfhi(u);
  }
  if (part) { part[(long)blockIdx.y * C + col] = a; if (col + 1 < C) part[(long)blockIdx.y * C + col + 1] = b; return; }
  acc_add(out + col, a);
  if (col + 1 < C) acc_add(out + col + 1, b);
}

int launch_colsum_bf16(const bf16_t* x, long ldx, int R, int C, float* out, hipStream_t stream) {
  COOT_REQUIRE(x && out && C % 2 == 0 && ldx % 2 == 0, "colsum: bad args");
  if (R <= 0) return 0;
  int rpb = 64;
  dim3 grid((C / 2 + 255) / 256, (R + rpb - 1) / rpb);
  float* part = R <= 512 ? nullptr : partials_workspace((size_t)grid.y * C);  // few rows: atomics, no second launch
  hipLaunchKernelGGL(colsum_bf16_kernel, grid, dim3(256), 0, stream, x, ldx, R, C, out, rpb, part);
  COOT_CHECK_LAUNCH("colsum_bf16");
  if (part) return launch_reduce_partials(part, (int)grid.y, C, C, out, stream);
  return 0;
}

// ---- weight cast / transpose (tiny; once per optimizer step) ---------------------------------
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* src, long lds, int R, int C, bf16_t* dst, long ldd,
                                                          int transpose, const float* colscale) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = ty; i < 32; i += 8) {
    int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < C) { v = src[(long)r * lds + c]; if (colscale) v *= colscale[c]; }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (!transpose) {
    for (int i = ty; i < 32; i += 8) {
      int r = r0 + i, c = c0 + tx;
      if (r < R && c < C) dst[(long)r * ldd + c] = f2bf(tile[i][tx]);
    }
  } else {
    for (int i = ty; i < 32; i += 8) {
      int c = c0 + i, r = r0 + tx;
      if (r < R && c < C) dst[(long)c * ldd + r] = f2bf(tile[tx][i]);
    }
  }
}

int launch_cast_weight(const float* src, long lds, int R, int C, bf16_t* dst, long ldd, int transpose,
                       const float* colscale, hipStream_t stream) {
  COOT_REQUIRE(src && dst, "cast_weight: null pointer");
  dim3 grid((C + 31) / 32, (R + 31) / 32);
  hipLaunchKernelGGL(cast_weight_kernel, grid, dim3(256), 0, stream, src, lds, R, C, dst, ldd, transpose, colscale);
  COOT_CHECK_LAUNCH("cast_weight");
  return 0;
}

__device__ __forceinline__ void matvec_rows(int blk, const float* W, long ldw, int N, int K, const float* v, const float* b, float* out) {
  const int lane = threadIdx.x & 63;
  const int n = blk * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains: the loads of a batch are in flight together
  int k = lane;
  for (; k + 192 < K; k += 256) {
    s0 += W[(long)n * ldw + k] * v[k]; s1 += W[(long)n * ldw + k + 64] * v[k + 64];
    s2 += W[(long)n * ldw + k + 128] * v[k + 128]; s3 += W[(long)n * ldw + k + 192] * v[k + 192];
  }
  for (; k < K; k += 64) s0 += W[(long)n * ldw + k] * v[k];
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if (lane == 0) out[n] = s + (b ? b[n] : 0.f);
}

// One thread per 8 consecutive bf16 of the DESTINATION (one 16-byte store): blockIdx.x -> (job, group) through the jobs'
// workgroup prefix.  The logical matrix is [n][k] = dst row, dst column (source [r][c] with (n, k) = (r, c), or (c, r) when
// transposed); k is the contiguous destination index in both layouts (row-major with ldd, or P48: fused.h p48_offset).
// History: a (max tiles, jobs) grid of 32 x 32 LDS-transposed tiles with 2-byte scattered stores: 21k workgroups for the
// 2.6M-parameter local network, 16 us per launch, four launches per step, two of them on the critical path.
struct PackTiles { int blk0[57]; };
__global__ __launch_bounds__(256) void pack_jobs_kernel(const float* P, char* wpack, PackJobs jobs, PackTiles pt) {
  if ((int)blockIdx.x >= pt.blk0[jobs.n]) {  // rider: the folded input-FC bias (was its own 12 us launch in front of the pack)
    matvec_rows((int)blockIdx.x - pt.blk0[jobs.n], jobs.mv.W, jobs.mv.ldw, jobs.mv.N, jobs.mv.K, jobs.mv.v, jobs.mv.b, jobs.mv.out);
    return;
  }
  int ji = 0;
  for (int t = 1; t < jobs.n; ++t) if ((int)blockIdx.x >= pt.blk0[t]) ji = t;
  const PackJob& jb = jobs.j[ji];
  const int Nn = jb.transpose ? jb.C : jb.R, Kk = jb.transpose ? jb.R : jb.C;  // logical [n][k]
  const int k8n = Kk >> 3;
  const long g = (long)(blockIdx.x - pt.blk0[ji]) * 256 + threadIdx.x;
  if (g >= (long)Nn * k8n) return;
  int n, k;
  long doff;
  if (jb.p48) {  // invert p48_offset for the group's first element: offset = 8 g
    const int l64 = (int)(g & 63), q = l64 >> 4, r = l64 & 15;
    long t = g >> 6;
    const int b = (int)(t % 3); t /= 3;
    const int kbn = Kk >> 5, kb = (int)(t % kbn), grp = (int)(t / kbn);
    n = grp * 48 + b * 16 + r; k = kb * 32 + q * 8;
    doff = g * 8;
  } else {
    n = (int)(g / k8n); k = (int)(g - (long)n * k8n) * 8;
    doff = (long)n * jb.ldd + k;
  }
  const float* src = P + jb.src_off;
  float v[8];
  if (!jb.transpose) {  // source row n, columns k .. k + 8 (contiguous)
    const float* sp = src + (long)n * jb.lds + k;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = sp[e];
    if (jb.colscale_off >= 0) {
      const float* cs = P + jb.colscale_off + k;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= cs[e];
    }
  } else {  // source rows k .. k + 8, column n
    const float* sp = src + (long)k * jb.lds + n;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = sp[(long)e * jb.lds];
    if (jb.colscale_off >= 0) {
      const float sc = P[jb.colscale_off + n];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= sc;
    }
  }
  bf16_t* dst = reinterpret_cast<bf16_t*>(wpack + jb.dst_byte_off);
  *reinterpret_cast<u32x4_t*>(dst + doff) = u32x4_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
}
int launch_pack_jobs(const float* P, void* wpack, const PackJobs& jobs, hipStream_t stream) {
  if (jobs.n <= 0 && !jobs.mv.W) return 0;
  PackTiles pt;
  int total = 0;
  for (int i = 0; i < jobs.n; ++i) {
    const PackJob& j = jobs.j[i];
    const int Nn = j.transpose ? j.C : j.R, Kk = j.transpose ? j.R : j.C;
    COOT_REQUIRE(Kk % 8 == 0 && (j.p48 ? (Nn % 48 == 0 && Kk % 32 == 0) : (j.ldd % 8 == 0)) && j.dst_byte_off % 16 == 0,
                 "pack: job %d: [%d x %d] not packable in 16-byte groups", i, Nn, Kk);
    pt.blk0[i] = total;
    total += (int)(((long)Nn * (Kk / 8) + 255) / 256);
  }
  pt.blk0[jobs.n] = total;
  if (jobs.mv.W) total += (jobs.mv.N + 3) / 4;
  hipLaunchKernelGGL(pack_jobs_kernel, dim3(total), dim3(256), 0, stream, P, (char*)wpack, jobs, pt);
  COOT_CHECK_LAUNCH("pack_jobs");
  return 0;
}

__global__ __launch_bounds__(256) void matvec_bias_kernel(const float* W, long ldw, int N, int K, const float* v, const float* b, float* out) {
  matvec_rows(blockIdx.x, W, ldw, N, K, v, b, out);
}

int launch_matvec_bias(const float* W, long ldw, int N, int K, const float* v, const float* b, float* out, hipStream_t stream) {
  COOT_REQUIRE(W && v && out, "matvec: null pointer");
  hipLaunchKernelGGL(matvec_bias_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, W, ldw, N, K, v, b, out);
  COOT_CHECK_LAUNCH("matvec_bias");
  return 0;
}

// thread = one column k for a slab of 8 rows n: coalesced along k, all 16 loads of the slab in flight at once (a 32-row
// serial loop was a 20 us chain of load latencies); dg0/db0 partials via atomics
constexpr int IPG_ROWS = 8;
__global__ __launch_bounds__(256) void infc_param_grads_kernel(const float* M, const float* W, const float* g0, const float* b0,
                                                               const float* c, int N, int K, float* dW, float* dg0, float* db0, float* db_in,
                                                               int overwrite) {
  if (db_in && blockIdx.x == 0 && blockIdx.y == 0)  // db_in += colsum(dh0) (was its own axpy launch)
    for (int n = threadIdx.x; n < N; n += 256) db_in[n] += c[n];
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const int n0 = blockIdx.y * IPG_ROWS;
  const float g = g0[k], b = b0[k];
  float m[IPG_ROWS], w[IPG_ROWS], d[IPG_ROWS];
#pragma unroll
  for (int i = 0; i < IPG_ROWS; ++i) {
    const bool on = n0 + i < N;
    const long o = (long)(on ? n0 + i : n0) * K + k;
    m[i] = on ? M[o] : 0.f; w[i] = on ? W[o] : 0.f; d[i] = (on && !overwrite) ? dW[o] : 0.f;
  }
  float sg = 0.f, sb = 0.f;
#pragma unroll
  for (int i = 0; i < IPG_ROWS; ++i) {
    if (n0 + i < N) {
      const float cn = c[n0 + i];
      dW[(long)(n0 + i) * K + k] = d[i] + m[i] * g + cn * b;
      sg += w[i] * m[i];
      sb += cn * w[i];
    }
  }
  acc_add(dg0 + k, sg);
  acc_add(db0 + k, sb);
}

int launch_infc_param_grads(const float* M, const float* W, const float* g0, const float* b0, const float* c,
                            int N, int K, float* dW, float* dg0, float* db0, float* db_in, hipStream_t stream, int overwrite) {
  hipLaunchKernelGGL(infc_param_grads_kernel, dim3((K + 255) / 256, (N + IPG_ROWS - 1) / IPG_ROWS), dim3(256), 0, stream, M, W, g0, b0, c, N, K,
                     dW, dg0, db0, db_in, overwrite);
  COOT_CHECK_LAUNCH("infc_param_grads");
  return 0;
}

__global__ void fill_f32_kernel(float* p, long n, float v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
int launch_fill_f32(float* p, long n, float v, hipStream_t stream) {
  if (n <= 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, stream, p, n, v);
  COOT_CHECK_LAUNCH("fill_f32");
  return 0;
}

__global__ void axpy_f32_kernel(float* y, const float* x, long n, float a) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) y[i] += a * x[i];
}
int launch_axpy_f32(float* y, const float* x, long n, float a, hipStream_t stream) {
  if (n <= 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpy_f32_kernel, dim3(blocks), dim3(256), 0, stream, y, x, n, a);
  COOT_CHECK_LAUNCH("axpy_f32");
  return 0;
}

}  // namespace coot

namespace coot {
COOT_DET_DEFINE_SETTER(rowops)
}  // namespace coot
