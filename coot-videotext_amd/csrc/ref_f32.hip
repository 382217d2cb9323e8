// fp32 REFERENCE MODE of a network, forward AND backward (coot_net_config.dtype = COOT_DTYPE_F32; SURVEY 7 "each: fwd + bwd, fp32
// reference mode and bf16 fast mode", 8b dtype enum).  TransformerLegacy.forward (nntrainer/models/transformer_legacy.py:200-288) in
// eval mode with every activation, weight and accumulation in fp32 — plain FMA kernels, exact erf GELU, no weight packs, no folded
// LayerNorm affine, no fusion: the op sequence of the reference, one kernel per op — and the derivative of exactly that sequence
// (the formulas of oracle/coot_oracle.py: ln_coot_bwd, mha_bwd, encoder_layer_bwd, genpool_bwd, net_bwd).
// A CHECKER, not a fast path: it separates logic errors from bf16 rounding (the bf16 path agrees with the reference to ~1e-3 of the
// output scale, this one to ~1e-6; gradients ~1e-2 against ~1e-5) and is never what bench.py times.  The forward keeps every
// intermediate in the caller's `saved` buffer, the backward reads them there.
#include "ref_f32.h"

#include <math.h>
#include <string.h>

#include "rowops.h"

namespace coot {
namespace {

// Y[m][n] (+)= epi(sum_k A(m, k) B(n, k) + b[n]);  A(m, k) = X[m][k] (x_t == 0) or X[k][m] (x_t == 1: a reduction over the ROWS of X,
// i.e. a weight gradient);  B(n, k) = W[n][k] (w_kn == 0) or W[k][n] (w_kn == 1)
struct RefGemm {
  const float* X; long ldx; int x_t; const float* W; long ldw; int w_kn; const float* b; int M, N, K;
  int act;                      // 0: none; 1: erf GELU (the pre-activation goes to Ypre when set); 2: times GELU'(aux[m][n])
  const float* aux; long ldaux; float* Ypre; long ldpre;
  const float* res; long ldres; // + res[m][n]
  const float* pe; int T0, L1, L2;  // + pe[pos(m)][n]
  float* Y; long ldy; int accum;    // accum: Y += (parameter gradients accumulate)
};
RefGemm mk_gemm(const float* X, long ldx, int x_t, const float* W, long ldw, int w_kn, const float* b, int M, int N, int K, float* Y, long ldy) {
  RefGemm g; memset(&g, 0, sizeof(g));
  g.X = X; g.ldx = ldx; g.x_t = x_t; g.W = W; g.ldw = ldw; g.w_kn = w_kn; g.b = b; g.M = M; g.N = N; g.K = K; g.Y = Y; g.ldy = ldy; g.L1 = g.L2 = 1;
  return g;
}
__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_exact(float v) {
  return 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * expf(-0.5f * v * v) * 0.39894228040143268f;
}
constexpr int RT = 64, RK = 16;
__global__ __launch_bounds__(256) void ref_gemm_kernel(RefGemm p) {
  __shared__ float Xs[RK][RT + 1], Ws[RK][RT + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * RT, n0 = blockIdx.x * RT;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += RK) {
    for (int e = threadIdx.x; e < RT * RK; e += 256) {
      {  // consecutive threads read consecutive addresses in either orientation
        const int r = p.x_t ? e % RT : e / RK, k = p.x_t ? e / RT : e % RK;
        const int m = m0 + r, kk = k0 + k;
        Xs[k][r] = (m < p.M && kk < p.K) ? (p.x_t ? p.X[(long)kk * p.ldx + m] : p.X[(long)m * p.ldx + kk]) : 0.f;
      }
      {
        const int r = p.w_kn ? e % RT : e / RK, k = p.w_kn ? e / RT : e % RK;
        const int n = n0 + r, kk = k0 + k;
        Ws[k][r] = (n < p.N && kk < p.K) ? (p.w_kn ? p.W[(long)kk * p.ldw + n] : p.W[(long)n * p.ldw + kk]) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RK; ++k) {
      float xv[4], wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { xv[i] = Xs[k][ty * 4 + i]; wv[i] = Ws[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xv[i], wv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= p.N) continue;
      float v = acc[i][j] + (p.b ? p.b[n] : 0.f);
      if (p.act == 1) {
        if (p.Ypre) p.Ypre[(long)m * p.ldpre + n] = v;
        v = gelu_exact(v);
      } else if (p.act == 2) {
        v *= gelu_grad_exact(p.aux[(long)m * p.ldaux + n]);
      }
      if (p.res) v += p.res[(long)m * p.ldres + n];
      if (p.pe) {
        const int pos = m < p.T0 ? m % p.L1 : (m - p.T0) % p.L2;
        v += p.pe[(long)pos * p.N + n];
      }
      float* y = p.Y + (long)m * p.ldy + n;
      *y = p.accum ? *y + v : v;
    }
  }
}
int ref_gemm(const RefGemm& p, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0) return 0;
  hipLaunchKernelGGL(ref_gemm_kernel, dim3((p.N + RT - 1) / RT, (p.M + RT - 1) / RT), dim3(256), 0, st, p);
  COOT_CHECK_LAUNCH("ref_gemm");
  return 0;
}

// out[n] += sum_m X[m][n]  (bias gradients)
__global__ __launch_bounds__(256) void ref_colsum_kernel(const float* X, long ldx, int M, int N, float* out, int rpb) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
  float a = 0.f;
  for (int r = r0; r < r1; ++r) a += X[(long)r * ldx + col];
  atomicAdd(out + col, a);
}
int ref_colsum(const float* X, long ldx, int M, int N, float* out, hipStream_t st) {
  if (M <= 0 || N <= 0) return 0;
  const int rpb = 128;
  hipLaunchKernelGGL(ref_colsum_kernel, dim3((N + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, st, X, ldx, M, N, out, rpb);
  COOT_CHECK_LAUNCH("ref_colsum");
  return 0;
}

// y = x * GELU'(pre)
__global__ void ref_mul_gelu_grad_kernel(const float* x, const float* pre, float* y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * gelu_grad_exact(pre[i]);
}

// MultiHeadAttention core (transformer_legacy.py:536-563): one thread per (sequence, head, query); keys beyond the sequence's length
// are filled with -INF = -32752 (nntrainer/typext.py:24) BEFORE the softmax, exactly as masked_fill does — a fully padded query row
// (with no valid key at all) gives the uniform distribution over the filled scores, like the reference.
struct RefAttn {
  const float *q, *k, *v; long ldq, ldk, ldv; float* o; long ldo;
  const long long* lens; int N, Lq, Lk, H, dh; long qrow0, krow0;  // sequence n: query rows qrow0 + n Lq + i, key rows krow0 + n Lk + j
  float scale;
  // backward: dO (query rows), outputs dq (query rows), dk / dv (key rows), per-(query row, head) softmax statistics
  const float* dout; long lddo; float *dq, *dk, *dv; long lddq, lddk, lddv; float *lse, *delta;
};
template <int DH>
__global__ __launch_bounds__(64) void ref_attn_kernel(RefAttn p) {
  const int n = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int i = blockIdx.y * 64 + threadIdx.x;
  if (i >= p.Lq) return;
  constexpr int dh = DH;
  const int nvalid = (int)p.lens[n];
  const float* qr = p.q + (p.qrow0 + (long)n * p.Lq + i) * p.ldq + h * dh;
  float qv[DH], ov[DH];
#pragma unroll
  for (int c = 0; c < dh; ++c) { qv[c] = qr[c]; ov[c] = 0.f; }
  // two passes (max, then exp-sum and the weighted values): the reference's softmax subtracts the row maximum
  float mx = -3.0e38f;
  for (int j = 0; j < p.Lk; ++j) {
    const float* kr = p.k + (p.krow0 + (long)n * p.Lk + j) * p.ldk + h * dh;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < dh; ++c) s = fmaf(qv[c], kr[c], s);
    s = j < nvalid ? s * p.scale : kMaskFill;
    mx = fmaxf(mx, s);
  }
  float den = 0.f;
  for (int j = 0; j < p.Lk; ++j) {
    const float* kr = p.k + (p.krow0 + (long)n * p.Lk + j) * p.ldk + h * dh;
    const float* vr = p.v + (p.krow0 + (long)n * p.Lk + j) * p.ldv + h * dh;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < dh; ++c) s = fmaf(qv[c], kr[c], s);
    s = j < nvalid ? s * p.scale : kMaskFill;
    const float e = expf(s - mx);
    den += e;
#pragma unroll
    for (int c = 0; c < dh; ++c) ov[c] = fmaf(e, vr[c], ov[c]);
  }
  float* orow = p.o + (p.qrow0 + (long)n * p.Lq + i) * p.ldo + h * dh;
  const float inv = 1.0f / den;
#pragma unroll
  for (int c = 0; c < dh; ++c) orow[c] = ov[c] * inv;
}
// Backward (oracle mha_bwd): P = softmax(S), dP_ij = <dO_i, V_j>, dS_ij = P_ij (dP_ij - sum_j P_ij dP_ij) / sqrt(dh) — a masked key's
// score is the constant fill: no gradient flows into its K row (masked_fill cuts the graph), its V row still receives P_ij dO_i.
// Query side: one thread per (sequence, head, query) — softmax statistics (lse, delta) and dQ.
template <int DH>
__global__ __launch_bounds__(64) void ref_attn_bwd_q_kernel(RefAttn p) {
  const int n = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int i = blockIdx.y * 64 + threadIdx.x;
  if (i >= p.Lq) return;
  const int nvalid = (int)p.lens[n];
  const long qrow = p.qrow0 + (long)n * p.Lq + i;
  const float* qr = p.q + qrow * p.ldq + h * DH;
  const float* dor = p.dout + qrow * p.lddo + h * DH;
  float qv[DH], dov[DH], dqv[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) { qv[c] = qr[c]; dov[c] = dor[c]; dqv[c] = 0.f; }
  float mx = -3.0e38f;
  for (int j = 0; j < p.Lk; ++j) {
    const float* kr = p.k + (p.krow0 + (long)n * p.Lk + j) * p.ldk + h * DH;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) s = fmaf(qv[c], kr[c], s);
    s = j < nvalid ? s * p.scale : kMaskFill;
    mx = fmaxf(mx, s);
  }
  float den = 0.f, dl = 0.f;
  for (int j = 0; j < p.Lk; ++j) {
    const float* kr = p.k + (p.krow0 + (long)n * p.Lk + j) * p.ldk + h * DH;
    const float* vr = p.v + (p.krow0 + (long)n * p.Lk + j) * p.ldv + h * DH;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) { s = fmaf(qv[c], kr[c], s); dp = fmaf(dov[c], vr[c], dp); }
    s = j < nvalid ? s * p.scale : kMaskFill;
    const float e = expf(s - mx);
    den += e; dl = fmaf(e, dp, dl);
  }
  const float lse = mx + logf(den), delta = dl / den;
  for (int j = 0; j < nvalid && j < p.Lk; ++j) {
    const float* kr = p.k + (p.krow0 + (long)n * p.Lk + j) * p.ldk + h * DH;
    const float* vr = p.v + (p.krow0 + (long)n * p.Lk + j) * p.ldv + h * DH;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) { s = fmaf(qv[c], kr[c], s); dp = fmaf(dov[c], vr[c], dp); }
    const float ds = expf(s * p.scale - lse) * (dp - delta) * p.scale;
#pragma unroll
    for (int c = 0; c < DH; ++c) dqv[c] = fmaf(ds, kr[c], dqv[c]);
  }
  float* dqr = p.dq + qrow * p.lddq + h * DH;
#pragma unroll
  for (int c = 0; c < DH; ++c) dqr[c] = dqv[c];
  p.lse[qrow * p.H + h] = lse; p.delta[qrow * p.H + h] = delta;
}
// Key side: one thread per (sequence, head, key): dK_j, dV_j summed over the queries in a fixed order (no atomics)
template <int DH>
__global__ __launch_bounds__(64) void ref_attn_bwd_kv_kernel(RefAttn p) {
  const int n = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int j = blockIdx.y * 64 + threadIdx.x;
  if (j >= p.Lk) return;
  const int nvalid = (int)p.lens[n];
  const long krow = p.krow0 + (long)n * p.Lk + j;
  const float* kr = p.k + krow * p.ldk + h * DH;
  const float* vr = p.v + krow * p.ldv + h * DH;
  float kv[DH], vv[DH], dkv[DH], dvv[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) { kv[c] = kr[c]; vv[c] = vr[c]; dkv[c] = 0.f; dvv[c] = 0.f; }
  const bool valid = j < nvalid;
  for (int i = 0; i < p.Lq; ++i) {
    const long qrow = p.qrow0 + (long)n * p.Lq + i;
    const float* qr = p.q + qrow * p.ldq + h * DH;
    const float* dor = p.dout + qrow * p.lddo + h * DH;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) { s = fmaf(qr[c], kv[c], s); dp = fmaf(dor[c], vv[c], dp); }
    s = valid ? s * p.scale : kMaskFill;
    const float pij = expf(s - p.lse[qrow * p.H + h]);
    const float ds = valid ? pij * (dp - p.delta[qrow * p.H + h]) * p.scale : 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) { dvv[c] = fmaf(pij, dor[c], dvv[c]); dkv[c] = fmaf(ds, qr[c], dkv[c]); }
  }
  float* dkr = p.dk + krow * p.lddk + h * DH;
  float* dvr = p.dv + krow * p.lddv + h * DH;
#pragma unroll
  for (int c = 0; c < DH; ++c) { dkr[c] = dkv[c]; dvr[c] = dvv[c]; }
}
#define REF_ATTN_DISPATCH(KERNEL, grid, a)                                                                    \
  switch ((a).dh) {                                                                                           \
    case 16: hipLaunchKernelGGL(KERNEL<16>, grid, dim3(64), 0, st, a); break;                                 \
    case 32: hipLaunchKernelGGL(KERNEL<32>, grid, dim3(64), 0, st, a); break;                                 \
    case 48: hipLaunchKernelGGL(KERNEL<48>, grid, dim3(64), 0, st, a); break;                                 \
    case 64: hipLaunchKernelGGL(KERNEL<64>, grid, dim3(64), 0, st, a); break;                                 \
    default: set_error("f32 reference attention: d_head = %d", (a).dh); return -1;                            \
  }
int ref_attn(const RefAttn& a, hipStream_t st) {
  const dim3 grid(a.N * a.H, (a.Lq + 63) / 64);
  REF_ATTN_DISPATCH(ref_attn_kernel, grid, a)
  COOT_CHECK_LAUNCH("ref_attn");
  return 0;
}
int ref_attn_bwd(const RefAttn& a, hipStream_t st) {
  const dim3 gq(a.N * a.H, (a.Lq + 63) / 64), gk(a.N * a.H, (a.Lk + 63) / 64);
  REF_ATTN_DISPATCH(ref_attn_bwd_q_kernel, gq, a)
  COOT_CHECK_LAUNCH("ref_attn_bwd_q");
  REF_ATTN_DISPATCH(ref_attn_bwd_kv_kernel, gk, a)
  COOT_CHECK_LAUNCH("ref_attn_bwd_kv");
  return 0;
}

// GenPool softmax over the sequence axis per channel + weighted sum (poolers.py:183-208): one thread per (sequence, channel)
__global__ void ref_genpool_kernel(const float* s, const float* z, const long long* lens, int N, int L, int D, long row0, float* pooled, long ldp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * D) return;
  const int n = idx / D, c = idx % D, nvalid = (int)lens[n];
  const float* sc = s + (row0 + (long)n * L) * D + c;
  const float* zc = z + (row0 + (long)n * L) * D + c;
  float mx = -3.0e38f;
  for (int l = 0; l < L; ++l) mx = fmaxf(mx, l < nvalid ? sc[(long)l * D] : kMaskFill);
  float den = 0.f, acc = 0.f;
  for (int l = 0; l < L; ++l) {
    const float e = expf((l < nvalid ? sc[(long)l * D] : kMaskFill) - mx);
    den += e; acc = fmaf(e, zc[(long)l * D], acc);
  }
  pooled[(long)n * ldp + c] = acc / den;
}
// its backward (oracle genpool_bwd): w = softmax_l(s), dz_l = dp w_l (every row), ds_l = w_l (dp z_l - dp pooled) on valid rows (the
// fill of a padded row is a constant)
__global__ void ref_genpool_bwd_kernel(const float* s, const float* z, const long long* lens, int N, int L, int D, long row0, const float* dpooled,
                                       long ldp, float* ds, float* dz) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * D) return;
  const int n = idx / D, c = idx % D, nvalid = (int)lens[n];
  const long off = (row0 + (long)n * L) * D + c;
  const float* sc = s + off;
  const float* zc = z + off;
  float mx = -3.0e38f;
  for (int l = 0; l < L; ++l) mx = fmaxf(mx, l < nvalid ? sc[(long)l * D] : kMaskFill);
  float den = 0.f, acc = 0.f;
  for (int l = 0; l < L; ++l) {
    const float e = expf((l < nvalid ? sc[(long)l * D] : kMaskFill) - mx);
    den += e; acc = fmaf(e, zc[(long)l * D], acc);
  }
  const float pooled = acc / den, dp = dpooled[(long)n * ldp + c], inv = 1.0f / den;
  for (int l = 0; l < L; ++l) {
    const float w = expf((l < nvalid ? sc[(long)l * D] : kMaskFill) - mx) * inv;
    dz[off + (long)l * D] = dp * w;
    ds[off + (long)l * D] = l < nvalid ? w * dp * (zc[(long)l * D] - pooled) : 0.f;
  }
}
// TemporalAvgPool "avg_special" (poolers.py:232-241): the sum runs over ALL L rows, padded ones included; divided by the length
__global__ void ref_avgpool_kernel(const float* z, const long long* lens, int N, int L, int D, float* pooled, long ldp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * D) return;
  const int n = idx / D, c = idx % D;
  float a = 0.f;
  for (int l = 0; l < L; ++l) a += z[((long)n * L + l) * D + c];
  pooled[(long)n * ldp + c] = a / (float)lens[n];
}
__global__ void ref_avgpool_bwd_kernel(const float* dpooled, long ldp, const long long* lens, int N, int L, int D, float* dz) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * L * D) return;
  const int c = (int)(idx % D);
  const int n = (int)(idx / ((long)L * D));
  dz[idx] = dpooled[(long)n * ldp + c] / (float)lens[n];
}
__global__ void ref_copy_rows_kernel(const float* src, long lds, int R, int C, float* dst, long ldd) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * C) return;
  const long r = idx / C; const int c = (int)(idx % C);
  dst[r * ldd + c] = src[r * lds + c];
}
int ref_copy_rows(const float* src, long lds, int R, int C, float* dst, long ldd, hipStream_t st) {
  if (R <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(ref_copy_rows_kernel, dim3((unsigned)(((long)R * C + 255) / 256)), dim3(256), 0, st, src, lds, R, C, dst, ldd);
  COOT_CHECK_LAUNCH("ref_copy");
  return 0;
}

// LayerNormalization backward (normalizations.py:98-101: unbiased std, eps added to the std; oracle ln_coot_bwd):
//   xc = x - mean, s = std + eps, h = dy gain;  dx = (h - mean(h)) / s - <h, xc> / s^2 * xc / ((D - 1) std)   [second term 0 where std == 0]
//   dgain += dy xc / s;  dbias += dy.   A workgroup owns LB_ROWS rows: row statistics by its four waves, then every thread walks its
//   columns down the rows (column sums in registers, one atomic per column and workgroup).
struct RefLnBwd { const float* x; long ldx; const float* dy; long lddy; const float* gain; int R, D; float* dx; long lddx; float* dgain; float* dbias; };
constexpr int LB_ROWS = 32, LB_MAXC = 16;
__global__ __launch_bounds__(256) void ref_ln_bwd_kernel(RefLnBwd p) {
  __shared__ float st_mean[LB_ROWS], st_rs[LB_ROWS], st_hmean[LB_ROWS], st_k2[LB_ROWS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * LB_ROWS, D = p.D;
  for (int rl = wave; rl < LB_ROWS; rl += 4) {
    const int row = r0 + rl;
    if (row >= p.R) continue;
    const float* xr = p.x + (long)row * p.ldx;
    const float* dyr = p.dy + (long)row * p.lddy;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f, hs = 0.f, hx = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float d = xr[c] - mean, h = dyr[c] * p.gain[c];
      q = fmaf(d, d, q); hs += h; hx = fmaf(h, d, hx);
    }
    q = wave_sum(q); hs = wave_sum(hs); hx = wave_sum(hx);
    const float stdv = sqrtf(q / (float)(D - 1)), rs = 1.0f / (stdv + kLnEps);
    if (lane == 0) {
      st_mean[rl] = mean; st_rs[rl] = rs; st_hmean[rl] = hs / (float)D;
      st_k2[rl] = stdv > 0.f ? hx * rs * rs / ((float)(D - 1) * stdv) : 0.f;
    }
  }
  __syncthreads();
  float dg[LB_MAXC], db[LB_MAXC];
#pragma unroll
  for (int i = 0; i < LB_MAXC; ++i) { dg[i] = 0.f; db[i] = 0.f; }
  for (int rl = 0; rl < LB_ROWS; ++rl) {
    const int row = r0 + rl;
    if (row >= p.R) break;
    const float mean = st_mean[rl], rs = st_rs[rl], hmean = st_hmean[rl], k2 = st_k2[rl];
#pragma unroll
    for (int i = 0; i < LB_MAXC; ++i) {
      const int c = threadIdx.x + 256 * i;
      if (c < D) {
        const float d = p.x[(long)row * p.ldx + c] - mean, dyv = p.dy[(long)row * p.lddy + c];
        if (p.dx) p.dx[(long)row * p.lddx + c] = (dyv * p.gain[c] - hmean) * rs - k2 * d;
        dg[i] = fmaf(dyv, d * rs, dg[i]); db[i] += dyv;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LB_MAXC; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < D) { atomicAdd(p.dgain + c, dg[i]); atomicAdd(p.dbias + c, db[i]); }
  }
}
int ref_ln_bwd(const float* x, long ldx, const float* dy, long lddy, const float* gain, int R, int D, float* dx, long lddx, float* dgain, float* dbias,
               hipStream_t st) {
  COOT_REQUIRE(D <= 256 * LB_MAXC, "f32 reference LayerNorm backward: D = %d (max %d)", D, 256 * LB_MAXC);
  if (R <= 0) return 0;
  RefLnBwd p{x, ldx, dy, lddy, gain, R, D, dx, lddx, dgain, dbias};
  hipLaunchKernelGGL(ref_ln_bwd_kernel, dim3((R + LB_ROWS - 1) / LB_ROWS), dim3(256), 0, st, p);
  COOT_CHECK_LAUNCH("ref_ln_bwd");
  return 0;
}

struct Bump32 {
  float* base; size_t cap, off = 0; bool overflow = false;
  Bump32(void* b, size_t bytes) : base((float*)b), cap(bytes / sizeof(float)) {}
  float* get(size_t n) { n = (n + 63) & ~(size_t)63; float* p = base ? base + off : nullptr; off += n; if (base && off > cap) overflow = true; return p; }
};

// everything the forward computes, kept for the backward.  A layer's rows: T tokens (self-attention layers) or N sequences (the
// context block: one query per sequence); its keys / values always come from T token rows.
struct LayerS { float *q; long ldq; float *k, *v; long ldkv; float *ctx, *r1, *z1, *hpre, *h, *r2, *out; };
struct RefSaved { float *xn, *hpre0, *z0; std::vector<LayerS> layers, ctx; float *hp_pre, *hp, *s; };
void layout_saved_f32(const RefNetDesc& d, long T, Bump32& A, RefSaved& S) {
  const size_t D = d.D, F = d.F, Nq = d.Nmax;
  S.xn = d.use_input_fc ? A.get((size_t)T * d.Din) : nullptr;
  S.hpre0 = d.use_input_fc ? A.get(T * D) : nullptr;
  S.z0 = A.get(T * D);
  S.layers.clear(); S.ctx.clear();
  for (int i = 0; i < d.num_layers; ++i) {
    LayerS L; float* qkv = A.get(T * 3 * D);
    L.q = qkv; L.k = qkv ? qkv + D : nullptr; L.v = qkv ? qkv + 2 * D : nullptr; L.ldq = L.ldkv = 3L * D;
    L.ctx = A.get(T * D); L.r1 = A.get(T * D); L.z1 = A.get(T * D); L.hpre = A.get(T * F); L.h = A.get(T * F); L.r2 = A.get(T * D); L.out = A.get(T * D);
    S.layers.push_back(L);
  }
  if (d.use_context)
    for (int i = 0; i < d.ctx_num_layers; ++i) {
      LayerS L; L.q = A.get(Nq * D); L.ldq = D; float* kv = A.get(T * 2 * D);
      L.k = kv; L.v = kv ? kv + D : nullptr; L.ldkv = 2L * D;
      L.ctx = A.get(Nq * D); L.r1 = A.get(Nq * D); L.z1 = A.get(Nq * D); L.hpre = A.get(Nq * F); L.h = A.get(Nq * F); L.r2 = A.get(Nq * D); L.out = A.get(Nq * D);
      S.ctx.push_back(L);
    }
  S.hp_pre = S.hp = S.s = nullptr;
  if (d.pooler == 0) { S.hp_pre = A.get(T * (size_t)d.pool_hidden); S.hp = A.get(T * (size_t)d.pool_hidden); S.s = A.get(T * D); }
}
// the backward's temporaries
struct RefScr { float *dz, *dx, *dr, *dr1, *dh, *dqkv, *dctx, *ds, *dxn, *lse, *delta, *n_buf[8]; };
void layout_scratch_f32(const RefNetDesc& d, long T, Bump32& A, RefScr& X) {
  const size_t D = d.D, F = d.F, Nq = d.Nmax;
  const size_t wide = F > (size_t)d.pool_hidden ? F : (size_t)d.pool_hidden;
  X.dz = A.get(T * D); X.dx = A.get(T * D); X.dr = A.get(T * D); X.dr1 = A.get(T * D); X.dh = A.get(T * wide); X.dqkv = A.get(T * 3 * D);
  X.dctx = A.get(T * D); X.ds = A.get(T * D); X.dxn = d.use_input_fc ? A.get((size_t)T * d.Din) : nullptr;
  X.lse = A.get((size_t)T * d.H); X.delta = A.get((size_t)T * d.H);
  for (int i = 0; i < 8; ++i) X.n_buf[i] = d.use_context ? A.get(Nq * (D > F ? D : F)) : nullptr;
}

#define RUNR(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

int ref_ln(const float* x, long ldx, int R, int D, const float* gain, const float* bias, const float* pe, int pe_L, float* y, hipStream_t st) {
  LnFwd l; l.x = x; l.x_f32 = 1; l.ldx = ldx; l.R = R; l.D = D; l.gain = gain; l.bias = bias; l.pe = pe; l.pe_L = pe_L; l.y32 = y; l.ldy32 = D;
  return launch_ln_fwd(l, st);
}

// one TransformerEncoderLayer (transformer_legacy.py:420-438), post-LN sublayers; xq [Rq, D] queries (self: == xkv)
int ref_layer(const RefNetDesc& d, const float* P, const RefLayerP& lp, const float* xq, int Rq, const float* xkv, int Rkv, bool self,
              const RefSegs& sg, const LayerS& S, hipStream_t st) {
  const int D = d.D, F = d.F, H = d.H, dh = D / H;
  if (self) {
    RUNR(ref_gemm(mk_gemm(xq, D, 0, P + lp.wqkv, D, 0, P + lp.bq, Rq, 3 * D, D, S.q, 3L * D), st));
  } else {
    RUNR(ref_gemm(mk_gemm(xq, D, 0, P + lp.wqkv, D, 0, P + lp.bq, Rq, D, D, S.q, (long)D), st));
    RUNR(ref_gemm(mk_gemm(xkv, D, 0, P + lp.wqkv + (size_t)D * D, D, 0, P + lp.bk, Rkv, 2 * D, D, S.k, 2L * D), st));
  }
  long qrow = 0, krow = 0;
  for (int s = 0; s < sg.n; ++s) {
    RefAttn a; memset(&a, 0, sizeof(a));
    a.q = S.q; a.k = S.k; a.v = S.v; a.ldq = S.ldq; a.ldk = a.ldv = S.ldkv; a.o = S.ctx; a.ldo = D; a.lens = sg.lens[s]; a.N = sg.N[s];
    a.Lq = self ? sg.L[s] : 1; a.Lk = sg.L[s]; a.H = H; a.dh = dh; a.qrow0 = qrow; a.krow0 = krow; a.scale = 1.0f / sqrtf((float)dh);
    RUNR(ref_attn(a, st));
    qrow += (long)sg.N[s] * a.Lq; krow += (long)sg.N[s] * sg.L[s];
  }
  {
    RefGemm g = mk_gemm(S.ctx, D, 0, P + lp.wo, D, 0, P + lp.bo, Rq, D, D, S.r1, (long)D);
    g.res = xq; g.ldres = D;
    RUNR(ref_gemm(g, st));
  }
  RUNR(ref_ln(S.r1, D, Rq, D, P + lp.ln1g, P + lp.ln1b, nullptr, 1, S.z1, st));
  {
    RefGemm g = mk_gemm(S.z1, D, 0, P + lp.w1, D, 0, P + lp.b1, Rq, F, D, S.h, (long)F);
    g.act = 1; g.Ypre = S.hpre; g.ldpre = F;
    RUNR(ref_gemm(g, st));
  }
  {
    RefGemm g = mk_gemm(S.h, F, 0, P + lp.w2, F, 0, P + lp.b2, Rq, D, F, S.r2, (long)D);
    g.res = S.z1; g.ldres = D;
    RUNR(ref_gemm(g, st));
  }
  return ref_ln(S.r2, D, Rq, D, P + lp.ln2g, P + lp.ln2b, nullptr, 1, S.out, st);
}

// its backward (oracle encoder_layer_bwd + mha_bwd).  dout [Rq, D] (row stride lddout): gradient wrt the layer's output.  Writes
// dxq [Rq, D] = gradient wrt the query-side input (residuals included) — for a self-attention layer the key / value side's share
// is part of it — and, for the context block, ADDS the key / value side's gradient into dz_acc [Rkv, D].  b1 / b2: two [Rq, max(D, F)]
// temporaries, dr / dr1 / dctx [Rq, D], dqkv [Rkv, 3 D].
struct LayerTmp { float *dr, *dh, *dz1, *dr1, *dctx, *dq; long lddq; float *dk, *dv; long lddkv; };
int ref_layer_bwd(const RefNetDesc& d, const float* P, float* G, const RefLayerP& lp, const float* xq, int Rq, const float* xkv, int Rkv, bool self,
                  const RefSegs& sg, const LayerS& S, const float* dout, long lddout, const LayerTmp& W, const RefScr& X, float* dxq, float* dz_acc,
                  hipStream_t st) {
  const int D = d.D, F = d.F, H = d.H, dh = D / H;
  // out = LN2(r2)
  RUNR(ref_ln_bwd(S.r2, D, dout, lddout, P + lp.ln2g, Rq, D, W.dr, D, G + lp.ln2g, G + lp.ln2b, st));
  // r2 = h W2^T + b2 + z1
  { RefGemm g = mk_gemm(W.dr, D, 1, S.h, F, 1, nullptr, D, F, Rq, G + lp.w2, (long)F); g.accum = 1; RUNR(ref_gemm(g, st)); }
  RUNR(ref_colsum(W.dr, D, Rq, D, G + lp.b2, st));
  { RefGemm g = mk_gemm(W.dr, D, 0, P + lp.w2, F, 1, nullptr, Rq, F, D, W.dh, (long)F); g.act = 2; g.aux = S.hpre; g.ldaux = F; RUNR(ref_gemm(g, st)); }
  // hpre = z1 W1^T + b1
  { RefGemm g = mk_gemm(W.dh, F, 1, S.z1, D, 1, nullptr, F, D, Rq, G + lp.w1, (long)D); g.accum = 1; RUNR(ref_gemm(g, st)); }
  RUNR(ref_colsum(W.dh, F, Rq, F, G + lp.b1, st));
  { RefGemm g = mk_gemm(W.dh, F, 0, P + lp.w1, D, 1, nullptr, Rq, D, F, W.dz1, (long)D); g.res = W.dr; g.ldres = D; RUNR(ref_gemm(g, st)); }
  // z1 = LN1(r1)
  RUNR(ref_ln_bwd(S.r1, D, W.dz1, D, P + lp.ln1g, Rq, D, W.dr1, D, G + lp.ln1g, G + lp.ln1b, st));
  // r1 = ctx Wo^T + bo + xq
  { RefGemm g = mk_gemm(W.dr1, D, 1, S.ctx, D, 1, nullptr, D, D, Rq, G + lp.wo, (long)D); g.accum = 1; RUNR(ref_gemm(g, st)); }
  RUNR(ref_colsum(W.dr1, D, Rq, D, G + lp.bo, st));
  RUNR(ref_gemm(mk_gemm(W.dr1, D, 0, P + lp.wo, D, 1, nullptr, Rq, D, D, W.dctx, (long)D), st));
  // attention core
  long qrow = 0, krow = 0;
  for (int s = 0; s < sg.n; ++s) {
    RefAttn a; memset(&a, 0, sizeof(a));
    a.q = S.q; a.k = S.k; a.v = S.v; a.ldq = S.ldq; a.ldk = a.ldv = S.ldkv; a.lens = sg.lens[s]; a.N = sg.N[s];
    a.Lq = self ? sg.L[s] : 1; a.Lk = sg.L[s]; a.H = H; a.dh = dh; a.qrow0 = qrow; a.krow0 = krow; a.scale = 1.0f / sqrtf((float)dh);
    a.dout = W.dctx; a.lddo = D; a.dq = W.dq; a.lddq = W.lddq; a.dk = W.dk; a.dv = W.dv; a.lddk = a.lddv = W.lddkv; a.lse = X.lse; a.delta = X.delta;
    RUNR(ref_attn_bwd(a, st));
    qrow += (long)sg.N[s] * a.Lq; krow += (long)sg.N[s] * sg.L[s];
  }
  if (self) {  // q | k | v = x Wqkv^T + b
    { RefGemm g = mk_gemm(W.dq, 3L * D, 1, xq, D, 1, nullptr, 3 * D, D, Rq, G + lp.wqkv, (long)D); g.accum = 1; RUNR(ref_gemm(g, st)); }
    RUNR(ref_colsum(W.dq, 3L * D, Rq, 3 * D, G + lp.bq, st));
    RefGemm g = mk_gemm(W.dq, 3L * D, 0, P + lp.wqkv, D, 1, nullptr, Rq, D, 3 * D, dxq, (long)D);
    g.res = W.dr1; g.ldres = D;
    RUNR(ref_gemm(g, st));
  } else {
    { RefGemm g = mk_gemm(W.dq, D, 1, xq, D, 1, nullptr, D, D, Rq, G + lp.wqkv, (long)D); g.accum = 1; RUNR(ref_gemm(g, st)); }
    RUNR(ref_colsum(W.dq, D, Rq, D, G + lp.bq, st));
    { RefGemm g = mk_gemm(W.dq, D, 0, P + lp.wqkv, D, 1, nullptr, Rq, D, D, dxq, (long)D); g.res = W.dr1; g.ldres = D; RUNR(ref_gemm(g, st)); }
    { RefGemm g = mk_gemm(W.dk, 2L * D, 1, xkv, D, 1, nullptr, 2 * D, D, Rkv, G + lp.wqkv + (size_t)D * D, (long)D); g.accum = 1; RUNR(ref_gemm(g, st)); }
    RUNR(ref_colsum(W.dk, 2L * D, Rkv, 2 * D, G + lp.bk, st));
    RefGemm g = mk_gemm(W.dk, 2L * D, 0, P + lp.wqkv + (size_t)D * D, D, 1, nullptr, Rkv, D, 2 * D, dz_acc, (long)D);
    g.accum = 1;
    RUNR(ref_gemm(g, st));
  }
  return 0;
}

}  // namespace

size_t ref_f32_workspace_bytes(const RefNetDesc& d, long T) {
  Bump32 A(nullptr, 0); RefSaved S; layout_saved_f32(d, T, A, S);
  return A.off * sizeof(float) + 1024;
}
size_t ref_f32_scratch_bytes(const RefNetDesc& d, long T) {
  Bump32 A(nullptr, 0); RefScr X; layout_scratch_f32(d, T, A, X);
  return A.off * sizeof(float) + 1024;
}

int ref_f32_forward(const RefNetDesc& d, const float* P, const float* pe, const float* feats, const float* feats2, const RefSegs& sg,
                    const float* hidden, float* pooled, float* per_token, void* ws, size_t ws_bytes, hipStream_t st) {
  const int D = d.D, Din = d.Din;
  long T = 0; int Ntot = 0;
  for (int s = 0; s < sg.n; ++s) { T += (long)sg.N[s] * sg.L[s]; Ntot += sg.N[s]; }
  Bump32 A(ws, ws_bytes); RefSaved S; layout_saved_f32(d, T, A, S);
  COOT_REQUIRE(!A.overflow && Ntot <= d.Nmax, "net_fwd (f32 reference mode): saved buffer too small (%zu bytes)", ws_bytes);
  const long T0 = (long)sg.N[0] * sg.L[0];
  const int out_dim = D * (d.use_context ? 2 : 1);
  // ---- input: LayerNorm (own gain / bias: nothing is folded here) [+ Linear + GELU] + positional encoding (transformer_legacy.py:222-241)
  if (d.use_input_fc) {
    RUNR(ref_ln(feats, Din, (int)T0, Din, P + d.n_gain, P + d.n_bias, nullptr, 1, S.xn, st));
    if (sg.n > 1) RUNR(ref_ln(feats2, Din, (int)(T - T0), Din, P + d.n_gain, P + d.n_bias, nullptr, 1, S.xn + T0 * Din, st));
    RefGemm g = mk_gemm(S.xn, Din, 0, P + d.in_w, Din, 0, P + d.in_b, (int)T, D, Din, S.z0, (long)D);
    g.act = 1; g.Ypre = S.hpre0; g.ldpre = D; g.pe = pe; g.T0 = (int)T0; g.L1 = sg.L[0]; g.L2 = sg.n > 1 ? sg.L[1] : sg.L[0];
    RUNR(ref_gemm(g, st));
  } else {
    RUNR(ref_ln(feats, Din, (int)T0, Din, P + d.n_gain, P + d.n_bias, pe, sg.L[0], S.z0, st));
    if (sg.n > 1) RUNR(ref_ln(feats2, Din, (int)(T - T0), Din, P + d.n_gain, P + d.n_bias, pe, sg.L[1], S.z0 + T0 * D, st));
  }
  const float* z = S.z0;
  for (int i = 0; i < d.num_layers; ++i) {
    RUNR(ref_layer(d, P, d.layers[i], z, (int)T, z, (int)T, true, sg, S.layers[i], st));
    z = S.layers[i].out;
  }
  if (per_token) RUNR(ref_copy_rows(z, D, (int)T0, D, per_token, D, st));
  // ---- context block: one query per sequence, keys / values = the encoder output (:251-267)
  if (d.use_context) {
    COOT_REQUIRE(sg.n == 1 && hidden, "net_fwd (f32 reference mode): context networks take one segment and a hidden state");
    const float* cq = hidden;
    for (int i = 0; i < d.ctx_num_layers; ++i) {
      RUNR(ref_layer(d, P, d.ctx[i], cq, Ntot, z, (int)T, false, sg, S.ctx[i], st));
      cq = S.ctx[i].out;
    }
    RUNR(ref_copy_rows(cq, D, Ntot, D, pooled + D, out_dim, st));
  }
  // ---- pooling
  if (d.pooler == 0) {
    const int Hp = d.pool_heads, PH = d.pool_hidden, dhp = PH / Hp, dop = D / Hp;
    for (int h = 0; h < Hp; ++h) {
      // a_h = GELU(z W1[h] + b1[h])  (W1[h]: [D, dhp]);  s_h = a_h W2[h] + b2[h]  (W2[h]: [dhp, dop]) -> channels [h dop, (h + 1) dop)
      RefGemm g = mk_gemm(z, D, 0, P + d.pw1 + (size_t)h * D * dhp, dhp, 1, P + d.pb1 + (size_t)h * dhp, (int)T, dhp, D, S.hp + h * dhp, (long)PH);
      g.act = 1; g.Ypre = S.hp_pre + h * dhp; g.ldpre = PH;
      RUNR(ref_gemm(g, st));
      RUNR(ref_gemm(mk_gemm(S.hp + h * dhp, PH, 0, P + d.pw2 + (size_t)h * dhp * dop, dop, 1, P + d.pb2 + (size_t)h * dop, (int)T, dop, dhp, S.s + h * dop, (long)D), st));
    }
    long row = 0; int n0 = 0;
    for (int s = 0; s < sg.n; ++s) {
      hipLaunchKernelGGL(ref_genpool_kernel, dim3((sg.N[s] * D + 255) / 256), dim3(256), 0, st, (const float*)S.s, z, sg.lens[s], sg.N[s], sg.L[s], D,
                         row, pooled + (size_t)n0 * out_dim, (long)out_dim);
      COOT_CHECK_LAUNCH("ref_genpool");
      row += (long)sg.N[s] * sg.L[s]; n0 += sg.N[s];
    }
  } else {
    hipLaunchKernelGGL(ref_avgpool_kernel, dim3((sg.N[0] * D + 255) / 256), dim3(256), 0, st, z, sg.lens[0], sg.N[0], sg.L[0], D, pooled, (long)out_dim);
    COOT_CHECK_LAUNCH("ref_avgpool");
  }
  return 0;
}

int ref_f32_backward(const RefNetDesc& d, const float* P, float* G, const float* feats, const float* feats2, const RefSegs& sg,
                     const float* hidden, const float* dpooled, float* dhidden, float* dfeats, void* ws, size_t ws_bytes, void* scratch,
                     size_t scratch_bytes, hipStream_t st) {
  const int D = d.D, Din = d.Din, F = d.F;
  long T = 0; int Ntot = 0;
  for (int s = 0; s < sg.n; ++s) { T += (long)sg.N[s] * sg.L[s]; Ntot += sg.N[s]; }
  Bump32 A(ws, ws_bytes); RefSaved S; layout_saved_f32(d, T, A, S);
  Bump32 AX(scratch, scratch_bytes); RefScr X; layout_scratch_f32(d, T, AX, X);
  COOT_REQUIRE(!A.overflow && !AX.overflow && Ntot <= d.Nmax, "net_bwd (f32 reference mode): saved / scratch buffer too small (%zu / %zu bytes)", ws_bytes,
               scratch_bytes);
  const long T0 = (long)sg.N[0] * sg.L[0];
  const int out_dim = D * (d.use_context ? 2 : 1);
  const float* zL = d.num_layers > 0 ? S.layers[d.num_layers - 1].out : S.z0;
  // ---- pooling: X.dz = gradient wrt the encoder output
  if (d.pooler == 0) {
    const int Hp = d.pool_heads, PH = d.pool_hidden, dhp = PH / Hp, dop = D / Hp;
    long row = 0; int n0 = 0;
    for (int s = 0; s < sg.n; ++s) {
      hipLaunchKernelGGL(ref_genpool_bwd_kernel, dim3((sg.N[s] * D + 255) / 256), dim3(256), 0, st, (const float*)S.s, zL, sg.lens[s], sg.N[s], sg.L[s], D,
                         row, dpooled + (size_t)n0 * out_dim, (long)out_dim, X.ds, X.dz);
      COOT_CHECK_LAUNCH("ref_genpool_bwd");
      row += (long)sg.N[s] * sg.L[s]; n0 += sg.N[s];
    }
    for (int h = 0; h < Hp; ++h) {
      const float* W1 = P + d.pw1 + (size_t)h * D * dhp;    // [D, dhp]
      const float* W2 = P + d.pw2 + (size_t)h * dhp * dop;  // [dhp, dop]
      // s_h = a_h W2 + b2
      { RefGemm g = mk_gemm(S.hp + h * dhp, PH, 1, X.ds + h * dop, D, 1, nullptr, dhp, dop, (int)T, G + d.pw2 + (size_t)h * dhp * dop, (long)dop); g.accum = 1; RUNR(ref_gemm(g, st)); }
      RUNR(ref_colsum(X.ds + h * dop, D, (int)T, dop, G + d.pb2 + (size_t)h * dop, st));
      { RefGemm g = mk_gemm(X.ds + h * dop, D, 0, W2, dop, 0, nullptr, (int)T, dhp, dop, X.dh, (long)dhp); g.act = 2; g.aux = S.hp_pre + h * dhp; g.ldaux = PH; RUNR(ref_gemm(g, st)); }
      // hp_pre_h = z W1 + b1
      { RefGemm g = mk_gemm(zL, D, 1, X.dh, dhp, 1, nullptr, D, dhp, (int)T, G + d.pw1 + (size_t)h * D * dhp, (long)dhp); g.accum = 1; RUNR(ref_gemm(g, st)); }
      RUNR(ref_colsum(X.dh, dhp, (int)T, dhp, G + d.pb1 + (size_t)h * dhp, st));
      { RefGemm g = mk_gemm(X.dh, dhp, 0, W1, dhp, 0, nullptr, (int)T, D, dhp, X.dz, (long)D); g.accum = 1; RUNR(ref_gemm(g, st)); }
    }
  } else {
    hipLaunchKernelGGL(ref_avgpool_bwd_kernel, dim3((unsigned)(((long)T * D + 255) / 256)), dim3(256), 0, st, dpooled, (long)out_dim, sg.lens[0], sg.N[0],
                       sg.L[0], D, X.dz);
    COOT_CHECK_LAUNCH("ref_avgpool_bwd");
  }
  // ---- context block (its keys / values are the encoder output: their gradient adds into X.dz)
  if (d.use_context) {
    COOT_REQUIRE(sg.n == 1 && hidden, "net_bwd (f32 reference mode): context networks take one segment and a hidden state");
    const float* dcq = dpooled + D; long lddcq = out_dim;
    for (int i = d.ctx_num_layers - 1; i >= 0; --i) {
      LayerTmp W; W.dr = X.n_buf[0]; W.dh = X.n_buf[1]; W.dz1 = X.n_buf[2]; W.dr1 = X.n_buf[3]; W.dctx = X.n_buf[4]; W.dq = X.n_buf[5]; W.lddq = D;
      W.dk = X.dqkv; W.dv = X.dqkv + D; W.lddkv = 2L * D;
      float* dxq = X.n_buf[6 + (i & 1)];
      const float* xq = i == 0 ? hidden : S.ctx[i - 1].out;
      RUNR(ref_layer_bwd(d, P, G, d.ctx[i], xq, Ntot, zL, (int)T, false, sg, S.ctx[i], dcq, lddcq, W, X, dxq, X.dz, st));
      dcq = dxq; lddcq = D;
    }
    if (dhidden) RUNR(ref_copy_rows(dcq, lddcq, Ntot, D, dhidden, D, st));
  }
  // ---- self-attention encoder layers
  for (int i = d.num_layers - 1; i >= 0; --i) {
    LayerTmp W; W.dr = X.dr; W.dh = X.dh; W.dz1 = X.dx; W.dr1 = X.dr1; W.dctx = X.dctx; W.dq = X.dqkv; W.lddq = 3L * D;
    W.dk = X.dqkv + D; W.dv = X.dqkv + 2 * D; W.lddkv = 3L * D;
    const float* xin = i == 0 ? S.z0 : S.layers[i - 1].out;
    // (the layer's output gradient X.dz is dead once LN2's backward has read it: the input gradient goes back into it)
    RUNR(ref_layer_bwd(d, P, G, d.layers[i], xin, (int)T, xin, (int)T, true, sg, S.layers[i], X.dz, D, W, X, X.dz, nullptr, st));
  }
  // ---- input
  if (d.use_input_fc) {
    hipLaunchKernelGGL(ref_mul_gelu_grad_kernel, dim3((unsigned)(((long)T * D + 255) / 256)), dim3(256), 0, st, (const float*)X.dz, (const float*)S.hpre0, X.dr, (long)T * D);
    COOT_CHECK_LAUNCH("ref_mul_gelu_grad");
    { RefGemm g = mk_gemm(X.dr, D, 1, S.xn, Din, 1, nullptr, D, Din, (int)T, G + d.in_w, (long)Din); g.accum = 1; RUNR(ref_gemm(g, st)); }
    RUNR(ref_colsum(X.dr, D, (int)T, D, G + d.in_b, st));
    RUNR(ref_gemm(mk_gemm(X.dr, D, 0, P + d.in_w, Din, 1, nullptr, (int)T, Din, D, X.dxn, (long)Din), st));
    COOT_REQUIRE(!dfeats, "net_bwd (f32 reference mode): dfeats is only available for networks without input_fc");
    RUNR(ref_ln_bwd(feats, Din, X.dxn, Din, P + d.n_gain, (int)T0, Din, nullptr, 0, G + d.n_gain, G + d.n_bias, st));
    if (sg.n > 1) RUNR(ref_ln_bwd(feats2, Din, X.dxn + T0 * Din, Din, P + d.n_gain, (int)(T - T0), Din, nullptr, 0, G + d.n_gain, G + d.n_bias, st));
  } else {
    RUNR(ref_ln_bwd(feats, Din, X.dz, D, P + d.n_gain, (int)T0, Din, dfeats, Din, G + d.n_gain, G + d.n_bias, st));
    if (sg.n > 1) RUNR(ref_ln_bwd(feats2, Din, X.dz + T0 * D, D, P + d.n_gain, (int)(T - T0), Din, dfeats ? dfeats + T0 * Din : nullptr, Din, G + d.n_gain,
                                  G + d.n_bias, st));
  }
  (void)F;
  return 0;
}

}  // namespace coot
