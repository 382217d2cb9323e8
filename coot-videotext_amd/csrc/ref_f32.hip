// fp32 REFERENCE MODE of a network forward (coot_net_config.dtype = COOT_DTYPE_F32; SURVEY 7 "each: fp32 reference mode and bf16
// fast mode", 8b dtype enum).  TransformerLegacy.forward (nntrainer/models/transformer_legacy.py:200-288) in eval mode with every
// activation, weight and accumulation in fp32 — plain FMA kernels, exact erf GELU, no weight packs, no folded LayerNorm affine, no
// fusion: the op sequence of the reference, one kernel per op.  A CHECKER, not a fast path: it separates logic errors from bf16
// rounding (the bf16 path agrees with the reference to ~1e-3 of the output scale, this one to ~1e-6), it is never what bench.py
// times, and it is forward-only (coot_net_bwd refuses dtype F32; gradients are pinned by the fp64 oracle and the reference's fixtures).
#include "ref_f32.h"

#include <math.h>

#include "rowops.h"

namespace coot {
namespace {

// Y[m][n] = act(sum_k X[m][k] Wt(n, k) + b[n]) (+ res[m][n]) (+ pe[pos(m)][n]);  W is [N][K] (w_kn == 0) or [K][N] (w_kn == 1)
struct RefLinear {
  const float* X; long ldx; const float* W; long ldw; int w_kn; const float* b; int M, N, K; int act;  // act 1: erf GELU
  const float* res; long ldres; const float* pe; int T0, L1, L2; float* Y; long ldy;
};
constexpr int RT = 64, RK = 16;
__global__ __launch_bounds__(256) void ref_linear_kernel(RefLinear p) {
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
      const int r = e / RK, k = e % RK;
      const int m = m0 + r, n = n0 + r, kk = k0 + k;
      Xs[k][r] = (m < p.M && kk < p.K) ? p.X[(long)m * p.ldx + kk] : 0.f;
      Ws[k][r] = (n < p.N && kk < p.K) ? (p.w_kn ? p.W[(long)kk * p.ldw + n] : p.W[(long)n * p.ldw + kk]) : 0.f;
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
      if (p.act == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
      if (p.res) v += p.res[(long)m * p.ldres + n];
      if (p.pe) {
        const int pos = m < p.T0 ? m % p.L1 : (m - p.T0) % p.L2;
        v += p.pe[(long)pos * p.N + n];
      }
      p.Y[(long)m * p.ldy + n] = v;
    }
  }
}
int ref_linear(const RefLinear& p, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0) return 0;
  hipLaunchKernelGGL(ref_linear_kernel, dim3((p.N + RT - 1) / RT, (p.M + RT - 1) / RT), dim3(256), 0, st, p);
  COOT_CHECK_LAUNCH("ref_linear");
  return 0;
}

// MultiHeadAttention core (transformer_legacy.py:536-563): one thread per (sequence, head, query); keys beyond the sequence's length
// are filled with -INF = -32752 (nntrainer/typext.py:24) BEFORE the softmax, exactly as masked_fill does — a fully padded query row
// (with no valid key at all) gives the uniform distribution over the filled scores, like the reference.
struct RefAttn {
  const float *q, *k, *v; long ldq, ldk, ldv; float* o; long ldo;
  const long long* lens; int N, Lq, Lk, H, dh; long qrow0, krow0;  // sequence n: query rows qrow0 + n Lq + i, key rows krow0 + n Lk + j
  float scale;
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
int ref_attn(const RefAttn& a, hipStream_t st) {
  const dim3 grid(a.N * a.H, (a.Lq + 63) / 64);
  switch (a.dh) {
    case 16: hipLaunchKernelGGL(ref_attn_kernel<16>, grid, dim3(64), 0, st, a); break;
    case 32: hipLaunchKernelGGL(ref_attn_kernel<32>, grid, dim3(64), 0, st, a); break;
    case 48: hipLaunchKernelGGL(ref_attn_kernel<48>, grid, dim3(64), 0, st, a); break;
    case 64: hipLaunchKernelGGL(ref_attn_kernel<64>, grid, dim3(64), 0, st, a); break;
    default: set_error("f32 reference attention: d_head = %d", a.dh); return -1;
  }
  COOT_CHECK_LAUNCH("ref_attn");
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
// TemporalAvgPool "avg_special" (poolers.py:232-241): the sum runs over ALL L rows, padded ones included; divided by the length
__global__ void ref_avgpool_kernel(const float* z, const long long* lens, int N, int L, int D, float* pooled, long ldp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * D) return;
  const int n = idx / D, c = idx % D;
  float a = 0.f;
  for (int l = 0; l < L; ++l) a += z[((long)n * L + l) * D + c];
  pooled[(long)n * ldp + c] = a / (float)lens[n];
}
__global__ void ref_copy_rows_kernel(const float* src, long lds, int R, int C, float* dst, long ldd) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * C) return;
  const long r = idx / C; const int c = (int)(idx % C);
  dst[r * ldd + c] = src[r * lds + c];
}

struct Bump32 {
  float* base; size_t cap, off = 0; bool overflow = false;
  Bump32(void* b, size_t bytes) : base((float*)b), cap(bytes / sizeof(float)) {}
  float* get(size_t n) { n = (n + 63) & ~(size_t)63; float* p = base ? base + off : nullptr; off += n; if (base && off > cap) overflow = true; return p; }
};

struct RefWs { float *xn, *z, *za, *qkv, *ctx, *r, *z1, *h, *cq, *cqa, *cqb, *hp, *s; };
void layout_ref(const RefNetDesc& d, long T, Bump32& A, RefWs& W) {
  const size_t D = d.D, F = d.F;
  W.xn = d.use_input_fc ? A.get((size_t)T * d.Din) : nullptr;
  W.z = A.get(T * D); W.za = A.get(T * D); W.qkv = A.get(T * 3 * D); W.ctx = A.get(T * D); W.r = A.get(T * D); W.z1 = A.get(T * D);
  W.h = A.get(T * (F > (size_t)d.pool_hidden ? F : (size_t)d.pool_hidden));
  W.cq = A.get((size_t)d.Nmax * D); W.cqa = A.get((size_t)d.Nmax * D); W.cqb = A.get((size_t)d.Nmax * D);
  W.hp = nullptr; W.s = A.get(T * D);
}

#define RUNR(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

int ref_ln(const float* x, long ldx, int R, int D, const float* gain, const float* bias, const float* pe, int pe_L, float* y, hipStream_t st) {
  LnFwd l; l.x = x; l.x_f32 = 1; l.ldx = ldx; l.R = R; l.D = D; l.gain = gain; l.bias = bias; l.pe = pe; l.pe_L = pe_L; l.y32 = y; l.ldy32 = D;
  return launch_ln_fwd(l, st);
}

// one TransformerEncoderLayer (transformer_legacy.py:420-438), post-LN sublayers; xq [Rq, D] queries (self: == xkv), out [Rq, D]
int ref_layer(const RefNetDesc& d, const float* P, const RefLayerP& lp, const float* xq, int Rq, const float* xkv, int Rkv, bool self,
              const RefSegs& sg, const RefWs& W, float* out, hipStream_t st) {
  const int D = d.D, F = d.F, H = d.H, dh = D / H;
  float* q; float* k; float* v; long ldq, ldkv;
  if (self) {
    RefLinear g{xq, D, P + lp.wqkv, D, 0, P + lp.bq, Rq, 3 * D, D, 0, nullptr, 0, nullptr, 0, 1, 1, W.qkv, 3L * D};
    RUNR(ref_linear(g, st));
    q = W.qkv; k = W.qkv + D; v = W.qkv + 2 * D; ldq = ldkv = 3L * D;
  } else {
    RefLinear g{xq, D, P + lp.wqkv, D, 0, P + lp.bq, Rq, D, D, 0, nullptr, 0, nullptr, 0, 1, 1, W.cqa, (long)D};
    RUNR(ref_linear(g, st));
    RefLinear g2{xkv, D, P + lp.wqkv + (size_t)D * D, D, 0, P + lp.bk, Rkv, 2 * D, D, 0, nullptr, 0, nullptr, 0, 1, 1, W.qkv, 2L * D};
    RUNR(ref_linear(g2, st));
    q = W.cqa; ldq = D; k = W.qkv; v = W.qkv + D; ldkv = 2L * D;
  }
  float* ctx = self ? W.ctx : W.cqb;
  long qrow = 0, krow = 0;
  for (int s = 0; s < sg.n; ++s) {
    RefAttn a{q, k, v, ldq, ldkv, ldkv, ctx, (long)D, sg.lens[s], sg.N[s], self ? sg.L[s] : 1, sg.L[s], H, dh, qrow, krow, 1.0f / sqrtf((float)dh)};
    RUNR(ref_attn(a, st));
    qrow += (long)sg.N[s] * a.Lq; krow += (long)sg.N[s] * sg.L[s];
  }
  // the temporaries of the rest of the layer: token-sized for self-attention layers, sequence-sized for the context block (where the
  // projected queries in cqa are dead once the attention ran, and the attention output in cqb once the out-projection read it)
  float* r = self ? W.r : W.cqa;
  float* z1 = self ? W.z1 : W.cqb;
  {
    RefLinear g{ctx, D, P + lp.wo, D, 0, P + lp.bo, Rq, D, D, 0, xq, (long)D, nullptr, 0, 1, 1, r, (long)D};
    RUNR(ref_linear(g, st));
  }
  RUNR(ref_ln(r, D, Rq, D, P + lp.ln1g, P + lp.ln1b, nullptr, 1, z1, st));
  {
    RefLinear g{z1, D, P + lp.w1, D, 0, P + lp.b1, Rq, F, D, 1, nullptr, 0, nullptr, 0, 1, 1, W.h, (long)F};
    RUNR(ref_linear(g, st));
  }
  {
    RefLinear g{W.h, F, P + lp.w2, F, 0, P + lp.b2, Rq, D, F, 0, z1, (long)D, nullptr, 0, 1, 1, r, (long)D};
    RUNR(ref_linear(g, st));
  }
  return ref_ln(r, D, Rq, D, P + lp.ln2g, P + lp.ln2b, nullptr, 1, out, st);
}

}  // namespace

size_t ref_f32_workspace_bytes(const RefNetDesc& d, long T) {
  Bump32 A(nullptr, 0); RefWs W; layout_ref(d, T, A, W);
  return A.off * sizeof(float) + 1024;
}

int ref_f32_forward(const RefNetDesc& d, const float* P, const float* pe, const float* feats, const float* feats2, const RefSegs& sg,
                    const float* hidden, float* pooled, float* per_token, void* ws, size_t ws_bytes, hipStream_t st) {
  const int D = d.D, Din = d.Din;
  long T = 0; int Ntot = 0;
  for (int s = 0; s < sg.n; ++s) { T += (long)sg.N[s] * sg.L[s]; Ntot += sg.N[s]; }
  Bump32 A(ws, ws_bytes); RefWs W; layout_ref(d, T, A, W);
  COOT_REQUIRE(!A.overflow && Ntot <= d.Nmax, "net_fwd (f32 reference mode): workspace too small (%zu bytes)", ws_bytes);
  const long T0 = (long)sg.N[0] * sg.L[0];
  const int out_dim = D * (d.use_context ? 2 : 1);
  // ---- input: LayerNorm (own gain / bias: nothing is folded here) [+ Linear + GELU] + positional encoding (transformer_legacy.py:222-241)
  if (d.use_input_fc) {
    RUNR(ref_ln(feats, Din, (int)T0, Din, P + d.n_gain, P + d.n_bias, nullptr, 1, W.xn, st));
    if (sg.n > 1) RUNR(ref_ln(feats2, Din, (int)(T - T0), Din, P + d.n_gain, P + d.n_bias, nullptr, 1, W.xn + T0 * Din, st));
    RefLinear g{W.xn, Din, P + d.in_w, Din, 0, P + d.in_b, (int)T, D, Din, 1, nullptr, 0, pe, (int)T0, sg.L[0], sg.n > 1 ? sg.L[1] : sg.L[0], W.z, (long)D};
    RUNR(ref_linear(g, st));
  } else {
    RUNR(ref_ln(feats, Din, (int)T0, Din, P + d.n_gain, P + d.n_bias, pe, sg.L[0], W.z, st));
    if (sg.n > 1) RUNR(ref_ln(feats2, Din, (int)(T - T0), Din, P + d.n_gain, P + d.n_bias, pe, sg.L[1], W.z + T0 * D, st));
  }
  float* z = W.z; float* zo = W.za;
  for (int i = 0; i < d.num_layers; ++i) {
    RUNR(ref_layer(d, P, d.layers[i], z, (int)T, z, (int)T, true, sg, W, zo, st));
    float* t = z; z = zo; zo = t;
  }
  if (per_token) {
    hipLaunchKernelGGL(ref_copy_rows_kernel, dim3((unsigned)((T0 * D + 255) / 256)), dim3(256), 0, st, (const float*)z, (long)D, (int)T0, D, per_token, (long)D);
    COOT_CHECK_LAUNCH("ref_copy");
  }
  // ---- context block: one query per sequence, keys / values = the encoder output (:251-267)
  if (d.use_context) {
    COOT_REQUIRE(sg.n == 1 && hidden, "net_fwd (f32 reference mode): context networks take one segment and a hidden state");
    const float* cq = hidden;
    for (int i = 0; i < d.ctx_num_layers; ++i) {
      RUNR(ref_layer(d, P, d.ctx[i], cq, Ntot, z, (int)T, false, sg, W, W.cq, st));
      cq = W.cq;
    }
    hipLaunchKernelGGL(ref_copy_rows_kernel, dim3((unsigned)(((long)Ntot * D + 255) / 256)), dim3(256), 0, st, cq, (long)D, Ntot, D, pooled + D, (long)out_dim);
    COOT_CHECK_LAUNCH("ref_copy");
  }
  // ---- pooling
  if (d.pooler == 0) {
    const int Hp = d.pool_heads, PH = d.pool_hidden, dhp = PH / Hp, dop = D / Hp;
    for (int h = 0; h < Hp; ++h) {
      // a_h = GELU(z W1[h] + b1[h])  (W1[h]: [D, dhp]);  s_h = a_h W2[h] + b2[h]  (W2[h]: [dhp, dop]) -> channels [h dop, (h + 1) dop)
      RefLinear g{z, D, P + d.pw1 + (size_t)h * D * dhp, dhp, 1, P + d.pb1 + (size_t)h * dhp, (int)T, dhp, D, 1, nullptr, 0, nullptr, 0, 1, 1, W.h, (long)dhp};
      RUNR(ref_linear(g, st));
      RefLinear g2{W.h, dhp, P + d.pw2 + (size_t)h * dhp * dop, dop, 1, P + d.pb2 + (size_t)h * dop, (int)T, dop, dhp, 0, nullptr, 0, nullptr, 0, 1, 1, W.s + h * dop, (long)D};
      RUNR(ref_linear(g2, st));
    }
    long row = 0; int n0 = 0;
    for (int s = 0; s < sg.n; ++s) {
      hipLaunchKernelGGL(ref_genpool_kernel, dim3((sg.N[s] * D + 255) / 256), dim3(256), 0, st, (const float*)W.s, (const float*)z, sg.lens[s], sg.N[s], sg.L[s], D,
                         row, pooled + (size_t)n0 * out_dim, (long)out_dim);
      COOT_CHECK_LAUNCH("ref_genpool");
      row += (long)sg.N[s] * sg.L[s]; n0 += sg.N[s];
    }
  } else {
    hipLaunchKernelGGL(ref_avgpool_kernel, dim3((sg.N[0] * D + 255) / 256), dim3(256), 0, st, (const float*)z, sg.lens[0], sg.N[0], sg.L[0], D, pooled, (long)out_dim);
    COOT_CHECK_LAUNCH("ref_avgpool");
  }
  return 0;
}

}  // namespace coot
