// compute_total_constrastive_loss (coot/trainer_retrieval.py:148-182) in THREE launches.
//
// The step's loss section used to be ~80 launches of tiny kernels and memsets on the critical path (0.75 ms,
// profiles/r01b).  Restated so that every quantity is produced exactly once, by the workgroup that owns it:
//   1. cl_norm_kernel   one wave per row i of a PAIR of sets (vid|par, clip|sent, vid_ctx|par_ctx): F.normalize of both
//                       rows (bf16 a [Np, d] and transposed a^T [d, Np], zero padded), inv-norms, and the three diagonals
//                       <a_i,b_i>, <a_i,a_i>, <b_i,b_i> the hinge terms compare against.
//   2. cl_half_kernel   one workgroup per (half-term, 16 rows).  ContrastiveLoss (coot/loss_fn.py:63-100) has the
//                       symmetric form  G_ij = [m + S_ij - S_ii > 0] + [m + S_ij - S_jj > 0]  (i != j), so the gradient
//                       wrt B of L(A, B) is the "A-side" gradient of the swapped problem (S' = B A^T): every term is
//                       two half-terms, each computing a 16 x N strip of S on the MFMA (operands straight from L2),
//                       the hinge + violation counts in registers, G (exact small integers in bf16) in LDS, and
//                       dX_strip = G_strip . Y on the MFMA again.  No atomics: per-strip loss partials, per-row
//                       violation counts and dX strips go to per-half-term buffers.
//   3. cl_finish_kernel one wave per row of each set: sums the half-term strips + the diagonal term
//                       (gd_i = -(c1_i(A,B) + c1_i(B,A))), F.normalize backward, accumulates into d_emb; block 0 adds the
//                       loss partials in a fixed order (deterministic).
#include "loss.h"

namespace coot {

struct ClPair { const float* va; const float* vb; long lda, ldb; bf16_t *a, *b, *aT, *bT; float *inva, *invb; float *dab, *daa, *dbb; int N, Np, d; };
struct ClNormArgs { ClPair p[3]; int row0[4]; ClBlocks blk; };  // blk.world == 0: dense / strided sets

constexpr int CL_TP = 8;  // pitch padding of the transposition tile (elements)
__global__ __launch_bounds__(1024) void cl_norm_kernel(ClNormArgs A) {
  // one workgroup = 16 rows of one pair (the pairs are padded to multiples of 16 rows), one wave per row.  The transposed
  // copies a^T / b^T go through an LDS tile so that a thread stores the 16 rows of one feature as 32 contiguous bytes (every
  // wave scattering its own row's 2-byte elements at a stride of Np took 24 us for the 5 600 gathered rows of an 8-rank job).
  extern __shared__ __attribute__((aligned(16))) bf16_t Ts[];  // [2][16][d + CL_TP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grow = blockIdx.x * 16 + wave;
  const int pi = grow >= A.row0[2] ? 2 : (grow >= A.row0[1] ? 1 : 0);
  const ClPair& P = A.p[pi];
  const int row = grow - A.row0[pi], d = P.d, tp = d + CL_TP;
  bf16_t* Ta = Ts + wave * tp;
  bf16_t* Tb = Ts + (16 + wave) * tp;
  if (row >= P.N) {  // zero padding rows / columns (the MFMA strips read them)
    for (int c = lane; c < d; c += 64) {
      P.a[(long)row * d + c] = 0; P.b[(long)row * d + c] = 0;
      Ta[c] = 0; Tb[c] = 0;
    }
  } else {
    // the two rows in registers (d % 32 == 0, d <= 1024: up to four 4-element chunks per lane): one pass of 16-byte loads
    constexpr int MAXC = 4;
    f32x4_t xa[MAXC], xb[MAXC];
    const int nch = d / 4;
    float sa = 0.f, sb = 0.f;
    const float* ra = P.va + (long)row * P.lda;
    const float* rb = P.vb + (long)row * P.ldb;
    if (A.blk.world > 0) {  // rows of an all-gather: find the rank's block (row is wave-uniform: scalar code)
      const int lvl = pi == 1 ? 1 : 0;
      int r = 0;
      for (int t = 1; t < A.blk.world; ++t) if (row >= A.blk.row0[lvl][t]) r = t;
      const long i = row - A.blk.row0[lvl][r];
      ra = A.blk.blocks + A.blk.base[2 * pi][r] + i * P.lda;
      rb = A.blk.blocks + A.blk.base[2 * pi + 1][r] + i * P.ldb;
    }
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
      const int ch = lane + 64 * q;
      xa[q] = f32x4_t{0.f, 0.f, 0.f, 0.f}; xb[q] = xa[q];
      if (ch < nch) {
        xa[q] = *reinterpret_cast<const f32x4_t*>(ra + ch * 4);
        xb[q] = *reinterpret_cast<const f32x4_t*>(rb + ch * 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { sa = fmaf(xa[q][j], xa[q][j], sa); sb = fmaf(xb[q][j], xb[q][j], sb); }  // (explicit fma: see cl_small_kernel)
    }
    sa = wave_sum(sa); sb = wave_sum(sb);
    const float ia = 1.0f / fmaxf(sqrtf(sa), 1e-12f), ib = 1.0f / fmaxf(sqrtf(sb), 1e-12f);
    float dab = 0.f, daa = 0.f, dbb = 0.f;
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
      const int ch = lane + 64 * q;
      if (ch < nch) {
        bf16_t ha[4], hb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ha[j] = f2bf(xa[q][j] * ia); hb[j] = f2bf(xb[q][j] * ib);
          const float fa = bf2f(ha[j]), fb = bf2f(hb[j]);
          dab = fmaf(fa, fb, dab); daa = fmaf(fa, fa, daa); dbb = fmaf(fb, fb, dbb);
        }
        const u32x2_t pa = {(unsigned)ha[0] | ((unsigned)ha[1] << 16), (unsigned)ha[2] | ((unsigned)ha[3] << 16)};
        const u32x2_t pb = {(unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[2] | ((unsigned)hb[3] << 16)};
        *reinterpret_cast<u32x2_t*>(P.a + (long)row * d + ch * 4) = pa;
        *reinterpret_cast<u32x2_t*>(P.b + (long)row * d + ch * 4) = pb;
        *reinterpret_cast<u32x2_t*>(Ta + ch * 4) = pa;
        *reinterpret_cast<u32x2_t*>(Tb + ch * 4) = pb;
      }
    }
    dab = wave_sum(dab); daa = wave_sum(daa); dbb = wave_sum(dbb);
    if (lane == 0) { P.inva[row] = ia; P.invb[row] = ib; P.dab[row] = dab; P.daa[row] = daa; P.dbb[row] = dbb; }
  }
  __syncthreads();
  const int r0 = row - wave;  // first row of this workgroup within the pair
  for (int c = tid; c < 2 * d; c += 1024) {
    const int second = c >= d, cc = c - second * d;
    const bf16_t* T = Ts + second * 16 * tp + cc;
    unsigned w[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r] = (unsigned)T[(2 * r) * tp] | ((unsigned)T[(2 * r + 1) * tp] << 16);
    bf16_t* o = (second ? P.bT : P.aT) + (long)cc * P.Np + r0;
    *reinterpret_cast<u32x4_t*>(o) = u32x4_t{w[0], w[1], w[2], w[3]};
    *reinterpret_cast<u32x4_t*>(o + 8) = u32x4_t{w[4], w[5], w[6], w[7]};
  }
}

constexpr int CL_MAX_HALF = 12;
struct ClHalf {
  const bf16_t* X; const bf16_t* Y; const bf16_t* YT;  // X [Np, d], Y [Np, d], Y^T [d, Np]
  const float* diag;                                   // [N]  S_ii of this term
  float* dX;                                           // [N, d]   sum_{j != i} G_ij y_j   (unscaled)
  float* c1;                                           // [N]      #{j : m + S_ij - S_ii > 0}
  float* loss_part;                                    // [nblk]   sum of hinge values of the strip (only if primary)
  int N, Np, d, blk0, primary;
  int w0, wn;                                          // rows whose dX strip is needed (data parallel: this rank's rows)
  int cs, cols;                                        // column splits of a strip and columns per split (a multiple of 32): dX, c1 and
};                                                     // loss_part are per-split partials ([cs][Np, d], [cs][Np], [strip][cs]) cl_finish sums
struct ClHalfArgs { ClHalf h[CL_MAX_HALF]; int nh; int nblk; float margin; };

constexpr int CL_GP = 8;  // G strip pitch padding (elements)

constexpr int CL_NW = 8;  // waves per strip workgroup: the strip is a chain of L2 round trips, more waves = fewer trips each
__global__ __launch_bounds__(64 * CL_NW) void cl_half_kernel(ClHalfArgs A) {
  extern __shared__ __attribute__((aligned(16))) bf16_t Gs[];  // [16][cols + CL_GP]: the G strip over this workgroup's columns
  __shared__ int c1s[16];  // violation counts: integer LDS atomics (ds_add_u32), not float ones
  __shared__ float lred[CL_NW];
  int hi = 0;
  for (int t = 1; t < A.nh; ++t) if ((int)blockIdx.x >= A.h[t].blk0) hi = t;
  const ClHalf& H = A.h[hi];
  // a strip of 16 rows x Np columns is a chain of dependent L2 round trips (S column blocks, then G . Y over k): with the gathered
  // batch of a data-parallel job (Np = 2048 clips at 8 ranks, own rows = 16 strips per half-term) a strip per workgroup left the
  // chip at ~100 workgroups of 50 round trips each.  The columns are therefore split over cs workgroups (adjacent block indices:
  // they share the strip's X rows in L2); each writes partial dX / c1 / loss sums, added in a fixed order by cl_finish.
  const int rb = (blockIdx.x - H.blk0) / H.cs, sp = (blockIdx.x - H.blk0) % H.cs, i0 = rb * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = H.N, Np = H.Np, d = H.d, gp = H.cols + CL_GP;
  const int c0 = sp * H.cols, c1e = min(Np, c0 + H.cols);  // this workgroup's columns [c0, c1e)
  // data parallel: a rank scores ITS rows [w0, w0 + wn) against the gathered batch — strips without own rows are nobody's business
  // here (their hinge sums belong to the loss partial of the rank that owns them, which the caller adds up across ranks)
  if (i0 + 16 <= H.w0 || i0 >= H.w0 + H.wn) return;
  // the strip's own rows (MFMA A operand of every column block) live in LDS: holding them as register fragments (up to 128 VGPRs)
  // left one workgroup per CU — with the strip split over columns the launch has hundreds of short workgroups that should overlap
  bf16_t* Xs = Gs + 16 * gp;  // [16][d + CL_GP]
  const int xp = d + CL_GP;
  for (int c = tid; c < 16 * (d / 8); c += 64 * CL_NW) {
    const int r = c / (d / 8), cc = c % (d / 8);
    *reinterpret_cast<bf16x8_t*>(&Xs[r * xp + cc * 8]) = *reinterpret_cast<const bf16x8_t*>(H.X + (long)(i0 + r) * d + cc * 8);
  }
  if (tid < 16) c1s[tid] = 0;
  __syncthreads();
  // ---- S strip: wave w takes column blocks w, w + CL_NW, ...; X (A operand, m = row i), Y (B operand, n = column j) ----
  const int kbs = d / 32;
  const int l15 = lane & 15, l4 = lane >> 4;
  const bf16_t* xfrag = Xs + l15 * xp + l4 * 8;
  float di[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int i = i0 + l4 * 4 + r; di[r] = i < N ? H.diag[i] : 0.f; }
  float lsum = 0.f;
  int c1r[4] = {0, 0, 0, 0};
  for (int cb = c0 / 16 + wave; cb * 16 < c1e; cb += CL_NW) {
    const bf16_t* yrow = H.Y + (long)(cb * 16 + l15) * d + l4 * 8;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    // twelve k-blocks of Y per round trip (d = 384: the whole row): the loads of a group first, then its MFMAs.  (One predicated
    // load + MFMA per k-block compiled to load, s_waitcnt vmcnt(0), mfma — 12 to 24 serial L2 round trips per column block.)
    for (int k0 = 0; k0 < kbs; k0 += 12) {
      bf16x8_t yf[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int kb = min(k0 + j, kbs - 1);  // clamp instead of branching: every load of the group issues unconditionally
        yf[j] = *reinterpret_cast<const bf16x8_t*>(yrow + kb * 32);
      }
#pragma unroll
      for (int j = 0; j < 12; ++j)
        if (k0 + j < kbs) acc = COOT_MFMA_16x16x32(*reinterpret_cast<const bf16x8_t*>(xfrag + (k0 + j) * 32), yf[j], acc);
    }
    // acc[r] = S[i0 + l4*4 + r][cb*16 + l15]
    const int j = cb * 16 + l15;
    const float dj = j < N ? H.diag[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + l4 * 4 + r;
      float g = 0.f;
      if (i < N && j < N && i != j) {
        const float cs = A.margin + acc[r] - di[r];
        const float ci = A.margin + acc[r] - dj;
        const bool own = i >= H.w0 && i < H.w0 + H.wn;  // (a strip at the edge of the window holds rows of the neighbouring rank)
        if (cs > 0.f) { if (own) lsum += cs; g += 1.f; c1r[r] += 1; }
        if (ci > 0.f) { if (own) lsum += ci; g += 1.f; }
      }
      Gs[(l4 * 4 + r) * gp + j - c0] = f2bf(g);
    }
  }
  // zero the K padding of the strip (columns c1e .. the next multiple of 32)
  const int kw = (c1e - c0 + 31) & ~31;
  for (int c = c1e - c0 + tid; c < kw; c += 64 * CL_NW)
#pragma unroll
    for (int r = 0; r < 16; ++r) Gs[r * gp + c] = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int v = c1r[r];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    if (l15 == 0 && v != 0) atomicAdd(&c1s[l4 * 4 + r], v);
  }
  lsum = wave_sum(lsum);
  if (lane == 0) lred[wave] = lsum;
  __syncthreads();
  if (tid < 16 && i0 + tid < N) H.c1[(long)sp * Np + i0 + tid] = (float)c1s[tid];
  if (tid == 0 && H.primary) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < CL_NW; ++w) t += lred[w];
    H.loss_part[rb * H.cs + sp] = t;
  }
  // ---- dX strip [16, d] = G strip [16, cols] . Y [cols, d]: A operand = G (LDS), B operand = Y^T rows (k contiguous) ----
  const int nf = d / 16;
  float* dXs = H.dX + (long)sp * Np * d;
  for (int f0 = wave; f0 < nf; f0 += CL_NW * 3) {
    f32x4_t acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // four k-blocks x three feature fragments of Y^T per round trip (loads first, clamped addresses instead of branches:
    // the padded columns of G are zero, so what a clamped load brings in does not matter)
    const int nkb = kw / 32;
    for (int k0 = 0; k0 < nkb; k0 += 4) {
      bf16x8_t yf[4][3], gf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = min(k0 + j, nkb - 1) * 32 + l4 * 8;
        gf[j] = *reinterpret_cast<const bf16x8_t*>(&Gs[l15 * gp + kk]);
        const int kc = min(c0 + kk, Np - 8);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int f = min(f0 + CL_NW * q, nf - 1);
          yf[j][q] = *reinterpret_cast<const bf16x8_t*>(H.YT + (long)(f * 16 + l15) * Np + kc);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k0 + j < nkb) {
#pragma unroll
          for (int q = 0; q < 3; ++q) acc[q] = COOT_MFMA_16x16x32(gf[j], yf[j][q], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int f = f0 + CL_NW * q;
      if (f < nf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + l4 * 4 + r;
          if (i < N) dXs[(long)i * d + f * 16 + l15] = acc[q][r];
        }
      }
    }
  }
}

// ---- small sets: cl_norm + cl_half in ONE launch ---------------------------------------------------------------------------------
// The (vid, par) terms of the train step (64 rows of 768 features) sit between the text side's last forward launch and the video
// side's backward: three dependent launches of 4 / 16 / 32 workgroups, each a chain of L2 round trips.  When a whole set fits the
// LDS, a half-term's strip workgroup normalises what it needs itself — all rows of Y, its 16 rows of X — straight from the fp32
// embeddings, and both MFMA products read their operands from LDS: no cl_norm launch, no operand round trips.  Same arithmetic in
// the same order as cl_norm / cl_half (one wave per row with the same lane -> element map, k-blocks accumulated in order, the same
// hinge code; the sums of products are explicit fmaf in both — under -ffp-contract=fast the compiler fuses a * b + c or not as it
// likes, differently in two kernels), so the outputs cl_finish reads are bit-identical to the three-launch path's
// (tests/test_gpu_path.py::test_contrastive_small_sets_one_launch_equals_three).
struct ClSmall {
  const float* vx; const float* vy; long ldx, ldy;  // raw rows of the X / Y set
  bf16_t* xn; float* invx;                          // normalised bf16 rows + 1 / norm of X for cl_finish (strip rows), or null
  float* dX; float* c1; float* loss_part;           // as ClHalf (one column split)
  int N, Np, d, blk0, primary, same;                // same: X and Y are one set (cluster terms)
};
struct ClSmallArgs { ClSmall h[CL_MAX_HALF]; int nh; float margin; };
constexpr int CLS_NW = 16;

// one row (chunks lane, lane + 64, ... of 4 elements per lane): load — unconditional, a chunk past the row re-reads its last one, so
// that the loads of several rows issue back to back
template <int M>
__device__ __forceinline__ void cls_load_row(const float* r, int nch, int lane, f32x4_t (&x)[M]) {
#pragma unroll
  for (int q = 0; q < M; ++q) x[q] = *reinterpret_cast<const f32x4_t*>(r + min(lane + 64 * q, nch - 1) * 4);
}
template <int M>
__device__ __forceinline__ float cls_sumsq(int nch, int lane, f32x4_t (&x)[M]) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < M; ++q) {
    if (lane + 64 * q >= nch) x[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) s = fmaf(x[q][j], x[q][j], s);
  }
  return s;
}
// x <- bf16(x / norm) (F.normalize) as floats and as packed words (two elements per v_cvt_pk_bf16_f32; a padding row: zeros);
// returns 1 / norm.  The rows phase is VALU bound (16 waves normalise every row of both sets): ~5 instructions per element.
template <int M>
__device__ __forceinline__ float cls_scale_round(float sumsq, bool valid, f32x4_t (&x)[M], u32x2_t (&w)[M]) {
  const float inv = 1.0f / fmaxf(sqrtf(sumsq), 1e-12f);
  const float sc = valid ? inv : 0.f;
#pragma unroll
  for (int q = 0; q < M; ++q) {
    w[q] = u32x2_t{pack2bf(x[q][0] * sc, x[q][1] * sc), pack2bf(x[q][2] * sc, x[q][3] * sc)};
    x[q] = f32x4_t{bflo(w[q][0]), bfhi(w[q][0]), bflo(w[q][1]), bfhi(w[q][1])};
  }
  return inv;
}
// wave_sum of K values at once: the K butterflies are independent, their cross-lane steps issue back to back (one after the other
// they were most of this kernel: 18 dependent LDS-pipe round trips per row)
template <int K>
__device__ __forceinline__ void wave_sum_n(float (&v)[K]) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float t[K];
#pragma unroll
    for (int k = 0; k < K; ++k) t[k] = __shfl_xor(v[k], o, 64);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += t[k];
  }
}

// M: 256-element chunks of a row (d <= 256 M): registers for the rows in flight, not for the widest row the launch allows
#ifdef CLS_DBG_STAMPS  // (tools/cl_small_probe.py with a -DCLS_DBG_STAMPS build: phase clocks of workgroup 0, wave 0)
#define CLS_T(i) do { if (tid == 0 && blockIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tst[i] = __builtin_readcyclecounter(); } } while (0)
#else
#define CLS_T(i) do { } while (0)
#endif
template <int M>
__global__ __launch_bounds__(64 * CLS_NW) void cl_small_kernel(ClSmallArgs A) {
  extern __shared__ __attribute__((aligned(16))) bf16_t Ls[];  // Ys [Np][d + 8] | Xs [16][d + 8] | Gs [16][kw + 8] | diag [Np] (fp32)
  __shared__ int c1s[16];
  __shared__ float lred[CLS_NW];
  int hi = 0;
  for (int t = 1; t < A.nh; ++t) if ((int)blockIdx.x >= A.h[t].blk0) hi = t;
  const ClSmall& H = A.h[hi];
  const int rb = blockIdx.x - H.blk0, i0 = rb * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CLS_DBG_STAMPS
  unsigned long long tst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  CLS_T(0);
  const int N = H.N, Np = H.Np, d = H.d, yp = d + CL_GP, kw = (Np + 31) & ~31, gp = kw + CL_GP;
  bf16_t* Ys = Ls;
  bf16_t* Xs = Ys + (long)Np * yp;
  bf16_t* Gs = Xs + 16 * yp;
  float* diag = reinterpret_cast<float*>(Gs + 16 * gp);
  const int nch = d / 4;
  // ---- rows: wave w normalises rows w, w + 16, ... of Y (and of X where the diagonal or the strip needs them): two rows per
  // round, all their loads in flight at once and their reductions side by side (a round = one memory latency + two butterflies).
  // Padding rows (j >= N) load a valid row and turn into zeros. ----
  auto store_row = [&](int j, float dg, float ix, const u32x2_t (&wy)[M], const u32x2_t (&wx)[M]) {
    const bool mine = j >= i0 && j < i0 + 16;
#pragma unroll
    for (int q = 0; q < M; ++q) {
      const int ch = lane + 64 * q;
      if (ch < nch) {
        *reinterpret_cast<u32x2_t*>(Ys + (long)j * yp + ch * 4) = wy[q];
        if (mine) {
          *reinterpret_cast<u32x2_t*>(Xs + (j - i0) * yp + ch * 4) = wx[q];
          if (H.xn && j < N) *reinterpret_cast<u32x2_t*>(H.xn + (long)j * d + ch * 4) = wx[q];
        }
      }
    }
    if (lane == 0) { diag[j] = dg; if (mine && H.invx && j < N) H.invx[j] = ix; }
  };
  auto dot_rows = [&](const f32x4_t (&x)[M], const f32x4_t (&y)[M]) {
    float dg = 0.f;
#pragma unroll
    for (int q = 0; q < M; ++q)
      if (lane + 64 * q < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dg = fmaf(x[q][e], y[q][e], dg);
      }
    return dg;
  };
  for (int j0 = wave; j0 < Np; j0 += 2 * CLS_NW) {
    const int j1 = j0 + CLS_NW, r0 = min(j0, N - 1), r1 = min(j1, N - 1);
    if constexpr (M == 4) {  // 1024-wide rows: four of them in flight do not fit 128 VGPRs — one (y, x) pair at a time
      if (!H.same) {
        auto one = [&](int j, int r) {
          f32x4_t y[M], x[M];
          cls_load_row(H.vy + (long)r * H.ldy, nch, lane, y);
          cls_load_row(H.vx + (long)r * H.ldx, nch, lane, x);
          float ss[2] = {cls_sumsq(nch, lane, y), cls_sumsq(nch, lane, x)};
          wave_sum_n<2>(ss);
          u32x2_t wy[M], wx[M];
          cls_scale_round(ss[0], j < N, y, wy);
          const float ix = cls_scale_round(ss[1], j < N, x, wx);
          float dg1[1] = {dot_rows(x, y)};
          wave_sum_n<1>(dg1);
          store_row(j, dg1[0], ix, wy, wx);
        };
        one(j0, r0);
        if (j1 < Np) one(j1, r1);
        continue;
      }
    }
    f32x4_t y0[M], y1[M];
    cls_load_row(H.vy + (long)r0 * H.ldy, nch, lane, y0);
    cls_load_row(H.vy + (long)r1 * H.ldy, nch, lane, y1);
    float dg[2];
    if (!H.same) {
      f32x4_t x0[M], x1[M];
      cls_load_row(H.vx + (long)r0 * H.ldx, nch, lane, x0);
      cls_load_row(H.vx + (long)r1 * H.ldx, nch, lane, x1);
      if (j0 == wave) CLS_T(1);
      float ss[4] = {cls_sumsq(nch, lane, y0), cls_sumsq(nch, lane, y1), cls_sumsq(nch, lane, x0), cls_sumsq(nch, lane, x1)};
      wave_sum_n<4>(ss);
      if (j0 == wave) CLS_T(2);
      u32x2_t wy0[M], wy1[M], wx0[M], wx1[M];
      cls_scale_round(ss[0], j0 < N, y0, wy0); cls_scale_round(ss[1], j1 < N, y1, wy1);
      const float ix0 = cls_scale_round(ss[2], j0 < N, x0, wx0), ix1 = cls_scale_round(ss[3], j1 < N, x1, wx1);
      dg[0] = dot_rows(x0, y0); dg[1] = dot_rows(x1, y1);
      wave_sum_n<2>(dg);
      if (j0 == wave) CLS_T(3);
      store_row(j0, dg[0], ix0, wy0, wx0);
      if (j1 < Np) store_row(j1, dg[1], ix1, wy1, wx1);
      if (j0 == wave) CLS_T(4);
    } else {
      float ss[2] = {cls_sumsq(nch, lane, y0), cls_sumsq(nch, lane, y1)};
      wave_sum_n<2>(ss);
      u32x2_t wy0[M], wy1[M];
      const float ix0 = cls_scale_round(ss[0], j0 < N, y0, wy0), ix1 = cls_scale_round(ss[1], j1 < N, y1, wy1);
      dg[0] = dot_rows(y0, y0); dg[1] = dot_rows(y1, y1);
      wave_sum_n<2>(dg);
      store_row(j0, dg[0], ix0, wy0, wy0);
      if (j1 < Np) store_row(j1, dg[1], ix1, wy1, wy1);
    }
  }
  CLS_T(5);
  if (tid < 16) c1s[tid] = 0;
  __syncthreads();
  CLS_T(6);
  // ---- S strip (16 x Np): wave w takes column blocks w, w + 16, ... ----
  const int kbs = d / 32;
  const int l15 = lane & 15, l4 = lane >> 4;
  const bf16_t* xfrag = Xs + l15 * yp + l4 * 8;
  float di[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int i = i0 + l4 * 4 + r; di[r] = i < N ? diag[i] : 0.f; }
  float lsum = 0.f;
  int c1r[4] = {0, 0, 0, 0};
  for (int cb = wave; cb * 16 < Np; cb += CLS_NW) {
    const bf16_t* yfrag = Ys + (long)(cb * 16 + l15) * yp + l4 * 8;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 6  // (d % 32 == 0; LDS reads of the next k-blocks under the dependent MFMA chain)
    for (int kb = 0; kb < kbs; ++kb)
      acc = COOT_MFMA_16x16x32(*reinterpret_cast<const bf16x8_t*>(xfrag + kb * 32),
                                                    *reinterpret_cast<const bf16x8_t*>(yfrag + kb * 32), acc);
    const int j = cb * 16 + l15;
    const float dj = j < N ? diag[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + l4 * 4 + r;
      float g = 0.f;
      if (i < N && j < N && i != j) {
        const float cs = A.margin + acc[r] - di[r];
        const float ci = A.margin + acc[r] - dj;
        if (cs > 0.f) { lsum += cs; g += 1.f; c1r[r] += 1; }
        if (ci > 0.f) { lsum += ci; g += 1.f; }
      }
      Gs[(l4 * 4 + r) * gp + j] = f2bf(g);
    }
  }
  for (int c = Np + tid; c < kw; c += 64 * CLS_NW)  // K padding of the G strip
#pragma unroll
    for (int r = 0; r < 16; ++r) Gs[r * gp + c] = 0;
  // the four counts (integers: any order) and the hinge sum (the butterfly of wave_sum) side by side: independent cross-lane chains
  lsum += __shfl_xor(lsum, 32, 64);
  lsum += __shfl_xor(lsum, 16, 64);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    int t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = __shfl_xor(c1r[r], o, 64);
    const float tl = __shfl_xor(lsum, o, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) c1r[r] += t[r];
    lsum += tl;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (l15 == 0 && c1r[r] != 0) atomicAdd(&c1s[l4 * 4 + r], c1r[r]);
  if (lane == 0) lred[wave] = lsum;
  CLS_T(7);
  __syncthreads();
  CLS_T(8);
  if (tid < 16 && i0 + tid < N) H.c1[i0 + tid] = (float)c1s[tid];
  if (tid == 0 && H.primary) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < CLS_NW; ++w) t += lred[w];
    H.loss_part[rb] = t;
  }
  // ---- dX strip [16, d] = G strip [16, kw] . Y [kw, d]: A operand = G, B operand (k = column j, n = feature) = a column of the
  // row-major Y tile: eight 2-byte LDS reads per fragment (the set is small: 6 fragments per wave at 64 x 768) ----
  const int nf = d / 16, nkb = kw / 32;
  const unsigned short* Yu = reinterpret_cast<const unsigned short*>(Ys);
  for (int f = wave; f < nf; f += CLS_NW) {
    // all operand reads of the fragment first (up to 4 k-blocks: Np <= 128; a k-block past the strip re-reads the last one)
    bf16x8_t gf[4];
    unsigned short e[4][8];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int kc = min(kb, nkb - 1);
      gf[kb] = *reinterpret_cast<const bf16x8_t*>(&Gs[l15 * gp + kc * 32 + l4 * 8]);
#pragma unroll
      for (int t = 0; t < 8; ++t) e[kb][t] = Yu[(long)min(kc * 32 + l4 * 8 + t, Np - 1) * yp + f * 16 + l15];  // (G is zero past Np)
    }
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      if (kb < nkb) {
        const u32x4_t w = {(unsigned)e[kb][0] | ((unsigned)e[kb][1] << 16), (unsigned)e[kb][2] | ((unsigned)e[kb][3] << 16),
                           (unsigned)e[kb][4] | ((unsigned)e[kb][5] << 16), (unsigned)e[kb][6] | ((unsigned)e[kb][7] << 16)};
        acc = COOT_MFMA_16x16x32(gf[kb], *reinterpret_cast<const bf16x8_t*>(&w), acc);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + l4 * 4 + r;
      if (i < N) H.dX[(long)i * d + f * 16 + l15] = acc[r];
    }
  }
  CLS_T(9);
#ifdef CLS_DBG_STAMPS
  if (tid == 0 && blockIdx.x == 0)
    printf("cl_small wg0 wave0 clocks: loads %llu, sums %llu, scale+dot %llu, stores %llu, round 2.. %llu, barrier %llu, S strip %llu, barrier %llu, dX %llu; total %llu\n",
           tst[1] - tst[0], tst[2] - tst[1], tst[3] - tst[2], tst[4] - tst[3], tst[5] - tst[4], tst[6] - tst[5], tst[7] - tst[6], tst[8] - tst[7],
           tst[9] - tst[8], tst[9] - tst[0]);
#endif
}

// per set: up to 3 contributions  coef * dX_h[i] + dcoef * (c1_p[i] + c1_q[i]) * other[i]   (other = bf16 normalised rows)
struct ClContrib { const float* dX; float coef; const float* c1p; const float* c1q; float dcoef; const bf16_t* other; int cs; long dxs, c1s; };  // cs partials, strides dxs / c1s
struct ClSet { const float* v; long ldv; const float* inv; float* dv; int N, d, nc, row0, w0, wn; ClContrib c[3]; };  // dv: rows [w0, w0 + wn) only, compact
struct ClFinishArgs { ClSet s[6]; int rows; const float* loss_part[CL_MAX_HALF]; int loss_n[CL_MAX_HALF]; float loss_coef[CL_MAX_HALF]; int nl; float* loss; };

__global__ __launch_bounds__(256) void cl_finish_kernel(ClFinishArgs A) {
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == gridDim.x - 1) {
    // the launch's extra workgroup: the loss.  Lane l takes partials l, l + 64, ... of the concatenated strip sums (their loads in
    // flight together: one thread adding them one dependent load after the other was most of this kernel's duration), then the
    // butterfly — a fixed order of additions: the value does not depend on timing.
    if (threadIdx.x >= 64) return;
    float tot = 0.f;
    for (int t = 0; t < A.nl; ++t)
      for (int b = lane; b < A.loss_n[t]; b += 64) tot += A.loss_part[t][b] * A.loss_coef[t];
    tot = wave_sum(tot);
    if (lane == 0) atomicAdd(A.loss, tot);  // (two calls on different pairs may finish at the same time)
    return;
  }
  const int grow = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (grow >= A.rows) return;
  int si = 0;
#pragma unroll
  for (int t = 1; t < 6; ++t) if (grow >= A.s[t].row0) si = t;
  const ClSet& S = A.s[si];
  if (!S.dv) return;
  const int row = grow - S.row0, d = S.d;
  if (row < S.w0 || row >= S.w0 + S.wn) return;  // data-parallel: a rank keeps the gradient rows of its own videos / clips
  // One memory round trip: every load of the row — the violation counts, the half-term strips, the other set's rows, the raw row and
  // the gradient it adds to — issues before the first reduction (counts first, then chunk by chunk, was three dependent trips).
  // A lane owns the 4-element chunks lane, lane + 64, ... of the row (d % 32 == 0, d <= 1024).
  constexpr int MAXC = 4;
  const int nch = d / 4;
  float c1v[3] = {0.f, 0.f, 0.f};  // (counts: exact in fp32, any order)
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (t < S.nc && lane < S.c[t].cs) c1v[t] = S.c[t].c1p[lane * S.c[t].c1s + row] + S.c[t].c1q[lane * S.c[t].c1s + row];
  const float inv = S.inv[row];
  f32x4_t dx[3][MAXC], a[MAXC], old[MAXC];
  u32x2_t ob[3][MAXC];
#pragma unroll
  for (int q = 0; q < MAXC; ++q) {
    const int ch = lane + 64 * q;
    if (ch < nch) {
      const long o = (long)row * d + ch * 4;
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (t < S.nc) {
          dx[t][q] = *reinterpret_cast<const f32x4_t*>(S.c[t].dX + o);
          if (S.c[t].cs > 1) {  // column-split partials: all loads first, then the sum in a fixed order
            f32x4_t part[7];
#pragma unroll
            for (int u = 1; u < 8; ++u) part[u - 1] = *reinterpret_cast<const f32x4_t*>(S.c[t].dX + min(u, S.c[t].cs - 1) * S.c[t].dxs + o);
#pragma unroll
            for (int u = 1; u < 8; ++u) if (u < S.c[t].cs) dx[t][q] += part[u - 1];
          }
          ob[t][q] = *reinterpret_cast<const u32x2_t*>(S.c[t].other + o);
        }
      a[q] = *reinterpret_cast<const f32x4_t*>(S.v + (long)row * S.ldv + ch * 4);
      old[q] = *reinterpret_cast<const f32x4_t*>(S.dv + (long)(row - S.w0) * d + ch * 4);
    }
  }
  wave_sum_n<3>(c1v);
  f32x4_t g[MAXC];
  float dot = 0.f;
#pragma unroll
  for (int q = 0; q < MAXC; ++q) {
    const int ch = lane + 64 * q;
    g[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (ch < nch) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
        if (t < S.nc) {
          const f32x4_t of = {bflo(ob[t][q][0]), bfhi(ob[t][q][0]), bflo(ob[t][q][1]), bfhi(ob[t][q][1])};
          g[q] += S.c[t].coef * dx[t][q] + (-S.c[t].dcoef * c1v[t]) * of;
        }
      a[q] = a[q] * inv;
      dot += a[q][0] * g[q][0] + a[q][1] * g[q][1] + a[q][2] * g[q][2] + a[q][3] * g[q][3];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int q = 0; q < MAXC; ++q) {
    const int ch = lane + 64 * q;
    if (ch < nch) *reinterpret_cast<f32x4_t*>(S.dv + (long)(row - S.w0) * d + ch * 4) = old[q] + (g[q] - a[q] * dot) * inv;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
namespace {
inline int pad16(int n) { return (n + 15) & ~15; }
int g_cl_col_split = 0;  // coot_set_option("cl_col_split", k): 0 = by size, k >= 1 = k column splits per strip (tests)
// column splits of a 16-row strip over Np columns and the columns per split (multiple of 32); every split has columns
inline void col_split(int Np, int& cs, int& cols) {
  const int Np32 = (Np + 31) & ~31;
  cs = g_cl_col_split > 0 ? g_cl_col_split : Np32 / 256;
  cs = cs < 1 ? 1 : (cs > 8 ? 8 : cs);
  for (;; --cs) {
    cols = ((Np32 + cs - 1) / cs + 31) & ~31;
    if (cs == 1 || (cs - 1) * cols < Np) break;
  }
}
struct FBump {
  char* base; size_t cap; size_t off = 0; bool overflow = false;
  FBump(void* b, size_t c) : base((char*)b), cap(c) {}
  template <typename T> T* get(size_t n) {
    off = (off + 255) & ~(size_t)255;
    char* p = base ? base + off : nullptr;
    off += n * sizeof(T);
    if (base && off > cap) overflow = true;
    return (T*)p;
  }
};
struct PairBufs { bf16_t *a, *b, *aT, *bT; float *inva, *invb, *dab, *daa, *dbb; };
struct HalfBufs { float *dX, *c1, *lp; };
struct FusedLayout { PairBufs p[3]; HalfBufs h[CL_MAX_HALF]; };
// half-term order: for pair p (0 high, 1 low, 2 ctx): 4p+0 = (A,B) primary, 4p+1 = (B,A), 4p+2 = (A,A), 4p+3 = (B,B)
void layout_fused(int n_high, int n_low, int d_high, int d_low, FBump& A, FusedLayout& L) {
  const int Ns[3] = {n_high, n_low, n_high}, ds[3] = {d_high, d_low, d_low};
  for (int p = 0; p < 3; ++p) {
    const size_t Np = pad16(Ns[p]), d = ds[p];
    L.p[p].a = A.get<bf16_t>(Np * d); L.p[p].b = A.get<bf16_t>(Np * d); L.p[p].aT = A.get<bf16_t>(Np * d); L.p[p].bT = A.get<bf16_t>(Np * d);
    L.p[p].inva = A.get<float>(Np); L.p[p].invb = A.get<float>(Np); L.p[p].dab = A.get<float>(Np); L.p[p].daa = A.get<float>(Np);
    L.p[p].dbb = A.get<float>(Np);
    for (int q = 0; q < 4; ++q) {
      HalfBufs& h = L.h[4 * p + q];
      int cs, cols; col_split((int)Np, cs, cols);
      h.dX = A.get<float>(cs * Np * d); h.c1 = A.get<float>(cs * Np); h.lp = A.get<float>(cs * (Np / 16 + 1));
    }
  }
}
}  // namespace

void set_cl_col_split(int k) { g_cl_col_split = k; }
static int g_cl_small = 1;  // coot_set_option("cl_small", 0): small sets take the three-launch path as well (tests compare the two)
void set_cl_small(int on) { g_cl_small = on; }
// LDS bytes of a cl_small_kernel workgroup, 0 when the set does not qualify (one column split, <= 128 rows: the summation orders
// of cl_half are then reproduced exactly)
static size_t cl_small_smem(int N, int d) {
  const int Np = pad16(N), kw = (Np + 31) & ~31;
  int cs, cols; col_split(Np, cs, cols);
  if (N <= 0 || Np > 128 || cs != 1 || d % 32 != 0 || d > 1024) return 0;
  const size_t b = ((size_t)(Np + 16) * (d + CL_GP) + (size_t)16 * (kw + CL_GP)) * sizeof(bf16_t) + (size_t)Np * sizeof(float);
  return b <= 150 * 1024 ? b : 0;
}

size_t contrastive_fused_scratch_bytes(int n_high, int n_low, int d_high, int d_low) {
  FBump A(nullptr, 0); FusedLayout L; layout_fused(n_high, n_low, d_high, d_low, A, L); return A.off + 256;
}

// weights: w_pair[p] (alignment), w_self[p] (already includes the 1/2 of compute_cluster_loss)
int launch_contrastive_fused(const float* const v[6], float* const dv[6], int n_high, int n_low, int d_high, int d_low, const float w_pair[3],
                             const float w_self[3], float margin, float* loss, void* scratch, size_t scratch_bytes, hipStream_t st,
                             const long* ldv, const int* window, int pair_mask, const ClBlocks* blk) {
  // pair_mask: bit p set = pair p (0 vid | par, 1 clip | sent, 2 vid_ctx | par_ctx) is part of this call.  The pairs own
  // disjoint scratch buffers and gradient outputs and the loss word is added atomically, so two calls with complementary masks
  // may run on different streams (the step computes pairs 1 and 2 as soon as the local networks are done).
  auto active = [&](int p) { return ((pair_mask >> p) & 1) != 0; };
  {
    const bool hi_on = w_pair[0] != 0.f || w_self[0] != 0.f, lo_on = w_pair[1] != 0.f || w_self[1] != 0.f || w_pair[2] != 0.f || w_self[2] != 0.f;
    COOT_REQUIRE((!hi_on || d_high % 32 == 0) && (!lo_on || d_low % 32 == 0), "contrastive: embedding dims must be multiples of 32 (%d, %d)", d_high, d_low);
    COOT_REQUIRE(d_high <= 1024 && d_low <= 1024, "contrastive: embedding dims up to 1024 (%d, %d)", d_high, d_low);
  }
  FBump B(scratch, scratch_bytes); FusedLayout L; layout_fused(n_high, n_low, d_high, d_low, B, L);
  COOT_REQUIRE(!B.overflow, "contrastive: scratch too small (%zu < %zu)", scratch_bytes, B.off);
  const int Ns[3] = {n_high, n_low, n_high}, ds[3] = {d_high, d_low, d_low};
  const bool bwd = dv[0] != nullptr;
  ClNormArgs na; int rows = 0;
  for (int p = 0; p < 3; ++p) {
    ClPair& P = na.p[p]; const PairBufs& b = L.p[p];
    P.va = v[2 * p]; P.vb = v[2 * p + 1]; P.lda = ldv ? ldv[2 * p] : ds[p]; P.ldb = ldv ? ldv[2 * p + 1] : ds[p]; P.a = b.a; P.b = b.b; P.aT = b.aT; P.bT = b.bT; P.inva = b.inva; P.invb = b.invb;
    P.dab = b.dab; P.daa = b.daa; P.dbb = b.dbb; P.N = Ns[p]; P.Np = pad16(Ns[p]); P.d = ds[p];
    na.row0[p] = rows; rows += active(p) ? P.Np : 0;  // an inactive pair has no rows (cl_norm picks the LAST pair whose row0 <= row)
  }
  na.row0[3] = rows;
  na.blk.world = 0;
  if (blk) na.blk = *blk;
  // every active pair small enough for the LDS: its strip workgroups normalise their rows themselves (cl_small_kernel)
  bool small = g_cl_small != 0 && !blk && !window && rows > 0;
  size_t small_smem = 0;
  for (int p = 0; p < 3 && small; ++p)
    if (active(p)) {
      const size_t b = cl_small_smem(Ns[p], ds[p]);
      if (b == 0) small = false;
      if (b > small_smem) small_smem = b;
    }
  if (rows > 0 && !small) {
    int dmax = 0;
    for (int p = 0; p < 3; ++p) if (active(p) && ds[p] > dmax) dmax = ds[p];
    hipLaunchKernelGGL(cl_norm_kernel, dim3(rows / 16), dim3(1024), (size_t)32 * (dmax + CL_TP) * sizeof(bf16_t), st, na);  // (rows: multiples of 16 per pair)
    COOT_CHECK_LAUNCH("cl_norm");
  }
  ClHalfArgs ha; ha.nh = 0; ha.margin = margin; int nblk = 0, maxcols = 0, maxd = 0;
  ClSmallArgs sa; sa.nh = 0; sa.margin = margin;
  bool wrote_norm[3][2] = {{false, false}, {false, false}, {false, false}};
  ClFinishArgs fa; fa.nl = 0; fa.loss = loss;
  int hidx[3][4];
  for (int p = 0; p < 3; ++p) {
    const PairBufs& b = L.p[p];
    const int N = Ns[p], Np = pad16(N), d = ds[p];
    for (int q = 0; q < 4; ++q) {
      hidx[p][q] = -1;
      const bool on = active(p) && (q < 2 ? (w_pair[p] != 0.f) : (w_self[p] != 0.f));
      if (!on || N <= 0) continue;
      if (q == 1 && !bwd) continue;  // the swapped half only provides the gradient wrt B (and its violation counts)
      ClHalf& H = ha.h[ha.nh];
      const HalfBufs& hb = L.h[4 * p + q];
      H.X = (q == 0 || q == 2) ? b.a : b.b;
      H.Y = (q == 0 || q == 3) ? b.b : b.a;
      H.YT = (q == 0 || q == 3) ? b.bT : b.aT;
      H.diag = q < 2 ? b.dab : (q == 2 ? b.daa : b.dbb);
      H.dX = hb.dX; H.c1 = hb.c1; H.loss_part = hb.lp; H.N = N; H.Np = Np; H.d = d; H.blk0 = nblk; H.primary = (q != 1);
      H.w0 = 0; H.wn = N;
      col_split(Np, H.cs, H.cols);
      if (window) { H.w0 = window[p == 1 ? 2 : 0]; H.wn = window[p == 1 ? 3 : 1]; }
      if (H.primary) {  // the strips that hold rows of the window (all of them on one GPU): this call's share of the term
        const int s0 = H.w0 / 16, s1 = (H.w0 + H.wn + 15) / 16;
        fa.loss_part[fa.nl] = hb.lp + s0 * H.cs; fa.loss_n[fa.nl] = H.wn > 0 ? (s1 - s0) * H.cs : 0; fa.loss_coef[fa.nl] = (q == 0 ? w_pair[p] : w_self[p]) / ((float)N * (float)N);
        ++fa.nl;
      }
      if (small) {  // the same half-term from the raw rows; the first half-term whose X is a set leaves its normalised rows for cl_finish
        ClSmall& S = sa.h[sa.nh++];
        const int xs = (q == 0 || q == 2) ? 0 : 1, ys = (q == 0 || q == 3) ? 1 : 0;
        S.vx = v[2 * p + xs]; S.vy = v[2 * p + ys]; S.ldx = ldv ? ldv[2 * p + xs] : d; S.ldy = ldv ? ldv[2 * p + ys] : d;
        const bool writer = bwd && !wrote_norm[p][xs];
        wrote_norm[p][xs] = wrote_norm[p][xs] || writer;
        S.xn = writer ? (xs ? b.b : b.a) : nullptr; S.invx = writer ? (xs ? b.invb : b.inva) : nullptr;
        S.dX = hb.dX; S.c1 = hb.c1; S.loss_part = hb.lp; S.N = N; S.Np = Np; S.d = d; S.blk0 = nblk; S.primary = H.primary; S.same = q >= 2;
      }
      hidx[p][q] = ha.nh++;
      nblk += (Np / 16) * H.cs;
      if (H.cols > maxcols) maxcols = H.cols;
      if (d > maxd) maxd = d;
    }
  }
  ha.nblk = nblk;
  if (nblk > 0 && small) {
    int dmax = 0;
    for (int h = 0; h < sa.nh; ++h) if (sa.h[h].d > dmax) dmax = sa.h[h].d;
    if (dmax <= 256) hipLaunchKernelGGL(cl_small_kernel<1>, dim3(nblk), dim3(64 * CLS_NW), small_smem, st, sa);
    else if (dmax <= 512) hipLaunchKernelGGL(cl_small_kernel<2>, dim3(nblk), dim3(64 * CLS_NW), small_smem, st, sa);
    else if (dmax <= 768) hipLaunchKernelGGL(cl_small_kernel<3>, dim3(nblk), dim3(64 * CLS_NW), small_smem, st, sa);
    else hipLaunchKernelGGL(cl_small_kernel<4>, dim3(nblk), dim3(64 * CLS_NW), small_smem, st, sa);
    COOT_CHECK_LAUNCH("cl_small");
  } else if (nblk > 0) {
    const size_t smem = (size_t)16 * (maxcols + CL_GP + maxd + CL_GP) * sizeof(bf16_t);  // G strip + the strip's own rows
    COOT_REQUIRE(smem <= 150 * 1024, "contrastive: %d columns per strip workgroup exceed the LDS strip (max ~4600)", maxcols);
    hipLaunchKernelGGL(cl_half_kernel, dim3(nblk), dim3(64 * CL_NW), smem, st, ha);
    COOT_CHECK_LAUNCH("cl_half");
  }
  // finish: per set contributions
  int frows = 0;
  for (int s = 0; s < 6; ++s) {
    const int p = s / 2, isb = s & 1;
    ClSet& S = fa.s[s]; const PairBufs& b = L.p[p];
    S.v = v[s]; S.ldv = ldv ? ldv[s] : ds[p]; S.inv = isb ? b.invb : b.inva; S.dv = (bwd && active(p)) ? dv[s] : nullptr; S.N = Ns[p]; S.d = ds[p]; S.nc = 0; S.row0 = frows;
    S.w0 = 0; S.wn = Ns[p];
    if (window) { S.w0 = window[p == 1 ? 2 : 0]; S.wn = window[p == 1 ? 3 : 1]; }  // {high row0, high rows, low row0, low rows}
    frows += active(p) ? Ns[p] : 0;  // (cl_finish picks the LAST set whose row0 <= row: a set without rows is never picked)
    if (!bwd || !active(p)) continue;
    const float n2 = (float)Ns[p] * (float)Ns[p];
    int cs, cols; col_split(pad16(Ns[p]), cs, cols);
    const long c1s = pad16(Ns[p]), dxs = c1s * ds[p];
    if (hidx[p][0] >= 0) {  // alignment term: this set's half + diagonal (c1 of both orientations) against the other set's rows
      ClContrib& c = S.c[S.nc++];
      c.dX = L.h[4 * p + (isb ? 1 : 0)].dX; c.coef = w_pair[p] / n2; c.c1p = L.h[4 * p].c1; c.c1q = L.h[4 * p + 1].c1;
      c.dcoef = w_pair[p] / n2; c.other = isb ? b.a : b.b; c.cs = cs; c.dxs = dxs; c.c1s = c1s;
    }
    if (hidx[p][2 + isb] >= 0) {  // cluster term L(A, A): gradient = 2 * (G . A) with diagonal -2 c1
      ClContrib& c = S.c[S.nc++];
      c.dX = L.h[4 * p + 2 + isb].dX; c.coef = 2.f * w_self[p] / n2; c.c1p = L.h[4 * p + 2 + isb].c1; c.c1q = c.c1p;
      c.dcoef = 2.f * w_self[p] / n2; c.other = isb ? b.b : b.a;  // diagonal of G + G^T: -2 c1 each
      c.cs = cs; c.dxs = dxs; c.c1s = c1s;
    }
  }
  fa.rows = frows;
  hipLaunchKernelGGL(cl_finish_kernel, dim3((frows + 3) / 4 + 1), dim3(256), 0, st, fa);  // + the loss workgroup
  COOT_CHECK_LAUNCH("cl_finish");
  return 0;
}

}  // namespace coot
