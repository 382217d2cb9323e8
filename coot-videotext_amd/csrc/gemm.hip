// bf16 MFMA GEMM kernels for gfx950 (wave64, v_mfma_f32_16x16x32_bf16).
//
// Main loop: tile BM tokens x 128 features x 64 K per step, 256 threads = 4 waves (2 x 2), each wave owns
// (BM/2) x 64 of the output as 16x16 MFMA fragments.  Operands go HBM -> registers -> LDS (register staged,
// double-buffered LDS, one barrier per K step); rows are padded to 80 elements (160 B) which makes every
// ds_read_b128 fragment read bank-conflict free (checked against the gfx950 lane-group table).
//
// Epilogue: the fp32 accumulator tile is staged through LDS (the operand buffers are dead by then) so that
// every global access of the fused epilogue (bias, residual, saved pre-activation, GELU', output) is a
// 16-byte, row-contiguous access: a lane owns 8 consecutive features of one token.  The straightforward
// "store what the MFMA layout gives you" epilogue (8-byte stores in 32-byte runs) cost more than the K loop
// at K = 384 (profiles/README.md).
#include "gemm.h"
#include "det.h"
#include <algorithm>
#include "rowops.h"

namespace coot {

constexpr int BN = 128;
constexpr int BK = 64;
constexpr int PITCH = BK + 16;  // elements; 160 B rows
constexpr int CPITCH = BN + 4;  // floats; epilogue staging tile rows (528 B: conflict-free b128 writes)

__device__ __forceinline__ u32x4_t load16_guard(const bf16_t* p, bool ok) {
  u32x4_t z = {0u, 0u, 0u, 0u};
  if (ok) z = *reinterpret_cast<const u32x4_t*>(p);
  return z;
}

__device__ __forceinline__ void unpack8(u32x4_t u, float* v) {
  v[0] = bflo(u[0]); v[1] = bfhi(u[0]); v[2] = bflo(u[1]); v[3] = bfhi(u[1]);
  v[4] = bflo(u[2]); v[5] = bfhi(u[2]); v[6] = bflo(u[3]); v[7] = bfhi(u[3]);
}
__device__ __forceinline__ u32x4_t pack8(const float* v) {
  return u32x4_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
}

// Fused epilogue on 8 consecutive features [col, col+8) of token `row`; v = raw accumulators.
__device__ __forceinline__ void epilogue8(const GemmEpi& e, float* v, int row, int col, long zo, int N, const float* bias8, unsigned dkey) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = v[j] * e.alpha + bias8[j];
  float dsc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dsc[j] = 1.f;
  if (e.drop_thr) {
    drop_scales_key<8>(dkey, (unsigned long long)row * e.drop_ld + zo + col, e.drop_thr, e.drop_inv_keep, dsc);
    if (e.act != 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= dsc[j];
    }
  }
  if (e.save_pre) *reinterpret_cast<u32x4_t*>(e.save_pre + (long)row * e.ldpre + zo + col) = pack8(v);
  if (e.act == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
  }
  if (e.pe) {
    const int pos = row < e.pe_T0 ? row % e.pe_L : (row - e.pe_T0) % e.pe_L2;
    const f32x4_t* pp = reinterpret_cast<const f32x4_t*>(e.pe + (long)pos * N + col);
    f32x4_t p0 = pp[0], p1 = pp[1];
    v[0] += p0[0]; v[1] += p0[1]; v[2] += p0[2]; v[3] += p0[3]; v[4] += p1[0]; v[5] += p1[1]; v[6] += p1[2]; v[7] += p1[3];
  }
  if (e.res) {
    float r[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(e.res + (long)row * e.ldres + zo + col), r);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += r[j];
  }
  if (e.res32) {
    const f32x4_t* rp = reinterpret_cast<const f32x4_t*>(e.res32 + (long)row * e.ldres32 + zo + col);
    f32x4_t r0 = rp[0], r1 = rp[1];
    v[0] += r0[0]; v[1] += r0[1]; v[2] += r0[2]; v[3] += r0[3]; v[4] += r1[0]; v[5] += r1[1]; v[6] += r1[2]; v[7] += r1[3];
  }
  if (e.rowscale) {
    const float rsv = e.rowscale[row];
    float d[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(e.diag_src + (long)row * e.lddiag + col), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += rsv * d[j];
  }
  if (e.act == 2) {  // backward through GELU (and its dropout), after the residual adds
    float a[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(e.aux + (long)row * e.ldaux + zo + col), a);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(a[j]) * dsc[j];
  }
  if (e.out_f32) {
    f32x4_t* op = reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(e.out) + (long)row * e.ldc + zo + col);
    if (e.accumulate) {
      f32x4_t o0 = op[0], o1 = op[1];
      op[0] = f32x4_t{o0[0] + v[0], o0[1] + v[1], o0[2] + v[2], o0[3] + v[3]};
      op[1] = f32x4_t{o1[0] + v[4], o1[1] + v[5], o1[2] + v[6], o1[3] + v[7]};
    } else {
      op[0] = f32x4_t{v[0], v[1], v[2], v[3]};
      op[1] = f32x4_t{v[4], v[5], v[6], v[7]};
    }
  } else {
    *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(e.out) + (long)row * e.ldc + zo + col) = pack8(v);
  }
}

template <int BM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT g) {
  constexpr int TM = BM / 32;        // token fragments per wave
  constexpr int XCH = BM * 8 / 256;  // 16-byte chunks per thread for the token tile
  constexpr int WCH = BN * 8 / 256;
  constexpr int OPER_ELEMS = 2 * (BM + BN) * PITCH;                  // bf16 elements of the two double-buffered operands
  constexpr int STAGE_ELEMS = BM * CPITCH * 2;                        // fp32 staging tile, in bf16 units
  constexpr int SMEM_ELEMS = OPER_ELEMS > STAGE_ELEMS ? OPER_ELEMS : STAGE_ELEMS;
  __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM_ELEMS];
  bf16_t* Xs = smem;                       // [2][BM * PITCH]
  bf16_t* Ws = smem + 2 * BM * PITCH;      // [2][BN * PITCH]
  float* Cs = reinterpret_cast<float*>(smem);  // [BM][CPITCH] (after the K loop)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned dkey = g.epi.drop_thr ? drop_site_key(g.epi.drop_seed, g.epi.drop_seed_ptr, g.epi.drop_site) : 0u;
  // XCD-aware tile order: workgroups go round-robin to the 8 XCDs (linear id % 8), each with a private L2.  The column
  // blocks of one token slab share that slab, so they are given ids of the SAME residue, 8 apart (same L2, consecutive
  // in time): id = 8 nb G + 8 bx + (by % 8), by = 8 G + id % 8.  With id = bx + nb by every slab was pulled into nb L2s.
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_order) {
    const int nb = gridDim.x, id = blockIdx.x + nb * blockIdx.y, grp = id / (8 * nb), r = id - grp * (8 * nb);
    bx = r >> 3;
    by = grp * 8 + (r & 7);
  }
  const int row0 = by * BM, col0 = bx * BN;
  const int M = g.M_dev ? min(g.M, *g.M_dev) : g.M;
  if (row0 >= M) return;
  const int z = blockIdx.z;
  const bf16_t* X = g.X + z * g.zX;
  const bf16_t* W = g.W + z * g.zW;
  const int N = g.N, K = g.K;

  u32x4_t xr[XCH], wr[WCH];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      int c = tid + 256 * i, r = c >> 3, kc = (c & 7) * 8;
      int gr = row0 + r, gk = k0 + kc;
      xr[i] = load16_guard(X + (long)gr * g.ldx + gk, gr < M && gk < K);
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      int c = tid + 256 * i, r = c >> 3, kc = (c & 7) * 8;
      int gr = col0 + r, gk = k0 + kc;
      wr[i] = load16_guard(W + (long)gr * g.ldw + gk, gr < N && gk < K);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      int c = tid + 256 * i, r = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<u32x4_t*>(&Xs[buf * BM * PITCH + r * PITCH + kc]) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      int c = tid + 256 * i, r = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<u32x4_t*>(&Ws[buf * BN * PITCH + r * PITCH + kc]) = wr[i];
    }
  };

  f32x4_t acc[TM][4];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  const int frow = lane & 15, fk = (lane >> 4) * 8;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const bf16_t* xb = Xs + buf * BM * PITCH;
    const bf16_t* wb = Ws + buf * BN * PITCH;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8_t xf[TM], wf[4];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        xf[a] = *reinterpret_cast<const bf16x8_t*>(&xb[(wm * (BM / 2) + a * 16 + frow) * PITCH + kk * 32 + fk]);
#pragma unroll
      for (int b = 0; b < 4; ++b)
        wf[b] = *reinterpret_cast<const bf16x8_t*>(&wb[(wn * 64 + b * 16 + frow) * PITCH + kk * 32 + fk]);
      // weights = MFMA A operand, tokens = B operand: acc[a][b][j] = C[token a*16 + (lane&15)][feature b*16 + (lane>>4)*4 + j]
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = COOT_MFMA_16x16x32(wf[b], xf[a], acc[a][b]);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- stage the accumulator tile through LDS: [BM][128] fp32 -------------------------------------------
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int r = wm * (BM / 2) + a * 16 + (lane & 15), c = wn * 64 + b * 16 + (lane >> 4) * 4;
      *reinterpret_cast<f32x4_t*>(&Cs[r * CPITCH + c]) = acc[a][b];
    }
  __syncthreads();

  // ---- fused epilogue, lane = 8 consecutive features of one token ---------------------------------------
  const GemmEpi& e = g.epi;
  const long zo = z * g.zOut;
  const int cch = tid & 15, rg = tid >> 4;
  const int col = col0 + cch * 8;
  const bool cok = col < N;  // N % 8 == 0 (launcher)
  float bias8[8], csum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias8[j] = 0.f; csum[j] = 0.f; }
  if (cok && e.bias) {
    const f32x4_t* bp = reinterpret_cast<const f32x4_t*>(e.bias + zo + col);
    f32x4_t b0 = bp[0], b1 = bp[1];
    bias8[0] = b0[0]; bias8[1] = b0[1]; bias8[2] = b0[2]; bias8[3] = b0[3];
    bias8[4] = b1[0]; bias8[5] = b1[1]; bias8[6] = b1[2]; bias8[7] = b1[3];
  }
#pragma unroll 2
  for (int i = 0; i < BM / 16; ++i) {
    const int rl = rg + 16 * i, row = row0 + rl;
    if (!cok || row >= M) continue;
    float v[8];
    const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(&Cs[rl * CPITCH + cch * 8]);
    const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(&Cs[rl * CPITCH + cch * 8 + 4]);
    v[0] = c0[0]; v[1] = c0[1]; v[2] = c0[2]; v[3] = c0[3]; v[4] = c1[0]; v[5] = c1[1]; v[6] = c1[2]; v[7] = c1[3];
    epilogue8(e, v, row, col, zo, N, bias8, dkey);
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[j] += v[j];
  }
  if (e.colsum) {
    // lanes l, l+16, l+32, l+48 of a wave hold the same feature chunk
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = csum[j];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      csum[j] = s;
    }
    if (e.colsum_ws) {  // one partial row per row-block, reduced by reduce_partials_kernel (no atomics)
      __syncthreads();
      float* red = Cs;  // [4][128]
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave * 128 + cch * 8 + j] = csum[j];
      }
      __syncthreads();
      if (tid < 128 && col0 + tid < N)
        e.colsum_ws[(long)(row0 / BM) * e.ld_colsum_ws + zo + col0 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
    } else if (lane < 16 && cok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc_add(e.colsum + zo + col + j, csum[j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Small-M variant (global networks: 64..512 token rows; the step's critical path is a chain of ~80 such GEMMs).
// The LDS-staged kernel above runs its 6..18 k-steps as a dependent load -> LDS -> MFMA chain on a handful of
// workgroups: 10..15 us for a microsecond of work.  Here a wave owns a 16-token x (16 NF)-feature tile and takes its
// MFMA fragments straight from global memory (the operands are L2 resident), up to 6 k-steps of loads in flight at
// once, no LDS, no barriers: ~2 memory round trips per launch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void epilogue4(const GemmEpi& e, float* v, int row, int col, long zo, int N, const float* bias4, unsigned dkey) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = v[j] * e.alpha + bias4[j];
  float dsc[4] = {1.f, 1.f, 1.f, 1.f};
  if (e.drop_thr) {
    drop_scales_key<4>(dkey, (unsigned long long)row * e.drop_ld + zo + col, e.drop_thr, e.drop_inv_keep, dsc);
    if (e.act != 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= dsc[j];
    }
  }
  auto ld4 = [](const bf16_t* p, float* r) {
    const u32x2_t u = *reinterpret_cast<const u32x2_t*>(p);
    r[0] = bflo(u[0]); r[1] = bfhi(u[0]); r[2] = bflo(u[1]); r[3] = bfhi(u[1]);
  };
  auto st4 = [](bf16_t* p, const float* r) { *reinterpret_cast<u32x2_t*>(p) = u32x2_t{pack2bf(r[0], r[1]), pack2bf(r[2], r[3])}; };
  if (e.save_pre) st4(e.save_pre + (long)row * e.ldpre + zo + col, v);
  if (e.act == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = gelu_f(v[j]);
  }
  if (e.pe) {
    const int pos = row < e.pe_T0 ? row % e.pe_L : (row - e.pe_T0) % e.pe_L2;
    const f32x4_t p0 = *reinterpret_cast<const f32x4_t*>(e.pe + (long)pos * N + col);
    v[0] += p0[0]; v[1] += p0[1]; v[2] += p0[2]; v[3] += p0[3];
  }
  if (e.res) {
    float r[4];
    ld4(e.res + (long)row * e.ldres + zo + col, r);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += r[j];
  }
  if (e.res32) {
    const f32x4_t r0 = *reinterpret_cast<const f32x4_t*>(e.res32 + (long)row * e.ldres32 + zo + col);
    v[0] += r0[0]; v[1] += r0[1]; v[2] += r0[2]; v[3] += r0[3];
  }
  if (e.rowscale) {
    const float rsv = e.rowscale[row];
    float d[4];
    ld4(e.diag_src + (long)row * e.lddiag + col, d);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += rsv * d[j];
  }
  if (e.act == 2) {
    float a[4];
    ld4(e.aux + (long)row * e.ldaux + zo + col, a);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= gelu_grad_f(a[j]) * dsc[j];
  }
  if (e.out_f32) {
    f32x4_t* op = reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(e.out) + (long)row * e.ldc + zo + col);
    if (e.accumulate) { const f32x4_t o0 = op[0]; op[0] = f32x4_t{o0[0] + v[0], o0[1] + v[1], o0[2] + v[2], o0[3] + v[3]}; }
    else op[0] = f32x4_t{v[0], v[1], v[2], v[3]};
  } else {
    st4(reinterpret_cast<bf16_t*>(e.out) + (long)row * e.ldc + zo + col, v);
  }
}

template <int NF>
__global__ __launch_bounds__(256) void gemm_nt_small_kernel(GemmNT g) {
  const unsigned dkey = g.epi.drop_thr ? drop_site_key(g.epi.drop_seed, g.epi.drop_seed_ptr, g.epi.drop_site) : 0u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int z = blockIdx.z;
  const int M = g.M_dev ? min(g.M, *g.M_dev) : g.M, N = g.N, K = g.K;
  const int row0 = blockIdx.y * 16, col0 = (blockIdx.x * 4 + wave) * 16 * NF;
  if (row0 >= M || col0 >= N) return;
  const bf16_t* X = g.X + z * g.zX;
  const bf16_t* W = g.W + z * g.zW;
  const int xr = min(row0 + l15, M - 1);  // rows past M are computed on a clamped row and never stored
  const bf16_t* xp = X + (long)xr * g.ldx + l4 * 8;
  const bf16_t* wp[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) wp[f] = W + (long)min(col0 + f * 16 + l15, N - 1) * g.ldw + l4 * 8;
  f32x4_t acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int KS = 6;  // k-steps of loads in flight
  for (int k0 = 0; k0 < K; k0 += 32 * KS) {
    bf16x8_t xf[KS], wf[NF][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = k0 + s * 32 + l4 * 8;
      const bool ok = k < K;  // K % 8 == 0: a lane's 8 elements are all inside or all outside
      xf[s] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) xf[s] = *reinterpret_cast<const bf16x8_t*>(xp + k0 + s * 32);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        wf[f][s] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) wf[f][s] = *reinterpret_cast<const bf16x8_t*>(wp[f] + k0 + s * 32);
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f] = COOT_MFMA_16x16x32(wf[f][s], xf[s], acc[f]);
  }
  // acc[f][j] = C[token row0 + l15][feature col0 + 16 f + 4 l4 + j]
  const GemmEpi& e = g.epi;
  const long zo = z * g.zOut;
  const int row = row0 + l15;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int col = col0 + f * 16 + l4 * 4;
    const bool ok = row < M && col < N;  // N % 8 == 0 and col % 4 == 0: a lane's 4 features are all inside or all outside
    float v[4] = {acc[f][0], acc[f][1], acc[f][2], acc[f][3]};
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok && e.bias) { const f32x4_t b = *reinterpret_cast<const f32x4_t*>(e.bias + zo + col); b4[0] = b[0]; b4[1] = b[1]; b4[2] = b[2]; b4[3] = b[3]; }
    if (ok) epilogue4(e, v, row, col, zo, N, b4, dkey);
    if (e.colsum) {  // 16 tokens of this wave share a feature: reduce over l15, one atomic per feature (<= M/16 adds per address)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = ok ? v[j] : 0.f;
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
        if (l15 == 0 && col < N) acc_add(e.colsum + zo + col + j, s);
      }
    }
  }
}

// ---- optional per-launch timing with HIP events (bench.py roofline leg) -------------------------------
struct TimingSlot { hipEvent_t a, b; double flops; int kind; int big_k; };
static int g_timing_on = 0;
static TimingSlot g_slots[8192];
static int g_nslots = 0, g_slots_created = 0;
void gemm_timing_enable(int on) { g_timing_on = on; g_nslots = 0; }
// selector: 0 = every MFMA GEMM launch (all kinds), 1 = input-FC instances of gemm_nt (K >= 1024),
//           2 = gemm_nt (LDS staged), 3 = gemm_nt_small, 4 = gemm_tn (+ its split reduction), 5 = fused token-tile chains
int gemm_timing_collect(int selector, double* ms, double* flops, int* launches) {
  double tm = 0, fl = 0; int n = 0;
  for (int i = 0; i < g_nslots; ++i) {
    const TimingSlot& sl = g_slots[i];
    if (selector == 1 && !(sl.kind == TIMING_NT && sl.big_k)) continue;
    if (selector >= 2 && sl.kind != selector) continue;
    if (hipEventSynchronize(sl.b) != hipSuccess) return -1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, sl.a, sl.b) != hipSuccess) return -1;
    tm += t; fl += sl.flops; ++n;
  }
  *ms = tm; *flops = fl; *launches = n;
  return 0;
}
void* timing_begin(int kind, double flops, int big_k, hipStream_t stream) {
  if (!g_timing_on || g_nslots >= 8192) return nullptr;
  TimingSlot* s = &g_slots[g_nslots];
  if (g_nslots >= g_slots_created) { (void)hipEventCreate(&s->a); (void)hipEventCreate(&s->b); g_slots_created = g_nslots + 1; }
  ++g_nslots;
  s->flops = flops; s->kind = kind; s->big_k = big_k;
  (void)hipEventRecord(s->a, stream);
  return s;
}
void timing_end(void* slot, hipStream_t stream) {
  if (slot) (void)hipEventRecord(static_cast<TimingSlot*>(slot)->b, stream);
}

static int g_xcd_order = 7;  // coot_set_option("xcd_order", bits): 1 = gemm_nt tile order, 2 = weight-gradient tiles of one split on one XCD, 4 = short attention (sequence, head) order
static thread_local gemm_weight_hook_t g_weight_hook = nullptr;
void set_gemm_weight_hook(gemm_weight_hook_t fn) { g_weight_hook = fn; }
int launch_gemm_nt(const GemmNT& g_in, hipStream_t stream) {
  GemmNT g = g_in;
  COOT_REQUIRE(g.X && g.W && g.epi.out, "gemm_nt: null operand");
  if (g_weight_hook) { const int rc = g_weight_hook(g.W, stream); if (rc) return rc; }
  COOT_REQUIRE(g.K % 8 == 0 && g.ldx % 8 == 0 && g.ldw % 8 == 0 && g.zX % 8 == 0 && g.zW % 8 == 0,
               "gemm_nt: K/ldx/ldw must be multiples of 8 (K=%d ldx=%ld ldw=%ld)", g.K, g.ldx, g.ldw);
  COOT_REQUIRE(g.N % 8 == 0 && g.epi.ldc % 8 == 0 && g.zOut % 8 == 0, "gemm_nt: N/ldc must be multiples of 8 (N=%d ldc=%ld)", g.N, g.epi.ldc);
  COOT_REQUIRE(!(g.epi.accumulate && !g.epi.out_f32), "gemm_nt: accumulate needs fp32 out");
  COOT_REQUIRE(g.epi.ldres % 8 == 0 && g.epi.ldaux % 8 == 0 && g.epi.ldpre % 8 == 0 && g.epi.ldres32 % 4 == 0 && g.epi.lddiag % 8 == 0,
               "gemm_nt: epilogue strides must be multiples of 8");
  if (g.M <= 0 || g.N <= 0) return 0;
  g.xcd_order = g_xcd_order & 1;
  if (g.M <= 512) {  // global networks / loss strips: direct-from-L2 fragments, no LDS
    void* ts = timing_begin(TIMING_NT_SMALL, 2.0 * g.M * g.N * g.K * g.groups, 0, stream);
    g.epi.colsum_ws = nullptr;
    if ((long)((g.M + 15) / 16) * ((g.N + 31) / 32) >= 512) {
      dim3 grid((g.N + 4 * 32 - 1) / (4 * 32), (g.M + 15) / 16, g.groups);
      hipLaunchKernelGGL(gemm_nt_small_kernel<2>, grid, dim3(256), 0, stream, g);
    } else {
      dim3 grid((g.N + 4 * 16 - 1) / (4 * 16), (g.M + 15) / 16, g.groups);
      hipLaunchKernelGGL(gemm_nt_small_kernel<1>, grid, dim3(256), 0, stream, g);
    }
    timing_end(ts, stream);
    COOT_CHECK_LAUNCH("gemm_nt_small");
    return 0;
  }
  const int nb = (g.N + BN - 1) / BN;
  // small-M problems: halve the token tile to get more workgroups onto the 256 CUs
  const long blocks128 = (long)((g.M + 127) / 128) * nb * g.groups;
  const bool big = blocks128 >= 384;
  const int row_blocks = big ? (g.M + 127) / 128 : (g.M + 63) / 64;
  const int ccols = g.groups > 1 ? (int)(g.groups * g.zOut) : g.N;
  if (g.epi.colsum) {
    g.epi.colsum_ws = partials_workspace((size_t)row_blocks * ccols);
    g.epi.ld_colsum_ws = ccols;
  }
  void* ts = timing_begin(TIMING_NT, 2.0 * g.M * g.N * g.K * g.groups, g.K >= 1024, stream);
  const dim3 grid(nb, (row_blocks + 7) / 8 * 8, g.groups);  // whole groups of 8 token slabs (XCD-aware tile order)
  if (big) hipLaunchKernelGGL(gemm_nt_kernel<128>, grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL(gemm_nt_kernel<64>, grid, dim3(256), 0, stream, g);
  timing_end(ts, stream);
  COOT_CHECK_LAUNCH("gemm_nt");
  if (g.epi.colsum_ws) return launch_reduce_partials(g.epi.colsum_ws, row_blocks, ccols, ccols, g.epi.colsum, stream);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// TN: C[Mo,No] += alpha * sum_t A[t,Mo] * B[t,No]   (reduction over the row index)
// Both operands are loaded row-major ([t][col], coalesced along col) and the MFMA fragments are formed
// either with the LDS transpose read ds_read_b64_tr_b16 (mode 0) or from a transposed LDS image written
// with 2-byte stores (mode 1, fallback).  The t range is split over workgroups; each split writes its fp32
// partial tile to a workspace with coalesced 16-byte stores and a second kernel reduces the splits into C
// (cross-XCD fp32 atomics were the bottleneck of the one-pass version; they remain as the no-workspace path).
// ---------------------------------------------------------------------------------------------
constexpr int TN_BT = 64;             // t rows per step
constexpr int TN_BC = 128;            // columns per operand tile
constexpr int TN_PITCH = TN_BC + 16;  // mode 0: [t][col], 288-byte rows (8 consecutive rows -> 64 distinct banks)

template <int MODE>
__device__ __forceinline__ void gemm_tn_body(const GemmTN& g, int bx, int by, int bz, int t_per_split, int splits, float* ws, int direct) {
  // mode 0: As[t][col] (pitch TN_PITCH).  mode 1: At[col][t] (pitch PITCH).
  constexpr int ASZ = MODE == 0 ? TN_BT * TN_PITCH : TN_BC * PITCH;
  constexpr int OPER_ELEMS = 4 * ASZ;
  constexpr int STAGE_ELEMS = TN_BC * CPITCH * 2;
  constexpr int SMEM_ELEMS = OPER_ELEMS > STAGE_ELEMS ? OPER_ELEMS : STAGE_ELEMS;
  __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM_ELEMS];
  bf16_t* As = smem;            // [2][ASZ]
  bf16_t* Bs = smem + 2 * ASZ;  // [2][ASZ]
  float* Cs = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = bz % g.groups, split = bz / g.groups;
  const int m0 = by * TN_BC, n0 = bx * TN_BC;
  const int t_begin = split * t_per_split;
  const int t_end = min(g.T, t_begin + t_per_split);
  const bf16_t* A = g.A + z * g.zA;
  const bf16_t* B = g.B + z * g.zB;

  u32x4_t ar[4], br[4];
  auto gload = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int c = tid + 256 * i, r = c >> 4, cc = (c & 15) * 8;
      int t = t0 + r;
      ar[i] = load16_guard(A + (long)t * g.lda + m0 + cc, t < t_end && m0 + cc < g.Mo);
      br[i] = load16_guard(B + (long)t * g.ldb + n0 + cc, t < t_end && n0 + cc < g.No);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int c = tid + 256 * i, r = c >> 4, cc = (c & 15) * 8;
      if (MODE == 0) {
        *reinterpret_cast<u32x4_t*>(&As[buf * ASZ + r * TN_PITCH + cc]) = ar[i];
        *reinterpret_cast<u32x4_t*>(&Bs[buf * ASZ + r * TN_PITCH + cc]) = br[i];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          As[buf * ASZ + (cc + 2 * j) * PITCH + r] = (bf16_t)(ar[i][j] & 0xFFFFu);
          As[buf * ASZ + (cc + 2 * j + 1) * PITCH + r] = (bf16_t)(ar[i][j] >> 16);
          Bs[buf * ASZ + (cc + 2 * j) * PITCH + r] = (bf16_t)(br[i][j] & 0xFFFFu);
          Bs[buf * ASZ + (cc + 2 * j + 1) * PITCH + r] = (bf16_t)(br[i][j] >> 16);
        }
      }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // column sums of the A operand (bias gradient), by the workgroups of the first column tile: a thread meets the same
  // 8-column chunk (tid & 15) in every row group, out-of-range rows were loaded as zeros
  const bool do_cs = g.a_colsum != nullptr && bx == 0;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto cs_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cs[0] += bflo(ar[i][0]); cs[1] += bfhi(ar[i][0]); cs[2] += bflo(ar[i][1]); cs[3] += bfhi(ar[i][1]);
      cs[4] += bflo(ar[i][2]); cs[5] += bfhi(ar[i][2]); cs[6] += bflo(ar[i][3]); cs[7] += bfhi(ar[i][3]);
    }
  };

  const int nsteps = t_begin < t_end ? (t_end - t_begin + TN_BT - 1) / TN_BT : 0;
  if (nsteps > 0) {
    gload(t_begin);
    if (do_cs) cs_acc();
    sstore(0);
  }
  __syncthreads();
  const int grp = lane >> 4, p = lane & 15;
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) gload(t_begin + (st + 1) * TN_BT);
    const bf16_t* ab = As + buf * ASZ;
    const bf16_t* bb = Bs + buf * ASZ;
#pragma unroll
    for (int ks = 0; ks < TN_BT / 32; ++ks) {
      bf16x8_t af[4], bfr[4];
      if (MODE == 0) {
        // k-slot (grp*8 + j)     <-> t = ks*32 +      4*grp + j   (first  tr read)
        // k-slot (grp*8 + 4 + j) <-> t = ks*32 + 16 + 4*grp + j   (second tr read)
        // lane p of a 16-lane group supplies the 8-byte row chunk [t = base + (p>>2)][c0 + (p&3)*4 .. +3]
        const int trow = ks * 32 + 4 * grp + (p >> 2);
        const int tcol = (p & 3) * 4;
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const bf16_t* pa = &ab[trow * TN_PITCH + wm * 64 + a * 16 + tcol];
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pa));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pa + 16 * TN_PITCH));
          s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          af[a] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bf16_t* pb = &bb[trow * TN_PITCH + wn * 64 + b * 16 + tcol];
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb + 16 * TN_PITCH));
          s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfr[b] = __builtin_bit_cast(bf16x8_t, v);
        }
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          af[a] = *reinterpret_cast<const bf16x8_t*>(&ab[(wm * 64 + a * 16 + p) * PITCH + ks * 32 + grp * 8]);
#pragma unroll
        for (int b = 0; b < 4; ++b)
          bfr[b] = *reinterpret_cast<const bf16x8_t*>(&bb[(wn * 64 + b * 16 + p) * PITCH + ks * 32 + grp * 8]);
      }
      // MFMA-A = B-matrix columns (output col), MFMA-B = A-matrix columns (output row):
      // lane reg j = C[row = m0 + wm*64 + a*16 + (lane&15)][col = n0 + wn*64 + b*16 + (lane>>4)*4 + j]
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = COOT_MFMA_16x16x32(bfr[b], af[a], acc[a][b]);
    }
    if (st + 1 < nsteps) { if (do_cs) cs_acc(); sstore(buf ^ 1); }
    __syncthreads();
  }
  if (do_cs) {  // 16 row groups x 16 column chunks -> 128 column sums through LDS (the operand buffers are dead)
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(tid >> 4) * 128 + (tid & 15) * 8 + j] = cs[j];
    __syncthreads();
    if (tid < 128 && m0 + tid < g.Mo) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v += red[r * 128 + tid];
      acc_add(g.a_colsum + m0 + tid, v);
    }
    __syncthreads();
  }
  if (ws == nullptr && !direct) {  // one-pass path: fp32 atomics straight into C
    float* C = g.C + z * g.zC;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = m0 + wm * 64 + a * 16 + (lane & 15);
      if (row >= g.Mo) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = n0 + wn * 64 + b * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col + j < g.No) acc_add(C + (long)row * g.ldc + col + j, acc[a][b][j] * g.alpha);
      }
    }
    return;
  }
  // two-pass path: partial tile -> LDS -> coalesced 16-byte stores into ws[split][group][Mo][No]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int r = wm * 64 + a * 16 + (lane & 15), c = wn * 64 + b * 16 + (lane >> 4) * 4;
      *reinterpret_cast<f32x4_t*>(&Cs[r * CPITCH + c]) = acc[a][b];
    }
  __syncthreads();
  const int c4 = (tid & 31) * 4, rg = tid >> 5;  // 32 lanes x float4 = one 128-column row
  if (direct) {  // a single split: this workgroup owns its output tile, C += alpha * tile without a second launch
    float* C = g.C + z * g.zC;
    if (n0 + c4 < g.No) {
#pragma unroll 4
      for (int i = 0; i < TN_BC / 8; ++i) {
        const int rl = rg + 8 * i, row = m0 + rl;
        if (row < g.Mo) {
          f32x4_t* cp = reinterpret_cast<f32x4_t*>(C + (long)row * g.ldc + n0 + c4);
          const f32x4_t t = *reinterpret_cast<const f32x4_t*>(&Cs[rl * CPITCH + c4]);
          f32x4_t o = g.overwrite ? f32x4_t{0.f, 0.f, 0.f, 0.f} : *cp;
          o[0] += t[0] * g.alpha; o[1] += t[1] * g.alpha; o[2] += t[2] * g.alpha; o[3] += t[3] * g.alpha;
          *cp = o;
        }
      }
    }
    return;
  }
  float* wsp = ws + ((long)split * g.groups + z) * g.Mo * g.No;
  if (n0 + c4 < g.No) {
#pragma unroll 4
    for (int i = 0; i < TN_BC / 8; ++i) {
      const int rl = rg + 8 * i, row = m0 + rl;
      if (row < g.Mo) *reinterpret_cast<f32x4_t*>(wsp + (long)row * g.No + n0 + c4) = *reinterpret_cast<const f32x4_t*>(&Cs[rl * CPITCH + c4]);
    }
  }
}

// ---- wide variant: output tile 384 (m) x 128 (n), 512 threads -------------------------------------------------------
// The 128 x 128 tiles above re-read both operand slabs once per output tile (64 FLOP per byte from L2/HBM): the batched
// weight-gradient launch of one backward pass moved 1.7 GB at 5 TB/s.  With d_model = 384 every dY / X operand of the
// encoder is 384 (or 3 x 384) columns wide: a 384 x 128 tile reads the whole A slab once per n-tile and each B column
// block exactly once (98 FLOP per byte, a third less traffic).  8 waves = 4 (m, 96 rows) x 2 (n, 64 columns), 96
// accumulator registers, partial tiles stored straight from registers (a lane owns 4 consecutive n of one m: 16-byte stores).
constexpr int TNW_BM = 384, TNW_BN = 128;
constexpr int TNW_PA = TNW_BM + 16;  // 800-byte rows (== 8 dwords mod 64, like TN_PITCH: tr reads conflict free)
constexpr int TNW_PB = TNW_BN + 16;

__device__ __forceinline__ void gemm_tn_wide_body(const GemmTN& g, int bx, int by, int bz, int t_per_split, float* ws, int direct) {
  constexpr int ASZ = TN_BT * TNW_PA, BSZ = TN_BT * TNW_PB;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * ASZ + 2 * BSZ];
  bf16_t* As = smem;
  bf16_t* Bs = smem + 2 * ASZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = bz % g.groups, split = bz / g.groups;
  const int m0 = by * TNW_BM, n0 = bx * TNW_BN;
  const int t_begin = split * t_per_split;
  const int t_end = min(g.T, t_begin + t_per_split);
  const bf16_t* A = g.A + z * g.zA;
  const bf16_t* B = g.B + z * g.zB;
  const bool do_cs = g.a_colsum != nullptr && bx == 0;

  u32x4_t ar[6], br[2];
  int arow[6], acol[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { const int c = tid + 512 * i; arow[i] = c / 48; acol[i] = (c % 48) * 8; }
  const int brow = tid >> 4, bcol = (tid & 15) * 8;
  auto gload = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int t = t0 + arow[i];
      ar[i] = load16_guard(A + (long)t * g.lda + m0 + acol[i], t < t_end && m0 + acol[i] < g.Mo);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = t0 + brow + 32 * i;
      br[i] = load16_guard(B + (long)t * g.ldb + n0 + bcol, t < t_end && n0 + bcol < g.No);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<u32x4_t*>(&As[buf * ASZ + arow[i] * TNW_PA + acol[i]]) = ar[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4_t*>(&Bs[buf * BSZ + (brow + 32 * i) * TNW_PB + bcol]) = br[i];
  };
  // bias gradient: column sums of the A slab; a thread meets the column chunks (tid % 48 + 32 i) % 48, i.e. 3 distinct ones
  float cs[3][8];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[q][j] = 0.f;
  auto cs_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float* c8 = cs[i % 3];
      c8[0] += bflo(ar[i][0]); c8[1] += bfhi(ar[i][0]); c8[2] += bflo(ar[i][1]); c8[3] += bfhi(ar[i][1]);
      c8[4] += bflo(ar[i][2]); c8[5] += bfhi(ar[i][2]); c8[6] += bflo(ar[i][3]); c8[7] += bfhi(ar[i][3]);
    }
  };

  f32x4_t acc[6][4];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nsteps = t_begin < t_end ? (t_end - t_begin + TN_BT - 1) / TN_BT : 0;
  if (nsteps > 0) {
    gload(t_begin);
    if (do_cs) cs_acc();
    sstore(0);
  }
  __syncthreads();
  const int grp = lane >> 4, p = lane & 15;
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const bool prof = g.stamps != nullptr && bx == 0 && by == 0 && bz == 0;
  unsigned long long pt[4] = {0ull, 0ull, 0ull, 0ull}, tp = 0ull, tstart = 0ull;
#define TNW_PT(k) do { if (prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long tn = __builtin_amdgcn_s_memtime(); pt[k] += tn - tp; tp = tn; } } while (0)
  if (prof) { tstart = tp = __builtin_amdgcn_s_memtime(); }
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) gload(t_begin + (st + 1) * TN_BT);
    TNW_PT(0);
    const bf16_t* ab = As + buf * ASZ;
    const bf16_t* bb = Bs + buf * BSZ;
#pragma unroll
    for (int ks = 0; ks < TN_BT / 32; ++ks) {
      // k-slot (grp*8 + j) <-> t = ks*32 + 4*grp + j, k-slot (grp*8 + 4 + j) <-> t = ks*32 + 16 + 4*grp + j (same map for both operands)
      const int trow = ks * 32 + 4 * grp + (p >> 2);
      const int tcol = (p & 3) * 4;
      bf16x8_t bfr[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bf16_t* pb = &bb[trow * TNW_PB + wn * 64 + b * 16 + tcol];
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb + 16 * TNW_PB));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfr[b] = __builtin_bit_cast(bf16x8_t, v);
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const bf16_t* pa = &ab[trow * TNW_PA + wm * 96 + a * 16 + tcol];
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pa));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pa + 16 * TNW_PA));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, v);
        // lane reg j = C[row = m0 + wm*96 + a*16 + (lane&15)][col = n0 + wn*64 + b*16 + (lane>>4)*4 + j]
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = COOT_MFMA_16x16x32(bfr[b], af, acc[a][b]);
      }
    }
    TNW_PT(1);
    if (st + 1 < nsteps) { if (do_cs) cs_acc(); sstore(buf ^ 1); }
    TNW_PT(2);
    __syncthreads();
    TNW_PT(3);
  }
  if (prof && tid == 0) {
    g.stamps[0] = (unsigned long long)nsteps;
    for (int k = 0; k < 4; ++k) g.stamps[1 + k] = pt[k];
    g.stamps[5] = tp - tstart;
  }
#undef TNW_PT
  if (do_cs) {
    // Per-thread partial column sums -> LDS rows -> 384 threads add them up.  (The first version used LDS float atomics:
    // 24 ds_add_f32 wave-instructions that serialise per lane, ~1.5k cycles each — the workgroups carrying a bias gradient
    // ran 40 us longer than the others and set the duration of the whole launch; the global networks' weight-gradient
    // launch, a few k-steps per workgroup, took 43 us because of them.)
    // For iteration q the threads t = ch + 48 k (k <= 10) hold column chunk (ch + 32 q) % 48: rows k of red[11][384].
    float* red = reinterpret_cast<float*>(smem);  // the operand buffers are free: every wave passed the last barrier of the k loop
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      __syncthreads();
      float* rw = red + (tid / 48) * TNW_BM + acol[q];
      *reinterpret_cast<f32x4_t*>(rw) = f32x4_t{cs[q][0], cs[q][1], cs[q][2], cs[q][3]};
      *reinterpret_cast<f32x4_t*>(rw + 4) = f32x4_t{cs[q][4], cs[q][5], cs[q][6], cs[q][7]};
      __syncthreads();
      if (tid < TNW_BM) {
        const int ch = (tid / 8 + 96 - 32 * q) % 48;   // threads with t % 48 == ch hold this column in iteration q
        const int nk = (512 - ch + 47) / 48;
        for (int kk = 0; kk < nk; ++kk) tot += red[kk * TNW_BM + tid];
      }
    }
    if (tid < TNW_BM && m0 + tid < g.Mo) acc_add(g.a_colsum + m0 + tid, tot);
  }
  // partial tile straight from the accumulators: direct (single split) C += alpha * tile, else ws[split][z][Mo][No]
  float* dst = direct ? g.C + z * g.zC : ws + ((long)split * g.groups + z) * g.Mo * g.No;
  const long ldd = direct ? g.ldc : g.No;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int row = m0 + wm * 96 + a * 16 + (lane & 15);
    if (row >= g.Mo) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = n0 + wn * 64 + b * 16 + (lane >> 4) * 4;
      if (col >= g.No) continue;  // No % 4 == 0 (launcher): all four inside or all outside
      f32x4_t* dp = reinterpret_cast<f32x4_t*>(dst + (long)row * ldd + col);
      if (direct) {
        f32x4_t o = g.overwrite ? f32x4_t{0.f, 0.f, 0.f, 0.f} : *dp;
        o[0] += acc[a][b][0] * g.alpha; o[1] += acc[a][b][1] * g.alpha; o[2] += acc[a][b][2] * g.alpha; o[3] += acc[a][b][3] * g.alpha;
        *dp = o;
      } else {
        *dp = acc[a][b];
      }
    }
  }
}

// ---- wide variant fed by LDS-DMA ------------------------------------------------------------------------------------
// Phase stamps of the register-staged wide kernel (tools/tn_probe.py, r02d): a 64-row k-step takes ~4200 shader clocks of
// which the 48 MFMAs of a wave are 770: issuing the 8 operand loads of a thread blocks every wave for ~1100 clocks (64 KB
// through the CU's 64 B/clk vector-memory path, nothing else scheduled meanwhile), LDS reads + MFMA 1200, wait for the loads +
// ds_write 680, barrier 1150 — four serial phases.  Here the operands go global -> LDS by DMA (global_load_lds_dwordx4: no
// staging registers, no ds_write pass) into a ring of five 32-row stages with four stages in flight across the barriers
// (counted s_waitcnt vmcnt, raw s_barrier), so the loads of k-steps s+1 .. s+3 fly under the MFMAs of k-step s.  12 waves: 8
// MFMA waves that never touch vector memory inside the k loop + 4 loader waves (one per SIMD) that do nothing else.
//   * DMA writes LDS lane-linear (wave-uniform base + 16 B x lane), so rows cannot be padded: the stage holds unpadded rows
//     (768 B of dY, 256 B of X) whose 32-byte granules are XOR-swizzled with (row & 7) — applied to the SOURCE address of the
//     DMA and to the ds_read_b64_tr_b16 address; the 8 rows x 32 B of a half-wave's transposing read then cover all 64 banks.
//   * the DMA statements are inline asm (M0 = LDS destination is compiler-reserved and must be written in the statement that
//     uses it): hipcc does not count them, every wait on them is written by hand.  No other vector-memory operation lives in
//     the k loop.  Rows past the split's end / columns past No come from a 16-byte zero page.
//   * bias-gradient column sums (a_colsum) are taken by the loader waves from the landed stages.
constexpr int TND_ROWS = 32, TND_NST = 5;  // stages in the ring: NST - 1 in flight, 5 x 32 KB = the whole LDS
constexpr int TND_A_BYTES = TND_ROWS * TNW_BM * 2, TND_B_BYTES = TND_ROWS * TNW_BN * 2, TND_STAGE = TND_A_BYTES + TND_B_BYTES;
__device__ __attribute__((aligned(16))) unsigned int g_tn_zero16[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void gemm_tn_dma_body(const GemmTN& g, int bx, int by, int bz, int t_per_split, float* ws, int direct) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[TND_NST * TND_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = bz % g.groups, split = bz / g.groups;
  const int m0 = by * TNW_BM, n0 = bx * TNW_BN;
  const int t_begin = split * t_per_split;
  const int t_end = min(g.T, t_begin + t_per_split);
  const int nst = t_begin < t_end ? (t_end - t_begin + TND_ROWS - 1) / TND_ROWS : 0;

  if (wave >= 8) {
    // ---- loader waves (one per SIMD): all DMA traffic of the workgroup.  A wave that issues vector-memory instructions
    // faster than the CU's 64 B/clk path takes them stalls at the issue; with the loads on the MFMA waves (first version:
    // 4 DMAs per wave per stage) that stall was 660 clocks of every 2100-clock k-step.  Here only these four waves stall.
    const int pw = wave - 8;
    const bf16_t* A = g.A + z * g.zA;
    const bf16_t* B = g.B + z * g.zB;
    const unsigned lds0 = (unsigned)(unsigned long)((unsigned char __attribute__((address_space(3)))*)smem);
    // slots (1 KB of a stage each) pw + 4 i: i < 6 -> dY rows (slots 0 .. 23), i = 6, 7 -> X rows (slots 24 .. 31): the LDS
    // destinations of a wave's eight DMAs are 4 KB apart.  Source = scalar base of the stage + a per-lane byte offset that
    // never changes (8 VGPRs); columns past No (last column tile of a 192-wide problem) re-read column 0 of the tile: they
    // only reach accumulator columns the epilogue does not store.
    unsigned off[8]; int srow[8];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int o = (pw + 4 * i) * 1024 + lane * 16, r = o / (TNW_BM * 2), w = o - r * (TNW_BM * 2);
      const int col = (((w >> 5) ^ (r & 7)) << 4) + ((w >> 4) & 1) * 8;
      off[i] = (unsigned)(r * (int)g.lda + col) * 2u; srow[i] = r;  // Mo % 384 == 0: every column exists
    }
#pragma unroll
    for (int i = 6; i < 8; ++i) {
      const int o = (pw + 4 * (i - 6)) * 1024 + lane * 16, r = o >> 8, w = o & 255;
      const int col = (((w >> 5) ^ (r & 7)) << 4) + ((w >> 4) & 1) * 8;
      off[i] = (unsigned)(r * (int)g.ldb + (n0 + col < g.No ? col : 0)) * 2u; srow[i] = r;
    }
    const char* ap = reinterpret_cast<const char*>(A + (long)t_begin * g.lda + m0);
    const char* bp = reinterpret_cast<const char*>(B + (long)t_begin * g.ldb + n0);
    const long stepA = (long)TND_ROWS * g.lda * 2, stepB = (long)TND_ROWS * g.ldb * 2;
    int ibuf = 0;
    auto issue = [&](int s) {  // called with s = 0, 1, 2, ... in order
      const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)ibuf * TND_STAGE + (unsigned)pw * 1024u);
      ibuf = ibuf + 1 == TND_NST ? 0 : ibuf + 1;
      const int t0 = t_begin + s * TND_ROWS;
      if (t0 + TND_ROWS <= t_end) {  // every row of the stage exists: one statement, M0 stepped by 4 KB between the DMAs
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[o0], %[sa]\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[o1], %[sa]\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[o2], %[sa]\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[o3], %[sa]\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[o4], %[sa]\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[o5], %[sa]\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[o6], %[sb]\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[o7], %[sb]\n\ts_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]), [o3] "v"(off[3]), [o4] "v"(off[4]), [o5] "v"(off[5]),
              [o6] "v"(off[6]), [o7] "v"(off[7]), [sa] "s"(ap), [sb] "s"(bp), [dst] "s"(dst0)
            : "memory", "scc");
      } else {  // the split's last, partial stage: rows past its end come from the zero page
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const void* p = t0 + srow[i] < t_end ? static_cast<const void*>((i < 6 ? ap : bp) + off[i]) : static_cast<const void*>(g_tn_zero16);
          glds16(p, __builtin_amdgcn_readfirstlane(dst0 + (unsigned)i * 4096u));
        }
      }
      ap += stepA; bp += stepB;
    };
    // bias gradient (a_colsum: column sums of the dY slab, tiles bx == 0 only): taken by the loader waves from the landed
    // stages — lane id < 192 sums 16-byte chunk id % 48 over rows 8 (id / 48) .. + 7 of every stage, fp32 atomics at the end
    const bool do_cs = g.a_colsum != nullptr && bx == 0;
    const int cid = pw * 64 + lane, cc = cid % 48, crg = cid / 48;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int lbuf = 0;
    const bool aprof = g.stamps != nullptr && bx == 0 && by == 0 && bz == 0;
    const bool lprof = aprof && pw == 0;
    unsigned long long lt[3] = {0ull, 0ull, 0ull}, lp = 0ull;
#define TND_LT(k) do { if (lprof) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); lt[k] += tn - lp; lp = tn; } } while (0)
    static_assert(TND_NST == 5, "the counted waits below are written for a ring of 5 stages");
    // Barrier s (s = -1 .. nst - 1) certifies stage s + 1: the MFMA waves read the first fragments of a stage one barrier
    // before they multiply them (no LDS latency in front of the first MFMA of a stage).  This wave passes barrier s with
    // stages s + 2, s + 3 in flight and then refills the buffer of stage s - 1 (read out: every wave is past barrier s).
    for (int s = 0; s < TND_NST - 1 && s < nst; ++s) issue(s);
    if (lprof) lp = __builtin_amdgcn_s_memtime();
    for (int s = -1; s < nst; ++s) {
      if (s + 3 < nst) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TND_LT(0);
      asm volatile("s_barrier" ::: "memory");
      TND_LT(1);
      if (s < 0) continue;
      if (s + TND_NST - 1 < nst) issue(s + TND_NST - 1);
      TND_LT(2);
      if (do_cs) {  // stage s is complete (barrier s - 1) and stays until some wave passes barrier s + 1
        if (cid < 192) {
          const unsigned char* st = smem + lbuf * TND_STAGE;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int r = crg * 8 + k;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(st + r * (TNW_BM * 2) + (((cc >> 1) ^ (r & 7)) << 5) + (cc & 1) * 16);
            cs[0] += bflo(v[0]); cs[1] += bfhi(v[0]); cs[2] += bflo(v[1]); cs[3] += bfhi(v[1]);
            cs[4] += bflo(v[2]); cs[5] += bfhi(v[2]); cs[6] += bflo(v[3]); cs[7] += bfhi(v[3]);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      lbuf = lbuf + 1 == TND_NST ? 0 : lbuf + 1;
    }
    if (do_cs && cid < 192) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc_add(g.a_colsum + m0 + cc * 8 + j, cs[j]);
    }
    if (lprof && lane == 0) for (int k = 0; k < 3; ++k) g.stamps[8 + k] = lt[k];  // loader wave 0: landing wait, barrier, DMA issue
#undef TND_LT
    return;
  }

  // ---- MFMA waves: 4 (m, 96 rows) x 2 (n, 64 columns) as in the register-staged kernel -----------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  f32x4_t acc[6][4];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a stage (the same k-slot <-> token row map as the register-staged kernel, both operands)
  const int grp = lane >> 4, p = lane & 15;
  const int trow = 4 * grp + (p >> 2), x7 = trow & 7, tcb = (p & 3) * 8;
  int aoff[6], boff[4];
#pragma unroll
  for (int a = 0; a < 6; ++a) aoff[a] = trow * (TNW_BM * 2) + (((wm * 6 + a) ^ x7) << 5) + tcb;
#pragma unroll
  for (int b = 0; b < 4; ++b) boff[b] = TND_A_BYTES + trow * (TNW_BN * 2) + (((wn * 4 + b) ^ x7) << 5) + tcb;
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_p;

  const bool aprof = g.stamps != nullptr && bx == 0 && by == 0 && bz == 0;
  const bool prof = aprof && wave == 0;
  // per-workgroup span (100 MHz real-time counter) at stamps[64 + 2 * block]: launch-wide load balance (tools/tn_probe.py)
  if (g.stamps && tid == 0) g.stamps[64 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  unsigned long long pt[4] = {0ull, 0ull, 0ull, 0ull}, tp = 0ull, tstart = 0ull;
#define TND_PT(k) do { if (prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long tn = __builtin_amdgcn_s_memtime(); pt[k] += tn - tp; tp = tn; } } while (0)
  if (prof) { tstart = tp = __builtin_amdgcn_s_memtime(); }
  // One stage: 24 MFMAs on fragments of `st` — the B fragments and the first two A fragments were read before the barrier in
  // front of it (bc, ac) — and, half way, the same first reads of the next stage `stn` (certified by that barrier) into bn, an.
  auto read_a = [&](const unsigned char* st, int a) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(st + aoff[a]));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(st + aoff[a] + 16 * TNW_BM * 2));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto read_b = [&](const unsigned char* st, int b) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(st + boff[b]));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(st + boff[b] + 16 * TNW_BN * 2));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto preload = [&](const unsigned char* st, bf16x8_t (&bn)[4], bf16x8_t (&an)[2]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) bn[b] = read_b(st, b);
    an[0] = read_a(st, 0); an[1] = read_a(st, 1);
  };
  auto stage = [&](const unsigned char* st, const unsigned char* stn, const bf16x8_t (&bc)[4], const bf16x8_t (&ac)[2],
                   bf16x8_t (&bn)[4], bf16x8_t (&an)[2]) {
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const bf16x8_t af = a < 2 ? ac[a] : read_a(st, a);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = COOT_MFMA_16x16x32(bc[b], af, acc[a][b]);
      if (a == 3) preload(stn, bn, an);  // (past the last stage: a buffer nobody needs — harmless)
    }
  };
  auto stage_ptr = [&](int k) { return smem + k * TND_STAGE; };
  auto next = [](int k) { return k + 1 == TND_NST ? 0 : k + 1; };
  bf16x8_t b0[4], a0[2], b1[4], a1[2];
  asm volatile("s_barrier" ::: "memory");  // barrier -1: stage 0 has landed
  if (nst > 0) preload(smem, b0, a0);
  int k0 = 0;
  int s = 0;
  for (; s + 1 < nst; s += 2) {
    const int k1 = next(k0), k2 = next(k1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TND_PT(1);
    stage(stage_ptr(k0), stage_ptr(k1), b0, a0, b1, a1);
    TND_PT(3);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TND_PT(1);
    stage(stage_ptr(k1), stage_ptr(k2), b1, a1, b0, a0);
    TND_PT(3);
    k0 = k2;
  }
  if (s < nst) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TND_PT(1);
    stage(stage_ptr(k0), stage_ptr(next(k0)), b0, a0, b1, a1);
    TND_PT(3);
  }
  if (g.stamps && tid == 0) g.stamps[65 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  if (prof && tid == 0) {
    g.stamps[0] = (unsigned long long)nst;
    for (int k = 0; k < 4; ++k) g.stamps[1 + k] = pt[k];
    g.stamps[5] = tp - tstart;
  }
#undef TND_PT
  // partial tile straight from the accumulators: direct (single split) C += alpha * tile, else ws[split][z][Mo][No]
  float* dst = direct ? g.C + z * g.zC : ws + ((long)split * g.groups + z) * g.Mo * g.No;
  const long ldd = direct ? g.ldc : g.No;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const int row = m0 + wm * 96 + a * 16 + (lane & 15);
    if (row >= g.Mo) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = n0 + wn * 64 + b * 16 + (lane >> 4) * 4;
      if (col >= g.No) continue;
      f32x4_t* dp = reinterpret_cast<f32x4_t*>(dst + (long)row * ldd + col);
      if (direct) {
        f32x4_t o = g.overwrite ? f32x4_t{0.f, 0.f, 0.f, 0.f} : *dp;
        o[0] += acc[a][b][0] * g.alpha; o[1] += acc[a][b][1] * g.alpha; o[2] += acc[a][b][2] * g.alpha; o[3] += acc[a][b][3] * g.alpha;
        *dp = o;
      } else {
        *dp = acc[a][b];
      }
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN g, int t_per_split, int splits, float* ws, int direct) {
  gemm_tn_body<MODE>(g, blockIdx.x, blockIdx.y, blockIdx.z, t_per_split, splits, ws, direct);
}

// Several weight-gradient problems in ONE launch (every dW of a network's backward pass): with all of them in flight each
// problem needs only a few splits over the token axis (4 instead of 32: an eighth of the fp32 partial-tile traffic, 8x
// longer pipelined loops per workgroup), and the global networks' eight tiny problems cost one launch instead of sixteen.
constexpr int TN_MAX_ITEMS = 10;
constexpr int TN_MAX_SPLITS = 8;  // token splits of one weight-gradient problem: the clamp in tn_batch_flush and the reduce kernel's partial loads
struct TnItem { GemmTN g; int blk0, gx, gy, t_per_split, splits, direct; long ws_off; };
struct TnBatch { int n; TnItem it[TN_MAX_ITEMS]; };

template <int MODE>
__global__ __launch_bounds__(256) void gemm_tn_batch_kernel(TnBatch b, float* ws_base) {
  int i = 0;
  for (int t = 1; t < b.n; ++t) if ((int)blockIdx.x >= b.it[t].blk0) i = t;
  const TnItem& it = b.it[i];
  const int local = blockIdx.x - it.blk0;
  const int bx = local % it.gx, by = (local / it.gx) % it.gy, bz = local / (it.gx * it.gy);
  gemm_tn_body<MODE>(it.g, bx, by, bz, it.t_per_split, it.splits, it.direct ? nullptr : ws_base + it.ws_off, it.direct);
}

__global__ __launch_bounds__(512) void gemm_tn_wide_batch_kernel(TnBatch b, float* ws_base) {
  int i = 0;
  for (int t = 1; t < b.n; ++t) if ((int)blockIdx.x >= b.it[t].blk0) i = t;
  const TnItem& it = b.it[i];
  const int local = blockIdx.x - it.blk0;
  // (an XCD-aware order — the gx column blocks of one dY slab on one XCD — was measured 6 % SLOWER on the whole step: the
  // blocks of a slab then start together on one L2 and queue on the same channels; id = bx + gx * (...) spreads them)
  const int bx = local % it.gx, by = (local / it.gx) % it.gy, bz = local / (it.gx * it.gy);
  gemm_tn_wide_body(it.g, bx, by, bz, it.t_per_split, it.direct ? nullptr : ws_base + it.ws_off, it.direct);
}

// The same launch with an explicit workgroup -> (problem, tile) table built on the host (tn_batch_flush): the workgroups that
// read the SAME token rows — all (m, n) tiles of one problem and one split — sit on ONE XCD (block b runs on XCD b % 8), so a
// 384-column dY slab / 128-column X block is fetched over the fabric once per split instead of once per tile that uses it.
// Without it the launch is bound by the fabric: FETCH_SIZE = the algorithmic re-read count (r02c: 0.99 GB of 1.15 GB per
// video-side launch, 5.4 TB/s).
// The table is a list of pieces (runs of consecutive tiles of one problem) ordered by XCD; kept small (~400 B of kernel
// arguments: a 1.5 KB per-workgroup table next to the 1.8 KB problem list crashed hipGraphLaunch on replay).
constexpr int TN_MAP_PIECES = 96;
struct TnPiece { unsigned short first; unsigned char item, count; };
struct TnMap { unsigned char xstart[12]; TnPiece pc[TN_MAP_PIECES]; };  // pieces [xstart[x], xstart[x + 1]) belong to XCD x
__host__ __device__ __forceinline__ bool tn_map_find(const TnMap& m, int block, int& item, int& local) {
  const int x = block & 7;
  int j = block >> 3;
  for (int k = m.xstart[x]; k < m.xstart[x + 1]; ++k) {
    const int c = m.pc[k].count;
    if (j < c) { item = m.pc[k].item; local = m.pc[k].first + j; return true; }
    j -= c;
  }
  return false;
}
__device__ __forceinline__ bool tn_map_lookup(const TnMap& m, int& item, int& local) { return tn_map_find(m, (int)blockIdx.x, item, local); }
__global__ __launch_bounds__(512) void gemm_tn_wide_mapped_kernel(TnBatch b, TnMap m, float* ws_base) {
  int i, local;
  if (!tn_map_lookup(m, i, local)) return;
  const TnItem& it = b.it[i];
  const int bx = local % it.gx, by = (local / it.gx) % it.gy, bz = local / (it.gx * it.gy);
  gemm_tn_wide_body(it.g, bx, by, bz, it.t_per_split, it.direct ? nullptr : ws_base + it.ws_off, it.direct);
}

__global__ __launch_bounds__(768) void gemm_tn_dma_kernel(TnBatch b, TnMap m, float* ws_base, int use_map) {
  int i, local;
  if (use_map) {
    if (!tn_map_lookup(m, i, local)) return;
  } else {
    i = 0;
    for (int t = 1; t < b.n; ++t) if ((int)blockIdx.x >= b.it[t].blk0) i = t;
    local = blockIdx.x - b.it[i].blk0;
  }
  const TnItem& it = b.it[i];
  const int bx = local % it.gx, by = (local / it.gx) % it.gy, bz = local / (it.gx * it.gy);
  gemm_tn_dma_body(it.g, bx, by, bz, it.t_per_split, it.direct ? nullptr : ws_base + it.ws_off, it.direct);
}

// grid rows [0, b.n): the problems of the batch; rows b.n .. : the deferred column sums that ride in this launch (da.nseg of them)
__global__ __launch_bounds__(256) void gemm_tn_batch_reduce_kernel(TnBatch b, const float* ws_base, DeferArgs da) {
  if ((int)blockIdx.y >= b.n) { colsum_defer_block(da, (int)blockIdx.y - b.n, (int)blockIdx.x); return; }
  const TnItem& it = b.it[blockIdx.y];
  if (it.direct) return;
  const GemmTN& g = it.g;
  const float* ws = ws_base + it.ws_off;
  const long per = (long)g.Mo * g.No;
  const long total4 = per * g.groups / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long e = i * 4;
    const int z = (int)(e / per);
    const long r = e % per;
    const int m = (int)(r / g.No), n = (int)(r % g.No);
    // all partials of the element first (<= 8 splits, tn_batch_flush; a split past the last re-reads it), then the sum in split order:
    // one load at a time per thread left this launch at a third of the memory system's rate
    f32x4_t part[TN_MAX_SPLITS];
#pragma unroll
    for (int sp = 0; sp < TN_MAX_SPLITS; ++sp) part[sp] = *reinterpret_cast<const f32x4_t*>(ws + ((long)min(sp, it.splits - 1) * g.groups + z) * per + r);
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < TN_MAX_SPLITS; ++sp) if (sp < it.splits) s += part[sp];
    float* c = g.C + z * g.zC + (long)m * g.ldc + n;
    if (g.overwrite) { c[0] = s[0] * g.alpha; c[1] = s[1] * g.alpha; c[2] = s[2] * g.alpha; c[3] = s[3] * g.alpha; }
    else { c[0] += s[0] * g.alpha; c[1] += s[1] * g.alpha; c[2] += s[2] * g.alpha; c[3] += s[3] * g.alpha; }
  }
}

// C[z][m][n] += alpha * sum_s ws[s][z][m][n]
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(GemmTN g, int splits, const float* ws) {
  const long per = (long)g.Mo * g.No;
  const long total4 = per * g.groups / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long e = i * 4;
    const int z = (int)(e / per);
    const long r = e % per;
    const int m = (int)(r / g.No), n = (int)(r % g.No);
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < splits; ++sp) s += *reinterpret_cast<const f32x4_t*>(ws + ((long)sp * g.groups + z) * per + r);
    float* c = g.C + z * g.zC + (long)m * g.ldc + n;
    if (g.overwrite) { c[0] = s[0] * g.alpha; c[1] = s[1] * g.alpha; c[2] = s[2] * g.alpha; c[3] = s[3] * g.alpha; }
    else { c[0] += s[0] * g.alpha; c[1] += s[1] * g.alpha; c[2] += s[2] * g.alpha; c[3] += s[3] * g.alpha; }
  }
}

static thread_local float* g_tn_ws = nullptr;
static thread_local size_t g_tn_ws_floats = 0;
void set_tn_default_workspace(float* ws, size_t floats) { g_tn_ws = ws; g_tn_ws_floats = floats; }
void get_tn_default_workspace(float** ws, size_t* floats) { *ws = g_tn_ws; *floats = g_tn_ws_floats; }
static int g_tn_mode = 0;
void set_tn_mode(int mode) { g_tn_mode = mode ? 1 : 0; }
int get_tn_mode() { return g_tn_mode; }

static int tn_splits(int T, int Mo, int No, int groups) {
  const int tiles = ((Mo + TN_BC - 1) / TN_BC) * ((No + TN_BC - 1) / TN_BC) * groups;
  // split the t reduction so that about one workgroup per CU exists; each split handles >= 256 rows
  int splits = (288 + tiles - 1) / tiles;
  const int max_splits = (T + 4 * TN_BT - 1) / (4 * TN_BT);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  return splits;
}

size_t gemm_tn_workspace_floats(int T, int Mo, int No, int groups) {
  return (size_t)tn_splits(T, Mo, No, groups) * groups * Mo * No;
}

static thread_local bool g_tn_collect = false;
static thread_local int g_tn_nitems = 0;
static thread_local GemmTN g_tn_items[TN_MAX_ITEMS];

static thread_local long g_tn_ws_used = 0;  // floats of the default workspace taken by the flushes of the current scope

void tn_batch_begin() { g_tn_collect = true; g_tn_nitems = 0; g_tn_ws_used = 0; }

static int g_tn_target_wgs = 256;  // wide tiles: workgroups per launch the split count aims at (one per CU)
void set_tn_target_wgs(int n) { g_tn_target_wgs = n > 0 ? n : 256; }
static int g_tn_dma = 1;
void set_tn_dma(int on) { g_tn_dma = on; }
int get_tn_dma() { return g_tn_dma; }
void set_xcd_order(int on) { g_xcd_order = on; }
int get_xcd_order() { return g_xcd_order; }

// Workgroup table of gemm_tn_wide_mapped_kernel / gemm_tn_dma_kernel: sharing groups (one problem, one split, one batch group:
// gx * gy tiles, cut into pieces of <= 16) go to the least loaded XCD, largest first; slot j of XCD x is block 8 j + x.
// Returns the grid size, 0 when the launch does not fit the table (the linear order is used).
static int tn_xcd_map(const TnBatch& b, TnMap& map) {
  struct Piece { int item, first, count, xcd; };
  Piece pc[TN_MAP_PIECES];
  int np = 0;
  for (int i = 0; i < b.n; ++i) {
    const TnItem& it = b.it[i];
    const int per = it.gx * it.gy, nb = it.g.groups * it.splits;
    for (int bz = 0; bz < nb; ++bz)
      for (int o = 0; o < per; o += 16) {
        if (np == TN_MAP_PIECES || bz * per + o > 0xFFFF) return 0;
        pc[np++] = Piece{i, bz * per + o, per - o < 16 ? per - o : 16, 0};
      }
  }
  std::stable_sort(pc, pc + np, [](const Piece& a, const Piece& c) { return a.count > c.count; });
  int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < np; ++k) {
    int x = 0;
    for (int q = 1; q < 8; ++q) if (load[q] < load[x]) x = q;
    pc[k].xcd = x; load[x] += pc[k].count;
  }
  int n = 0, mx = 0;
  for (int x = 0; x < 8; ++x) {
    map.xstart[x] = (unsigned char)n;
    for (int k = 0; k < np; ++k)
      if (pc[k].xcd == x) map.pc[n++] = TnPiece{(unsigned short)pc[k].first, (unsigned char)pc[k].item, (unsigned char)pc[k].count};
    if (load[x] > mx) mx = load[x];
  }
  map.xstart[8] = (unsigned char)n;
  for (int k = n; k < TN_MAP_PIECES; ++k) map.pc[k] = TnPiece{0, 0, 0};
  return 8 * mx;
}

// host-side check of the table (tests): problems given as (gx, gy, groups, splits); out_item / out_local [max_blocks] receive what
// each block of the launch would work on (-1: an empty slot); returns the grid size, 0 = the launch does not fit the table
int tn_debug_xcd_map(int n, const int* gx, const int* gy, const int* groups, const int* splits, int* out_item, int* out_local, int max_blocks) {
  if (n < 1 || n > TN_MAX_ITEMS) return -1;
  TnBatch b; b.n = n;
  for (int i = 0; i < n; ++i) { b.it[i].gx = gx[i]; b.it[i].gy = gy[i]; b.it[i].g.groups = groups[i]; b.it[i].splits = splits[i]; }
  TnMap map = {};
  const int grid = tn_xcd_map(b, map);
  if (grid > max_blocks) return -2;
  for (int blk = 0; blk < grid; ++blk) {
    int item = -1, local = -1;
    if (!tn_map_find(map, blk, item, local)) { item = -1; local = -1; }
    out_item[blk] = item; out_local[blk] = local;
  }
  return grid;
}

int tn_batch_flush(hipStream_t stream, bool take_colsums) {
  const int n = g_tn_nitems;
  g_tn_nitems = 0;
  if (n == 0) return take_colsums ? colsum_defer_flush(stream) : 0;
  // two launches at most: problems whose m extent is a multiple of 384 take the wide 384 x 128 tiles, the rest 128 x 128
  TnBatch bw, bn; bw.n = 0; bn.n = 0;
  long tiles_w = 0, tiles_n = 0;
  bool wide[TN_MAX_ITEMS];
  for (int i = 0; i < n; ++i) {
    const GemmTN& g = g_tn_items[i];
    wide[i] = g_tn_mode == 0 && g.Mo % TNW_BM == 0 && g.No % 4 == 0 && g.ldc % 4 == 0 && g.zC % 4 == 0;
    if (wide[i]) tiles_w += (long)((g.No + TNW_BN - 1) / TNW_BN) * (g.Mo / TNW_BM) * g.groups;
    else tiles_n += (long)((g.No + TN_BC - 1) / TN_BC) * ((g.Mo + TN_BC - 1) / TN_BC) * g.groups;
  }
  // every flush of a scope takes its own part of the workspace: an earlier flush may still be running on the aux stream
  int blk_w = 0, blk_n = 0; long ws_off = g_tn_ws_used; bool ws_w = false, ws_n = false;
  for (int i = 0; i < n; ++i) {
    const bool w = wide[i];
    TnBatch& b = w ? bw : bn;
    TnItem& it = b.it[b.n++];
    it.g = g_tn_items[i];
    const GemmTN& g = it.g;
    it.gx = w ? (g.No + TNW_BN - 1) / TNW_BN : (g.No + TN_BC - 1) / TN_BC;
    it.gy = w ? g.Mo / TNW_BM : (g.Mo + TN_BC - 1) / TN_BC;
    // wide: one 512-thread workgroup per CU -> about 256 workgroups; narrow: two per CU.  Every split >= 512 token rows.
    const long tiles = w ? tiles_w : tiles_n;
    int splits = w ? (int)(g_tn_target_wgs / tiles) : (int)((512 + tiles - 1) / tiles);
    const int max_splits = g.T / 512 > 0 ? g.T / 512 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits > TN_MAX_SPLITS) splits = TN_MAX_SPLITS;
    if (splits < 1) splits = 1;
    int tps = (g.T + splits - 1) / splits;
    tps = (tps + TN_BT - 1) / TN_BT * TN_BT;
    splits = (g.T + tps - 1) / tps;
    COOT_REQUIRE(splits <= TN_MAX_SPLITS, "tn_batch_flush: %d splits (the reduce kernel loads %d partials)", splits, TN_MAX_SPLITS);
    it.t_per_split = tps; it.splits = splits;
    it.direct = (splits == 1 && g.ldc % 4 == 0 && g.No % 4 == 0 && g.zC % 4 == 0) ? 1 : 0;
    int& blk = w ? blk_w : blk_n;
    it.blk0 = blk; blk += it.gx * it.gy * g.groups * splits;
    it.ws_off = ws_off;
    if (!it.direct) { ws_off += (long)splits * g.groups * g.Mo * g.No; (w ? ws_w : ws_n) = true; }
  }
  float* ws = g_tn_ws;
  if ((ws_w || ws_n) && (!ws || (size_t)ws_off > g_tn_ws_floats)) {  // no room for the batch: problem by problem
    g_tn_collect = false;
    for (int i = 0; i < n; ++i) { GemmTN g = g_tn_items[i]; int rc = launch_gemm_tn(g, stream); if (rc) { g_tn_collect = true; return rc; } }
    g_tn_collect = true;
    return take_colsums ? colsum_defer_flush(stream) : 0;
  }
  g_tn_ws_used = ws_off;
  double flops = 0;
  for (int i = 0; i < n; ++i) flops += 2.0 * g_tn_items[i].T * g_tn_items[i].Mo * g_tn_items[i].No * g_tn_items[i].groups;
  void* ts = timing_begin(TIMING_TN, flops, 0, stream);
  if (bw.n) {
    TnMap map = {};
    const int grid = (g_xcd_order & 2) ? tn_xcd_map(bw, map) : 0;
    const bool dma = g_tn_dma != 0;
    if (dma) hipLaunchKernelGGL(gemm_tn_dma_kernel, dim3(grid > 0 ? grid : blk_w), dim3(768), 0, stream, bw, map, ws, grid > 0 ? 1 : 0);
    else if (grid > 0) hipLaunchKernelGGL(gemm_tn_wide_mapped_kernel, dim3(grid), dim3(512), 0, stream, bw, map, ws);
    else hipLaunchKernelGGL(gemm_tn_wide_batch_kernel, dim3(blk_w), dim3(512), 0, stream, bw, ws);
    COOT_CHECK_LAUNCH("gemm_tn_wide_batch");
  }
  if (bn.n) {
    if (g_tn_mode == 0) hipLaunchKernelGGL(gemm_tn_batch_kernel<0>, dim3(blk_n), dim3(256), 0, stream, bn, ws);
    else hipLaunchKernelGGL(gemm_tn_batch_kernel<1>, dim3(blk_n), dim3(256), 0, stream, bn, ws);
    COOT_CHECK_LAUNCH("gemm_tn_batch");
  }
  // the caller's deferred column sums (independent of the GEMMs) ride in the first reduce launch as extra grid rows: a launch of its
  // own costs ~5 us of the stream whatever it does (profiles/README.md round 4).  256 column blocks of 16 cover 4 096 columns.
  DeferArgs da; da.nseg = 0;
  bool ride = take_colsums && (ws_w || ws_n) && colsum_defer_take(&da, 256 * 16);
  if (ws_w) {
    hipLaunchKernelGGL(gemm_tn_batch_reduce_kernel, dim3(256, bw.n + (ride ? da.nseg : 0)), dim3(256), 0, stream, bw, (const float*)ws, da);
    COOT_CHECK_LAUNCH("gemm_tn_batch_reduce");
    if (ride) { ride = false; da.nseg = 0; }
  }
  if (ws_n) {
    hipLaunchKernelGGL(gemm_tn_batch_reduce_kernel, dim3(256, bn.n + (ride ? da.nseg : 0)), dim3(256), 0, stream, bn, (const float*)ws, da);
    COOT_CHECK_LAUNCH("gemm_tn_batch_reduce");
  }
  timing_end(ts, stream);
  if (take_colsums) return colsum_defer_flush(stream);  // whatever did not ride (no reduce launch, or a very wide sum): its own launch; else a no-op
  return 0;
}
void tn_batch_end() { g_tn_collect = false; g_tn_nitems = 0; }

// set_tn_force_overwrite(true): every problem launched / recorded by this thread WRITES its C (coot_net_bwd inside a step whose
// caller did not zero the weight-matrix gradients: each is produced by exactly one problem)
static thread_local bool g_tn_force_overwrite = false;
void set_tn_force_overwrite(bool on) { g_tn_force_overwrite = on; }
int launch_gemm_tn(const GemmTN& g_in, hipStream_t stream) {
  GemmTN g = g_in;
  if (g_tn_force_overwrite) g.overwrite = 1;
  COOT_REQUIRE(g.A && g.B && g.C, "gemm_tn: null operand");
  COOT_REQUIRE(g.lda % 8 == 0 && g.ldb % 8 == 0 && g.Mo % 8 == 0 && g.No % 8 == 0 && g.zA % 8 == 0 && g.zB % 8 == 0,
               "gemm_tn: lda/ldb/Mo/No must be multiples of 8 (Mo=%d No=%d lda=%ld ldb=%ld)", g.Mo, g.No, g.lda, g.ldb);
  if (g.Mo <= 0 || g.No <= 0) return 0;
  if (g.T <= 0) {  // an empty sum: nothing to add; "written" means zeros
    if (g.overwrite) for (int z = 0; z < g.groups; ++z) for (int m = 0; m < g.Mo; ++m) { int rc = launch_fill_f32(g.C + z * g.zC + (long)m * g.ldc, g.No, 0.f, stream); if (rc) return rc; }
    return 0;
  }
  COOT_REQUIRE(!g.a_colsum || g.groups == 1, "gemm_tn: a_colsum needs groups == 1");
  if (g_tn_collect && !g.ws) {  // inside tn_batch_begin() .. tn_batch_flush(): deferred, launched together
    if (g_tn_nitems == TN_MAX_ITEMS) { int rc = tn_batch_flush(stream); if (rc) return rc; }
    g_tn_items[g_tn_nitems++] = g;
    return 0;
  }
  if (g.overwrite) {  // the single-problem path may use the atomic mode: zero C here and accumulate
    COOT_REQUIRE(g.groups == 1 && g.ldc == g.No, "gemm_tn: overwrite needs a dense single-group C");
    int rc = launch_fill_f32(g.C, (long)g.Mo * g.No, 0.f, stream);
    if (rc) return rc;
    GemmTN g2 = g; g2.overwrite = 0;
    return launch_gemm_tn(g2, stream);
  }
  int splits = tn_splits(g.T, g.Mo, g.No, g.groups);
  int t_per_split = (g.T + splits - 1) / splits;
  t_per_split = (t_per_split + TN_BT - 1) / TN_BT * TN_BT;
  splits = (g.T + t_per_split - 1) / t_per_split;
  float* ws = nullptr;
  const size_t need = (size_t)splits * g.groups * g.Mo * g.No;
  if (g.ws && need <= g.ws_floats) ws = g.ws;
  else if (!g.ws && g_tn_ws && need <= g_tn_ws_floats) ws = g_tn_ws;
  dim3 grid((g.No + TN_BC - 1) / TN_BC, (g.Mo + TN_BC - 1) / TN_BC, g.groups * splits);
  // one split (global networks): the tile owner adds into C itself — no workspace, no reduce launch (needs 16-byte rows)
  const int direct = (splits == 1 && g.ldc % 4 == 0 && g.No % 4 == 0 && g.zC % 4 == 0) ? 1 : 0;
  if (direct) ws = nullptr;
  void* ts = timing_begin(TIMING_TN, 2.0 * g.T * g.Mo * g.No * g.groups, 0, stream);
  if (g_tn_mode == 0)
    hipLaunchKernelGGL(gemm_tn_kernel<0>, grid, dim3(256), 0, stream, g, t_per_split, splits, ws, direct);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, dim3(256), 0, stream, g, t_per_split, splits, ws, direct);
  COOT_CHECK_LAUNCH("gemm_tn");
  if (ws) {
    const long total4 = (long)g.Mo * g.No * g.groups / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g, splits, ws);
    COOT_CHECK_LAUNCH("gemm_tn_reduce");
  }
  timing_end(ts, stream);
  return 0;
}

}  // namespace coot

namespace coot {
COOT_DET_DEFINE_SETTER(gemm)
}  // namespace coot
