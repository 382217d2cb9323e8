// bf16 MFMA GEMM kernels for gfx950 (wave64, v_mfma_f32_16x16x32_bf16).
//
// Tile: BM tokens x 128 features x 64 K per step, 256 threads = 4 waves (2 x 2), each wave
// owns (BM/2) x 64 of the output as 16x16 MFMA fragments.  The WEIGHT rows are the MFMA
// A-operand and the TOKEN rows the B-operand, so one lane's 4 accumulator registers are 4
// consecutive output features of one token -> 8/16-byte epilogue stores.
// Operands go HBM -> registers -> LDS (register staged, double-buffered LDS, one barrier per
// K step); rows are padded to 80 elements (160 B) which makes every ds_read_b128 fragment read
// bank-conflict free (checked against the gfx950 lane-group table).
#include "gemm.h"

namespace coot {

constexpr int BN = 128;
constexpr int BK = 64;
constexpr int PITCH = BK + 16;  // elements; 160 B rows

__device__ __forceinline__ u32x4_t load16_guard(const bf16_t* p, bool ok) {
  u32x4_t z = {0u, 0u, 0u, 0u};
  if (ok) z = *reinterpret_cast<const u32x4_t*>(p);
  return z;
}

template <int BM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNT g) {
  constexpr int TM = BM / 32;      // token fragments per wave
  constexpr int XCH = BM * 8 / 256;  // 16-byte chunks per thread for the token tile
  constexpr int WCH = BN * 8 / 256;
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2][BM * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t Ws[2][BN * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;
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
      *reinterpret_cast<u32x4_t*>(&Xs[buf][r * PITCH + kc]) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      int c = tid + 256 * i, r = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<u32x4_t*>(&Ws[buf][r * PITCH + kc]) = wr[i];
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
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8_t xf[TM], wf[4];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        xf[a] = *reinterpret_cast<const bf16x8_t*>(&Xs[buf][(wm * (BM / 2) + a * 16 + frow) * PITCH + kk * 32 + fk]);
#pragma unroll
      for (int b = 0; b < 4; ++b)
        wf[b] = *reinterpret_cast<const bf16x8_t*>(&Ws[buf][(wn * 64 + b * 16 + frow) * PITCH + kk * 32 + fk]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds token (lane&15) x 4 consecutive features ((lane>>4)*4 .. +3) ----
  const GemmEpi& e = g.epi;
  const long zo = z * g.zOut;
  float cs[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[b][j] = 0.f;

#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int row = row0 + wm * (BM / 2) + a * 16 + (lane & 15);
    const bool rok = row < M;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = col0 + wn * 64 + b * 16 + (lane >> 4) * 4;
      if (!rok || col >= N) continue;  // N % 4 == 0 is required by the launcher
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j] * e.alpha;
      if (e.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += e.bias[zo + col + j];
      }
      float dsc[4] = {1.f, 1.f, 1.f, 1.f};
      if (e.drop_thr) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          dsc[j] = drop_scale(e.drop_seed, e.drop_site, (unsigned long long)row * e.drop_ld + zo + col + j,
                              e.drop_thr, e.drop_inv_keep);
        if (e.act != 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= dsc[j];
        }
      }
      if (e.save_pre) {
        u32x2_t pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(e.save_pre + (long)row * e.ldpre + zo + col) = pk;
      }
      if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_f(v[j]);
      }
      if (e.pe) {
        const int pos = row < e.pe_T0 ? row % e.pe_L : (row - e.pe_T0) % e.pe_L2;
        const float* pp = e.pe + (long)pos * N + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += pp[j];
      }
      if (e.res) {
        u32x2_t rx = *reinterpret_cast<const u32x2_t*>(e.res + (long)row * e.ldres + zo + col);
        v[0] += bflo(rx[0]); v[1] += bfhi(rx[0]); v[2] += bflo(rx[1]); v[3] += bfhi(rx[1]);
      }
      if (e.res32) {
        const float* rp = e.res32 + (long)row * e.ldres32 + zo + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rp[j];
      }
      if (e.rowscale) {
        const float rsv = e.rowscale[row];
        u32x2_t dx = *reinterpret_cast<const u32x2_t*>(e.diag_src + (long)row * e.lddiag + col);
        v[0] += rsv * bflo(dx[0]); v[1] += rsv * bfhi(dx[0]); v[2] += rsv * bflo(dx[1]); v[3] += rsv * bfhi(dx[1]);
      }
      if (e.act == 2) {  // backward through GELU (and its dropout): applied after the residual adds
        u32x2_t ax = *reinterpret_cast<const u32x2_t*>(e.aux + (long)row * e.ldaux + zo + col);
        v[0] *= gelu_grad_f(bflo(ax[0])) * dsc[0];
        v[1] *= gelu_grad_f(bfhi(ax[0])) * dsc[1];
        v[2] *= gelu_grad_f(bflo(ax[1])) * dsc[2];
        v[3] *= gelu_grad_f(bfhi(ax[1])) * dsc[3];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[b][j] += v[j];
      if (e.out_f32) {
        float* op = reinterpret_cast<float*>(e.out) + (long)row * e.ldc + zo + col;
        if (e.accumulate) {
#pragma unroll
          for (int j = 0; j < 4; ++j) op[j] += v[j];
        } else {
          *reinterpret_cast<f32x4_t*>(op) = f32x4_t{v[0], v[1], v[2], v[3]};
        }
      } else {
        u32x2_t pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(e.out) + (long)row * e.ldc + zo + col) = pk;
      }
    }
  }
  if (e.colsum) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = col0 + wn * 64 + b * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = cs[b][j];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
        if ((lane & 15) == 0 && col + j < N) atomicAdd(e.colsum + zo + col + j, s);
      }
    }
  }
}

// ---- optional per-launch timing with HIP events (bench.py roofline leg) -------------------------------
struct TimingSlot { hipEvent_t a, b; double flops; int big_k; };
static int g_timing_on = 0;
static TimingSlot g_slots[8192];
static int g_nslots = 0, g_slots_created = 0;
void gemm_timing_enable(int on) { g_timing_on = on; g_nslots = 0; }
int gemm_timing_collect(int only_big_k, double* ms, double* flops, int* launches) {
  double tm = 0, fl = 0; int n = 0;
  for (int i = 0; i < g_nslots; ++i) {
    if (only_big_k && !g_slots[i].big_k) continue;
    if (hipEventSynchronize(g_slots[i].b) != hipSuccess) return -1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_slots[i].a, g_slots[i].b) != hipSuccess) return -1;
    tm += t; fl += g_slots[i].flops; ++n;
  }
  *ms = tm; *flops = fl; *launches = n;
  return 0;
}
static TimingSlot* timing_begin(const GemmNT& g, hipStream_t stream) {
  if (!g_timing_on || g_nslots >= 8192) return nullptr;
  TimingSlot* s = &g_slots[g_nslots];
  if (g_nslots >= g_slots_created) { hipEventCreate(&s->a); hipEventCreate(&s->b); g_slots_created = g_nslots + 1; }
  ++g_nslots;
  s->flops = 2.0 * g.M * g.N * g.K * g.groups; s->big_k = g.K >= 1024;
  hipEventRecord(s->a, stream);
  return s;
}

int launch_gemm_nt(const GemmNT& g, hipStream_t stream) {
  COOT_REQUIRE(g.X && g.W && g.epi.out, "gemm_nt: null operand");
  COOT_REQUIRE(g.K % 8 == 0 && g.ldx % 8 == 0 && g.ldw % 8 == 0, "gemm_nt: K/ldx/ldw must be multiples of 8 (K=%d ldx=%ld ldw=%ld)", g.K, g.ldx, g.ldw);
  COOT_REQUIRE(g.N % 4 == 0 && g.epi.ldc % 4 == 0 && g.zOut % 4 == 0, "gemm_nt: N/ldc must be multiples of 4 (N=%d ldc=%ld)", g.N, g.epi.ldc);
  COOT_REQUIRE(!(g.epi.accumulate && !g.epi.out_f32), "gemm_nt: accumulate needs fp32 out");
  if (g.M <= 0 || g.N <= 0) return 0;
  const int nb = (g.N + BN - 1) / BN;
  // small-M problems: halve the token tile to get more workgroups onto the 256 CUs
  const long blocks128 = (long)((g.M + 127) / 128) * nb * g.groups;
  TimingSlot* ts = timing_begin(g, stream);
  if (blocks128 >= 384) {
    dim3 grid(nb, (g.M + 127) / 128, g.groups);
    hipLaunchKernelGGL(gemm_nt_kernel<128>, grid, dim3(256), 0, stream, g);
  } else {
    dim3 grid(nb, (g.M + 63) / 64, g.groups);
    hipLaunchKernelGGL(gemm_nt_kernel<64>, grid, dim3(256), 0, stream, g);
  }
  if (ts) hipEventRecord(ts->b, stream);
  COOT_CHECK_LAUNCH("gemm_nt");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// TN: C[Mo,No] += alpha * sum_t A[t,Mo] * B[t,No]   (reduction over the row index)
// Both operands are loaded row-major ([t][col], coalesced along col) and the MFMA fragments are
// formed either with the LDS transpose read ds_read_b64_tr_b16 (mode 0) or from a transposed
// LDS image written with 2-byte stores (mode 1, fallback).
// ---------------------------------------------------------------------------------------------
constexpr int TN_BT = 64;           // t rows per step
constexpr int TN_BC = 128;          // columns per operand tile
constexpr int TN_PITCH = TN_BC + 16;  // mode 0: [t][col], 288-byte rows (8 consecutive rows -> 64 distinct banks)

template <int MODE>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN g, int t_per_split) {
  // mode 0: As[t][col] (pitch TN_PITCH).  mode 1: At[col][t] (pitch PITCH).
  constexpr int ASZ = MODE == 0 ? TN_BT * TN_PITCH : TN_BC * PITCH;
  __shared__ __attribute__((aligned(16))) bf16_t As[2][ASZ];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[2][ASZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.z % g.groups, split = blockIdx.z / g.groups;
  const int m0 = blockIdx.y * TN_BC, n0 = blockIdx.x * TN_BC;
  const int t_begin = split * t_per_split;
  const int t_end = min(g.T, t_begin + t_per_split);
  if (t_begin >= t_end) return;
  const bf16_t* A = g.A + z * g.zA;
  const bf16_t* B = g.B + z * g.zB;
  float* C = g.C + z * g.zC;

  // tile = 64 rows x 128 cols = 1024 chunks of 16 B; 4 per thread per operand
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
        *reinterpret_cast<u32x4_t*>(&As[buf][r * TN_PITCH + cc]) = ar[i];
        *reinterpret_cast<u32x4_t*>(&Bs[buf][r * TN_PITCH + cc]) = br[i];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          As[buf][(cc + 2 * j) * PITCH + r] = (bf16_t)(ar[i][j] & 0xFFFFu);
          As[buf][(cc + 2 * j + 1) * PITCH + r] = (bf16_t)(ar[i][j] >> 16);
          Bs[buf][(cc + 2 * j) * PITCH + r] = (bf16_t)(br[i][j] & 0xFFFFu);
          Bs[buf][(cc + 2 * j + 1) * PITCH + r] = (bf16_t)(br[i][j] >> 16);
        }
      }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nsteps = (t_end - t_begin + TN_BT - 1) / TN_BT;
  gload(t_begin);
  sstore(0);
  __syncthreads();
  const int grp = lane >> 4, p = lane & 15;
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    if (st + 1 < nsteps) gload(t_begin + (st + 1) * TN_BT);
#pragma unroll
    for (int ks = 0; ks < TN_BT / 32; ++ks) {
      bf16x8_t af[4], bfr[4];
      if (MODE == 0) {
        // k-slot (grp*8 + j)     <-> t = ks*32 +      4*grp + j   (first  tr read)
        // k-slot (grp*8 + 4 + j) <-> t = ks*32 + 16 + 4*grp + j   (second tr read)
        // lane p of a 16-lane group supplies the 8-byte row chunk [t = base + (p>>2)][c0 + (p&3)*4 .. +3]
        const int trow = ks * 32 + 4 * grp + (p >> 2);
        const int tcol = (p & 3) * 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const bf16_t* pa = &As[buf][trow * TN_PITCH + wm * 64 + a * 16 + tcol];
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pa));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pa + 16 * TN_PITCH));
          typedef short s16x8_t __attribute__((ext_vector_type(8)));
          s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          af[a] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bf16_t* pb = &Bs[buf][trow * TN_PITCH + wn * 64 + b * 16 + tcol];
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb + 16 * TN_PITCH));
          typedef short s16x8_t __attribute__((ext_vector_type(8)));
          s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfr[b] = __builtin_bit_cast(bf16x8_t, v);
        }
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          af[a] = *reinterpret_cast<const bf16x8_t*>(&As[buf][(wm * 64 + a * 16 + p) * PITCH + ks * 32 + grp * 8]);
#pragma unroll
        for (int b = 0; b < 4; ++b)
          bfr[b] = *reinterpret_cast<const bf16x8_t*>(&Bs[buf][(wn * 64 + b * 16 + p) * PITCH + ks * 32 + grp * 8]);
      }
      // MFMA-A = B-matrix columns (output col), MFMA-B = A-matrix columns (output row):
      // lane reg j = C[row = m0 + wm*64 + a*16 + (lane&15)][col = n0 + wn*64 + b*16 + (lane>>4)*4 + j]
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
    }
    if (st + 1 < nsteps) sstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = m0 + wm * 64 + a * 16 + (lane & 15);
    if (row >= g.Mo) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = n0 + wn * 64 + b * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (col + j < g.No) atomicAdd(C + (long)row * g.ldc + col + j, acc[a][b][j] * g.alpha);
    }
  }
}

static int g_tn_mode = 0;
void set_tn_mode(int mode) { g_tn_mode = mode ? 1 : 0; }
int get_tn_mode() { return g_tn_mode; }

int launch_gemm_tn(const GemmTN& g, hipStream_t stream) {
  COOT_REQUIRE(g.A && g.B && g.C, "gemm_tn: null operand");
  COOT_REQUIRE(g.lda % 8 == 0 && g.ldb % 8 == 0 && g.Mo % 8 == 0 && g.No % 8 == 0 && g.zA % 8 == 0 && g.zB % 8 == 0,
               "gemm_tn: lda/ldb/Mo/No must be multiples of 8 (Mo=%d No=%d lda=%ld ldb=%ld)", g.Mo, g.No, g.lda, g.ldb);
  if (g.T <= 0 || g.Mo <= 0 || g.No <= 0) return 0;
  const int tiles = ((g.Mo + TN_BC - 1) / TN_BC) * ((g.No + TN_BC - 1) / TN_BC) * g.groups;
  // split the t reduction so that ~512 workgroups exist; each split handles a multiple of 64 rows
  int splits = (512 + tiles - 1) / tiles;
  const int max_splits = (g.T + 4 * TN_BT - 1) / (4 * TN_BT);  // >= 256 rows per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int t_per_split = (g.T + splits - 1) / splits;
  t_per_split = (t_per_split + TN_BT - 1) / TN_BT * TN_BT;
  splits = (g.T + t_per_split - 1) / t_per_split;
  dim3 grid((g.No + TN_BC - 1) / TN_BC, (g.Mo + TN_BC - 1) / TN_BC, g.groups * splits);
  if (g_tn_mode == 0)
    hipLaunchKernelGGL(gemm_tn_kernel<0>, grid, dim3(256), 0, stream, g, t_per_split);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, dim3(256), 0, stream, g, t_per_split);
  COOT_CHECK_LAUNCH("gemm_tn");
  return 0;
}

}  // namespace coot
