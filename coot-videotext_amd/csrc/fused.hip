// Fused token-tile chains (see fused.h).  gfx950, wave64, 512 threads = 8 waves per workgroup.
//
// Geometry of one 384-wide GEMM pass over a tile of BT = 16 * RF tokens:
//   wave w owns output columns [48 w, 48 w + 48) = 3 MFMA column fragments, for ALL row fragments of the tile
//   weights = MFMA A operand (16 features x 32 k), tokens = B operand  =>  a lane's 4 accumulators are 4 consecutive
//   features of one token:  acc[a][b][j] = C[token 16 a + (lane & 15)][feature 48 w + 16 b + 4 (lane >> 4) + j]
//   token fragments: ds_read_b128 from the LDS tile (row pitch 800 B: conflict free for the gfx950 lane groups)
//   weight fragments: one 16-byte global load per lane from the P48 pack (1 KB contiguous per wave-load, L2 resident),
//   register-prefetched 3 k-blocks ahead; no barrier inside a pass.
#include "fused.h"

#include <type_traits>

namespace coot {
namespace {

constexpr int APITCH = 400;  // bf16 elements per LDS tile row (800 B)
constexpr int SPITCH = 388;  // floats per staging row (1552 B: conflict-free b128 writes from the accumulator layout)
constexpr int NTHR = 512;
constexpr long GSZ = 12L * 3 * 512;  // bf16 elements of one 48-column weight group at K = 384

__device__ __forceinline__ void unpack8(u32x4_t u, float* v) {
  v[0] = bflo(u[0]); v[1] = bfhi(u[0]); v[2] = bflo(u[1]); v[3] = bfhi(u[1]);
  v[4] = bflo(u[2]); v[5] = bfhi(u[2]); v[6] = bflo(u[3]); v[7] = bfhi(u[3]);
}
__device__ __forceinline__ u32x4_t pack8(const float* v) {
  return u32x4_t{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
}
__device__ __forceinline__ void load8f(const float* p, float* v) {
  const f32x4_t a = reinterpret_cast<const f32x4_t*>(p)[0], b = reinterpret_cast<const f32x4_t*>(p)[1];
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void drop8(const DropCfg& d, unsigned long long idx0, float* sc) {
  drop_scales<8>(eff_seed(d.seed, d.seed_ptr), d.site, idx0, d.thr, d.inv_keep, sc);
}

// 16-byte global accesses addressed as (uniform base pointer) + (32-bit byte offset): lets hipcc use the SGPR-base +
// VGPR-offset form instead of materialising a 64-bit VGPR address per tensor per chunk (which it then spills)
__device__ __forceinline__ u32x4_t gld16(const void* base, unsigned boff) {
  return *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(base) + boff);
}
__device__ __forceinline__ void gst16(void* base, unsigned boff, u32x4_t v) {
  *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(base) + boff) = v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter (s_waitcnt
// vmcnt(0)): inside the epilogue rounds that made every round wait for the previous round's global STORES to be
// acknowledged and for the next round's prefetched loads to land — the rounds ran at one memory round trip each.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int RF>
__device__ __forceinline__ void zero_acc(f32x4_t (&acc)[RF][3]) {
#pragma unroll
  for (int a = 0; a < RF; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

// acc[a][b] += tile rows [16 a, 16 a + 16) x (48 columns of weight group wg), K = 32 KB
template <int RF, int KB>
__device__ __forceinline__ void gemm_pass(const bf16_t* As, const bf16_t* wg, f32x4_t (&acc)[RF][3], int lane) {
  const bf16_t* arow = As + (lane & 15) * APITCH + (lane >> 4) * 8;
  const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(wg) + lane;
  constexpr int PD = 2, NB = PD + 1;  // prefetch distance (k-blocks) and register ring size
  bf16x8_t w[NB][3];
#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (s < KB) {
#pragma unroll
      for (int b = 0; b < 3; ++b) w[s][b] = wp[(s * 3 + b) * 64];
    }
  bf16x8_t xf[2][RF];
#pragma unroll
  for (int a = 0; a < RF; ++a) xf[0][a] = *reinterpret_cast<const bf16x8_t*>(arow + a * 16 * APITCH);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + PD < KB) {
#pragma unroll
      for (int b = 0; b < 3; ++b) w[(kb + PD) % NB][b] = wp[((kb + PD) * 3 + b) * 64];
    }
    if (kb + 1 < KB) {
#pragma unroll
      for (int a = 0; a < RF; ++a) xf[(kb + 1) & 1][a] = *reinterpret_cast<const bf16x8_t*>(arow + a * 16 * APITCH + (kb + 1) * 32);
    }
    // pin the prefetches where they are written: without this hipcc sinks the weight loads next to their first use
    // (vmcnt(0) in front of every k-block: one full L2 round trip per 24 MFMAs)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < RF; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[kb % NB][b], xf[kb & 1][a], acc[a][b], 0, 0, 0);
  }
}

// global bf16 [T, ld] rows [row0, row0 + 16 RF), columns [col0, col0 + 384) -> LDS tile (rows >= T: zeros)
template <int RF>
__device__ __forceinline__ void load_tile(bf16_t* As, const bf16_t* src, long ld, int col0, int row0, int T) {
  constexpr int CH = 16 * RF * 48, IT = (CH + NTHR - 1) / NTHR;
  u32x4_t v[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = threadIdx.x + NTHR * i, rl = c / 48, ch = c - rl * 48;
    v[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (c < CH && row0 + rl < T) v[i] = gld16(src, (unsigned)((row0 + rl) * (int)ld + col0 + ch * 8) * 2u);
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = threadIdx.x + NTHR * i, rl = c / 48, ch = c - rl * 48;
    if (c < CH) *reinterpret_cast<u32x4_t*>(&As[rl * APITCH + ch * 8]) = v[i];
  }
}

// Accumulators -> fp32 staging (32 rows per round) -> per-chunk elementwise functor on 8 consecutive features of one
// token (all global traffic 16-byte, row contiguous) -> bf16 back into the LDS tile (the next GEMM's operand).
//   pre(row, col)          -> P      issue the chunk's global loads (all chunks of a round first: latencies overlap)
//   fn(row, col, v[8], P)            elementwise math + global stores; leaves the tile values in v
// bias chunks of the columns a thread meets in every epilogue round (issue the loads BEFORE the GEMM pass)
template <int NCH, int IT>
__device__ __forceinline__ void load_bias(const float* bias, float (&bs)[IT][8]) {
#pragma unroll
  for (int i = 0; i < IT; ++i) load8f(bias + ((threadIdx.x + NTHR * i) % NCH) * 8, bs[i]);
}

template <int RF, typename PreF, typename Fn>
__device__ __forceinline__ void epilogue(f32x4_t (&acc)[RF][3], float* Stg, bf16_t* As, int row0, int T, const float (&bs)[3][8], PreF pre, Fn fn,
                                         int dbg = 0) {
  constexpr int RR = RF >= 2 ? 32 : 16, FR = RR / 16, ROUNDS = 16 * RF / RR, CH = RR * 48, IT = (CH + NTHR - 1) / NTHR;
  static_assert(IT <= 3 && ROUNDS <= 4, "bias registers / round switch");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  decltype(pre(0, 0)) pv[2][IT];
  auto issue_pre = [&](int r, decltype(pre(0, 0)) (&dst)[IT]) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = tid + NTHR * i, rl = c / 48, ch = c - rl * 48, row = row0 + r * RR + rl;
      if (c < CH && row < T) dst[i] = pre(row, ch * 8);
    }
  };
  issue_pre(0, pv[0]);
  // The round loop is NOT unrolled (only the accumulator -> staging copy depends on the round, through a switch): unrolling
  // it quadruples the elementwise code of every epilogue and the kernel (190 KB) then streams its own instructions
  // from L2 — three times the 64 KB instruction cache two CUs share.
#pragma unroll 1
  for (int r = 0; r < ROUNDS; ++r) {
    lds_barrier();  // staging free again; (r == 0) every wave is done reading the tile in the GEMM
    float* sw = &Stg[(lane & 15) * SPITCH + wave * 48 + (lane >> 4) * 4];
    auto put = [&](auto rc) {
      constexpr int R = decltype(rc)::value;
#pragma unroll
      for (int a2 = 0; a2 < FR; ++a2)
#pragma unroll
        for (int b = 0; b < 3; ++b) *reinterpret_cast<f32x4_t*>(sw + a2 * 16 * SPITCH + b * 16) = acc[R * FR + a2][b];
    };
    switch (r) {
      case 0: put(std::integral_constant<int, 0>{}); break;
      case 1: if constexpr (ROUNDS > 1) put(std::integral_constant<int, 1>{}); break;
      case 2: if constexpr (ROUNDS > 2) put(std::integral_constant<int, 2>{}); break;
      default: if constexpr (ROUNDS > 3) put(std::integral_constant<int, 3>{}); break;
    }
    const int par = r & 1;
    if (r + 1 < ROUNDS) {  // the next round's global loads fly during this round
      if (par) issue_pre(r + 1, pv[0]); else issue_pre(r + 1, pv[1]);
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = tid + NTHR * i, rl = c / 48, ch = c - rl * 48, row = row0 + r * RR + rl;
      if (c < CH) {
        float v[8];
        load8f(&Stg[rl * SPITCH + ch * 8], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bs[i][j];
        if (row < T) { if (!(dbg & 2)) fn(row, ch * 8, v, par ? pv[1][i] : pv[0][i]); }
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        *reinterpret_cast<u32x4_t*>(&As[(r * RR + rl) * APITCH + ch * 8]) = pack8(v);
      }
    }
  }
  lds_barrier();
}

// The same for a 192-column GEMM computed as 2 row halves x 4 column groups (wave = 4 rh + cg), accumulators
// acc[RF / 2][3]; results are not written back to the tile.
template <int RF, typename PreF, typename Fn>
__device__ __forceinline__ void epilogue_half(f32x4_t (&acc)[RF / 2][3], float* Stg, int row0, int T, const float (&bs)[2][8], PreF pre, Fn fn, int dbg = 0) {
  static_assert(RF >= 4 && RF % 4 == 0, "row halves of whole 32-row rounds");
  constexpr int RR = 32, ROUNDS = 16 * RF / RR, HR = ROUNDS / 2, CH = RR * 24, IT = (CH + NTHR - 1) / NTHR;
  static_assert(IT == 2, "bias registers");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cg = wave & 3, rh = wave >> 2;
#pragma unroll 1
  for (int r = 0; r < ROUNDS; ++r) {
    lds_barrier();
    if (rh == r / HR) {
      float* sw = &Stg[(lane & 15) * SPITCH + cg * 48 + (lane >> 4) * 4];
      auto put = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
          for (int b = 0; b < 3; ++b) *reinterpret_cast<f32x4_t*>(sw + a2 * 16 * SPITCH + b * 16) = acc[R * 2 + a2][b];
      };
      static_assert(HR == 2, "two rounds per row half");
      if (r % HR == 0) put(std::integral_constant<int, 0>{}); else put(std::integral_constant<int, 1>{});
    }
    decltype(pre(0, 0)) pv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = tid + NTHR * i, rl = c / 24, ch = c - rl * 24, row = row0 + r * RR + rl;
      if (c < CH && row < T) pv[i] = pre(row, ch * 8);
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = tid + NTHR * i, rl = c / 24, ch = c - rl * 24, row = row0 + r * RR + rl;
      if (c < CH && row < T) {
        float v[8];
        load8f(&Stg[rl * SPITCH + ch * 8], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bs[i][j];
        if (!(dbg & 2)) fn(row, ch * 8, v, pv[i]);
      }
    }
  }
  lds_barrier();
}

// sum over the 16 lanes of a DPP row (full-rate VALU, no LDS crossbar): every lane of the row receives the total
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
  return v;
}

// COOT LayerNorm (nntrainer/models/normalizations.py:98-101) of every tile row, in place.  Wave w owns rows
// [2 RF w, 2 RF w + 2 RF); 16 lanes share a row (4 rows per wave in flight): lane j holds the three 8-element chunks at
// columns 8 j, 128 + 8 j, 256 + 8 j (16-byte LDS / global accesses), reductions are 4 DPP steps.
template <int RF>
__device__ __forceinline__ void ln_tile(bf16_t* As, const float* gain, const float* bias, int row0, int T, bf16_t* out, float* out32,
                                        long ld32, const DropCfg& drop) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j16 = lane & 15, g = lane >> 4;
  constexpr int RW = 2 * RF, ITR = (RW + 3) / 4;
  float gn[3][8], bi[3][8];
#pragma unroll
  for (int m = 0; m < 3; ++m) { load8f(gain + m * 128 + j16 * 8, gn[m]); load8f(bias + m * 128 + j16 * 8, bi[m]); }
#pragma unroll
  for (int it = 0; it < ITR; ++it) {
    const int rw = it * 4 + g;
    if (RW % 4 != 0 && rw >= RW) break;
    const int rl = wave * RW + rw, row = row0 + rl;
    bf16_t* ar = As + rl * APITCH + j16 * 8;
    float x[3][8];
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      unpack8(*reinterpret_cast<const u32x4_t*>(ar + m * 128), x[m]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[m][e];
    }
    const float mean = row16_sum(s) * (1.0f / 384.0f);
    float q = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[m][e] -= mean; q += x[m][e] * x[m][e]; }
    const float rs = 1.0f / (sqrtf(row16_sum(q) * (1.0f / 383.0f)) + kLnEps);
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = x[m][e] * rs * gn[m][e] + bi[m][e];
      if (drop.thr) {
        float sc[8];
        drop_scales<8>(eff_seed(drop.seed, drop.seed_ptr), drop.site, (unsigned long long)row * FZ_D + m * 128 + j16 * 8, drop.thr, drop.inv_keep, sc);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] *= sc[e];
      }
      const u32x4_t o = pack8(y);
      *reinterpret_cast<u32x4_t*>(ar + m * 128) = o;
      if (row < T) {
        if (out) gst16(out, (unsigned)(row * FZ_D + m * 128 + j16 * 8) * 2u, o);
        if (out32) {
          float* o32 = out32 + (long)row * ld32 + m * 128 + j16 * 8;
          *reinterpret_cast<f32x4_t*>(o32) = f32x4_t{y[0], y[1], y[2], y[3]};
          *reinterpret_cast<f32x4_t*>(o32 + 4) = f32x4_t{y[4], y[5], y[6], y[7]};
        }
      }
    }
  }
}

struct PreNone {};
struct PreRes { u32x4_t res; };

template <int RF>
__global__ __launch_bounds__(512) void post_attn_fwd_kernel(PostAttnFwd p) {
  constexpr int BT = 16 * RF, RR = RF >= 2 ? 32 : 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[BT * APITCH * 2 + RR * SPITCH * 4];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  float* Stg = reinterpret_cast<float*>(smem + BT * APITCH * 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * BT, T = p.T;
  f32x4_t acc[RF][3];
  int tsn = 0;
  auto stamp = [&]() { if (p.tstamps && blockIdx.x == 0 && threadIdx.x == 0) p.tstamps[tsn] = __builtin_amdgcn_s_memtime(); ++tsn; };
  stamp();

  // ---- attention output projection + residual -> r1 ------------------------------------------------------------
  load_tile<RF>(As, p.ctx, FZ_D, 0, row0, T);
  __syncthreads();
  stamp();
  float bs[3][8];
  load_bias<48, 3>(p.bo, bs);
  zero_acc<RF>(acc);
  if (!(p.debug & 1)) gemm_pass<RF, 12>(As, p.wo + wave * GSZ, acc, lane);
  stamp();
  epilogue<RF>(acc, Stg, As, row0, T, bs,
      [&](int row, int col) { return PreRes{gld16(p.xres, (unsigned)(row * FZ_D + col) * 2u)}; },
      [&](int row, int col, float (&v)[8], const PreRes& pr) {
        float r[8];
        unpack8(pr.res, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        gst16(p.r1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, p.debug);
  stamp();
  // ---- LN1 (+ dropout) -> z1 ---------------------------------------------------------------------------------------
  if (!(p.debug & 4)) ln_tile<RF>(As, p.ln1g, p.ln1b, row0, T, p.z1, nullptr, 0, p.d_postln);
  __syncthreads();
  stamp();
  // ---- FF1: Linear -> Dropout -> GELU ------------------------------------------------------------------------------
  load_bias<48, 3>(p.b1, bs);
  zero_acc<RF>(acc);
  if (!(p.debug & 1)) gemm_pass<RF, 12>(As, p.w1 + wave * GSZ, acc, lane);
  stamp();
  epilogue<RF>(acc, Stg, As, row0, T, bs, [&](int, int) { return PreNone{}; },
      [&](int row, int col, float (&v)[8], const PreNone&) {
        if (p.d_ff1.thr) {
          float sc[8];
          drop8(p.d_ff1, (unsigned long long)row * FZ_D + col, sc);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= sc[j];
        }
        gst16(p.h1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
        gst16(p.a1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, p.debug);
  stamp();
  // ---- FF2: Linear -> Dropout, + residual z1 -> r2 -----------------------------------------------------------------
  load_bias<48, 3>(p.b2, bs);
  zero_acc<RF>(acc);
  if (!(p.debug & 1)) gemm_pass<RF, 12>(As, p.w2 + wave * GSZ, acc, lane);
  epilogue<RF>(acc, Stg, As, row0, T, bs,
      [&](int row, int col) { return PreRes{gld16(p.z1, (unsigned)(row * FZ_D + col) * 2u)}; },
      [&](int row, int col, float (&v)[8], const PreRes& pr) {
        float r[8];
        unpack8(pr.res, r);
        if (p.d_ff2.thr) {
          float sc[8];
          drop8(p.d_ff2, (unsigned long long)row * FZ_D + col, sc);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= sc[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        gst16(p.r2, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, p.debug);
  stamp();
  // ---- LN2 -> z2 ------------------------------------------------------------------------------------------------------
  {
    DropCfg none;
    if (!(p.debug & 4)) ln_tile<RF>(As, p.ln2g, p.ln2b, row0, T, p.z2, p.z2_f32, p.ldz2_f32, none);
  }
  __syncthreads();
  stamp();
  if constexpr (RF >= 4) {
    if (!p.do_pool) return;
    // ---- GenPool scores (poolers.py:171-181): per head h, a = GELU(dropout(z W1[h] + b1[h])), s = dropout(a W2[h] + b2[h]) ----
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      if (h == 1) {  // the tile holds a_0 now: bring z2 back (just written, L2 resident)
        load_tile<RF>(As, p.z2, FZ_D, 0, row0, T);
        __syncthreads();
      }
      load_bias<48, 3>(p.pb1 + h * FZ_D, bs);
      zero_acc<RF>(acc);
      if (!(p.debug & 1)) gemm_pass<RF, 12>(As, p.pw1 + (h * 8 + wave) * GSZ, acc, lane);
      epilogue<RF>(acc, Stg, As, row0, T, bs, [&](int, int) { return PreNone{}; },
          [&](int row, int col, float (&v)[8], const PreNone&) {
            if (p.d_pool1.thr) {
              float sc[8];
              drop8(p.d_pool1, (unsigned long long)row * (2 * FZ_D) + h * FZ_D + col, sc);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= sc[j];
            }
            gst16(p.hp, (unsigned)(row * (2 * FZ_D) + h * FZ_D + col) * 2u, pack8(v));
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
            gst16(p.ap, (unsigned)(row * (2 * FZ_D) + h * FZ_D + col) * 2u, pack8(v));
          }, p.debug);
      // FC2 of head h: 192 output columns = 4 column groups x 2 row halves
      f32x4_t acc2[RF / 2][3];
      float bs2[2][8];
      load_bias<24, 2>(p.pb2 + h * (FZ_D / 2), bs2);
      zero_acc<RF / 2>(acc2);
      if (!(p.debug & 1)) gemm_pass<RF / 2, 12>(As + (wave >> 2) * (BT / 2) * APITCH, p.pw2 + (h * 4 + (wave & 3)) * GSZ, acc2, lane);
      epilogue_half<RF>(acc2, Stg, row0, T, bs2, [&](int, int) { return PreNone{}; },
          [&](int row, int col, float (&v)[8], const PreNone&) {
            if (p.d_pool2.thr) {
              float sc[8];
              drop8(p.d_pool2, (unsigned long long)row * FZ_D + h * (FZ_D / 2) + col, sc);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= sc[j];
            }
            gst16(p.s, (unsigned)(row * FZ_D + h * (FZ_D / 2) + col) * 2u, pack8(v));
          }, p.debug);
      stamp();
    }
  }
}

}  // namespace

int launch_post_attn_fwd(const PostAttnFwd& p, hipStream_t st) {
  COOT_REQUIRE(p.ctx && p.xres && p.wo && p.w1 && p.w2 && p.bo && p.b1 && p.b2 && p.ln1g && p.ln1b && p.ln2g && p.ln2b && p.r1 && p.z1 &&
               p.h1 && p.a1 && p.r2 && p.z2, "post_attn_fwd: null pointer");
  COOT_REQUIRE(!p.do_pool || (p.pw1 && p.pw2 && p.pb1 && p.pb2 && p.hp && p.ap && p.s), "post_attn_fwd: pooling pointers");
  if (p.T <= 0) return 0;
  hipLaunchKernelGGL(post_attn_fwd_kernel<8>, dim3((p.T + 127) / 128), dim3(NTHR), 0, st, p);
  COOT_CHECK_LAUNCH("post_attn_fwd");
  return 0;
}

}  // namespace coot
