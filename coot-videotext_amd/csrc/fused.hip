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
#include "det.h"
#include "gemm.h"

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
#ifdef FZ_NO_STORES
  // measurement build (tools/build_variant.sh nostore "-DFZ_NO_STORES"; wrong results): every 16-byte global store of the chain kernels
  // is kept in the code — address, packed value, the branch — but never executes (no offset has this value), so what remains is the
  // kernels' time WITHOUT their store traffic: the upper bound of anything that overlaps stores with GEMM passes (profiles/README.md round 5)
  if (boff != 0xFFFFFFF0u) return;
#endif
  *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(base) + boff) = v;
}

// Measurement build only (tools/build_variant.sh xcd5 "-DFZ_XCDS=5"): the forward chain kernels' tiles are dealt to the first FZ_XCDS
// XCDs only (workgroup b runs on XCD b % 8; the launchers widen the grid, the other XCDs' workgroups return at once) — does a tile's
// latency at a given NUMBER of resident tiles depend on how many share one XCD's L2 (profiles/README.md round 5)?
#ifdef FZ_XCDS
__device__ __forceinline__ int fz_tile_index(int ntiles) {
  const int x = blockIdx.x & 7;
  if (x >= FZ_XCDS) return -1;
  const int t = (blockIdx.x >> 3) * FZ_XCDS + x;
  return t < ntiles ? t : -1;
}
static inline int fz_grid(int tiles) { return (tiles + FZ_XCDS - 1) / FZ_XCDS * 8; }
#else
__device__ __forceinline__ int fz_tile_index(int) { return blockIdx.x; }
static inline int fz_grid(int tiles) { return tiles; }
#endif

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

// ---- one 384-wide GEMM pass: acc[a][b] += tile rows [16 a, 16 a + 16) x (48 columns of weight group wg), K = 32 KB -------------
// The weight fragments of a pass go through a register ring of NB = PD + 1 k-blocks, PD of them in flight.  The ring is CARRIED from
// pass to pass: the last PD iterations of a pass load the first PD k-blocks of the pass that FOLLOWS it (`next`), so those fly during
// the epilogue / LayerNorm / attention phase between the two passes and a pass starts on data that has landed — a chain kernel is a
// sequence of 6-12 passes separated by such phases, and each used to start with an exposed L2 round trip (the global networks' single-
// launch passes are latency chains: there it was ~1.6 us per pass, a fifth of the kernel).  hipcc did some of this hoisting on its own
// as long as every pass's addresses were kernel-wide values — the same values it then spilled; with per-pass addresses (the launders
// that keep the kernels out of scratch) it cannot, so the schedule is explicit.
//   PD: small tiles (the global networks, RF <= 2) do almost no MFMA work per k-block, a pass is the latency of streaming 295 KB of
//   weights: 5 k-blocks (15 KB per wave) in flight.  Full tiles: a pass is at the MFMA time of 2 waves/SIMD, PD = 2 (3 / 4: no faster).
//   NB must divide every pass's KB (12, 6): a pass then starts at ring slot 0 whatever preceded it.
constexpr int kPdFull = 2;       // k-blocks in flight on full tiles (3 / 4 measured no faster)
#ifndef FZ_PD_SMALL
#define FZ_PD_SMALL 5
#endif
constexpr int kPdSmall = FZ_PD_SMALL;  // k-blocks in flight on 16 / 32-row tiles (the global networks' latency chains)
//   CARRY: 128-row tiles do not carry (the PD x 12 ring registers live across an epilogue that already holds 96 accumulators + its
//   chunk registers: 104-148 B of scratch per lane; measured 0.25 % faster WITH the spills, profiles/r04_ab_carry_ring.txt — not
//   adopted); their passes start with their own first k-blocks.  A ring type's CARRY decides for every pass run on it.  (Round 4 kept
//   these as -D experiment macros; the measured settings are now constants.)
constexpr int kCarryMaxRF = 4;
template <int PD_, bool CARRY_>
struct WRing { static constexpr int PD = PD_, NB = PD_ + 1; static constexpr bool CARRY = CARRY_; bf16x8_t w[PD_ + 1][3]; };
template <int RF> using TileRing = WRing<(RF <= 2 ? kPdSmall : kPdFull), (RF <= kCarryMaxRF)>;

// the first PD k-blocks of weight group wg -> ring slots 0 .. PD - 1 (the first pass of a kernel; later passes inherit theirs)
template <typename Ring>
__device__ __forceinline__ void gemm_load_first(Ring& R, const bf16_t* wg, int lane) {
  const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(wg) + lane;
#ifndef FZ_NO_WLOAD  // (measurement build FZ_NO_WLOAD: the passes run on whatever the ring registers hold — no weight stream from L2)
#pragma unroll
  for (int s = 0; s < Ring::PD; ++s)
#pragma unroll
    for (int b = 0; b < 3; ++b) R.w[s][b] = wp[(s * 3 + b) * 64];
#else
  (void)wp;
#endif
  // pin the loads where they are written: hipcc otherwise sinks them next to their first use
  __builtin_amdgcn_sched_barrier(0);
}
template <typename Ring>
__device__ __forceinline__ void gemm_issue(Ring& R, const bf16_t* wg, int lane) {
  if constexpr (Ring::CARRY) gemm_load_first(R, wg, lane);
}

// the pass proper; the ring holds k-blocks 0 .. PD - 1 of wg.  NEXT: the last PD iterations load k-blocks 0 .. PD - 1 of `next`.
template <int RF, int KB, bool NEXT, typename Ring>
__device__ __forceinline__ void gemm_run(Ring& R, const bf16_t* As, const bf16_t* wg, const bf16_t* next, f32x4_t (&acc)[RF][3], int lane) {
  constexpr int PD = Ring::PD, NB = Ring::NB;
  static_assert(KB % NB == 0 && KB >= PD, "the ring must wrap at the end of a pass");
  asm volatile("" : "+v"(lane));  // this pass's fragment addresses are its own (not shared with — and kept live for — the other passes)
  if constexpr (!Ring::CARRY) gemm_load_first(R, wg, lane);  // nothing is carried: the pass starts with its own L2 round trip
  const bf16_t* arow = As + (lane & 15) * APITCH + (lane >> 4) * 8;
  const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(wg) + lane;
  const bf16x8_t* np = reinterpret_cast<const bf16x8_t*>(next) + lane;
  // token fragments: ONE register set, refreshed row fragment by row fragment — xf[a] of the next k-block is read right
  // after the three MFMAs that consume the current xf[a] (its next use is 21 MFMAs away, several LDS latencies).  A
  // second full set (double buffering) costs 4 RF registers the epilogues then spill.
  bf16x8_t xf[RF];
#pragma unroll
  for (int a = 0; a < RF; ++a) xf[a] = *reinterpret_cast<const bf16x8_t*>(arow + a * 16 * APITCH);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
#ifndef FZ_NO_WLOAD
    if (kb + PD < KB) {
#pragma unroll
      for (int b = 0; b < 3; ++b) R.w[(kb + PD) % NB][b] = wp[((kb + PD) * 3 + b) * 64];
    } else if constexpr (NEXT && Ring::CARRY) {
#pragma unroll
      for (int b = 0; b < 3; ++b) R.w[(kb + PD) % NB][b] = np[((kb + PD - KB) * 3 + b) * 64];
    }
#else
    (void)wp; (void)np;
#endif
    // pin the prefetches where they are written: without this hipcc sinks the weight loads next to their first use
    // (vmcnt(0) in front of every k-block: one full L2 round trip per 24 MFMAs)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < RF; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) {
#ifndef FZ_NO_MFMA  // (measurement build FZ_NO_MFMA: the operands are consumed by one VALU op per fragment instead of the matrix pipe)
        acc[a][b] = COOT_MFMA_16x16x32(R.w[kb % NB][b], xf[a], acc[a][b]);
#else
        acc[a][b][0] += __builtin_bit_cast(float, (int)(__builtin_bit_cast(u32x4_t, R.w[kb % NB][b])[0] ^ __builtin_bit_cast(u32x4_t, xf[a])[b]));
#endif
      }
      if (kb + 1 < KB) xf[a] = *reinterpret_cast<const bf16x8_t*>(arow + a * 16 * APITCH + (kb + 1) * 32);
    }
  }
}

// global bf16 [., ld] rows [row0, row0 + 16 RF), columns [col0, col0 + 384) -> LDS tile (whole tiles are allocated)
template <int RF, int CPR = 48>
__device__ __forceinline__ void load_tile(bf16_t* As, const bf16_t* src, long ld, int col0, int row0) {
  constexpr int CH = 16 * RF * CPR, IT = (CH + NTHR - 1) / NTHR;
  // launder the (uniform) tile origin: otherwise hipcc shares the 12 per-thread offsets of this call with every later
  // phase that addresses the same rows (z2 reload, residual loads) and keeps them alive — in scratch — across the kernel
  asm volatile("" : "+s"(row0));
  u32x4_t v[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = threadIdx.x + NTHR * i, rl = c / CPR, ch = c - rl * CPR;
    v[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (CH % NTHR == 0 || c < CH) v[i] = gld16(src, (unsigned)((row0 + rl) * (int)ld + col0 + ch * 8) * 2u);
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = threadIdx.x + NTHR * i, rl = c / CPR, ch = c - rl * CPR;
    if (CH % NTHR == 0 || c < CH) *reinterpret_cast<u32x4_t*>(&As[rl * APITCH + ch * 8]) = v[i];
  }
}

// bias chunks of the columns a thread meets in every epilogue round (issue the loads BEFORE the GEMM pass)
template <int NCH, int IT>
__device__ __forceinline__ void load_bias(const float* bias, float (&bs)[IT][8]) {
#pragma unroll
  for (int i = 0; i < IT; ++i) load8f(bias + ((threadIdx.x + NTHR * i) % NCH) * 8, bs[i]);
}

// Accumulators -> fp32 staging (32 rows per round) -> per-chunk elementwise functor on 8 consecutive features of one
// token (all global traffic 16-byte, row contiguous) -> bf16 back into the LDS tile (the next GEMM's operand).
//   pre(row, col)          -> P      issue the chunk's global loads (one round ahead: they fly during the previous round)
//   fn(row, col, v[8], P)            elementwise math + global stores; leaves the tile values in v
// No row bounds checks: every tensor the chains touch is allocated in whole 128-row tiles (api.hip: layout_saved), rows
// past T hold don't-care values that stay in their own rows.  The three chunk bodies of a round are branch free, so the
// scheduler interleaves them (the epilogues are latency/issue bound: 2 waves per SIMD).
// The round loop is NOT unrolled (only the accumulator -> staging copy depends on the round, through a switch): unrolled,
// the kernel was 190 KB of straight-line code, three times the instruction cache two CUs share.
template <int RF, int NCG, typename PreF, typename Fn>
__device__ __forceinline__ void epilogue(f32x4_t (&acc)[RF * NCG / 8][3], float* Stg, bf16_t* As, int row0, const float* bias_lds,
                                         PreF pre, Fn fn, bool to_lds, unsigned long long* prof = nullptr) {
#ifdef FZ_PROFILE_EPI  // sub-phase cycle counters of one epilogue (thread 0 of block 0), tools/fused_stamps.py --epi
  unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pacc[6] = {0, 0, 0, 0, 0, 0};
  const bool profiling = prof && blockIdx.x == 0 && threadIdx.x == 0;
#define FZ_PT(k) do { if (prof) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pt[k] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define FZ_PT(k) do { } while (0)
#endif
  // NCG = 8: 384 output columns, wave = column group, all rows; a round = 32 rows x 384 columns.
  // NCG = 4: 192 output columns, wave = 4 * row half + column group; a round = 64 rows x 192 columns, staged as two
  //          32-row groups side by side (staging columns 0..191 and 192..383), so the tile pass is the same 3 chunks/thread.
  constexpr int RR = RF >= 2 ? 32 : 16, FR = RR / 16, RPR = (NCG == 8 ? RR : 2 * RR), ROUNDS = 16 * RF / RPR, CH = RR * 48, IT = CH / NTHR;
  static_assert(CH % NTHR == 0 && IT == 3 && ROUNDS <= 4 && (NCG == 8 || (NCG == 4 && ((ROUNDS == 2 && RF == 8) || (ROUNDS == 1 && RF == 4)))),
                "tile pass geometry");
  // launder the thread index: the per-thread row / column offsets below are a handful of VALU instructions; shared (CSE'd) between the
  // ten epilogues of a chain kernel they stay live from the first to the last one — in scratch, across the GEMM passes
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  const int cg = wave % NCG, rh = wave / NCG;
  int rl[IT], col[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = tid + NTHR * i, r_ = c / 48, ch = c - r_ * 48;
    if constexpr (NCG == 8) { rl[i] = r_; col[i] = ch * 8; }
    else { rl[i] = r_ + (ch >= 24 ? 32 : 0); col[i] = (ch >= 24 ? ch - 24 : ch) * 8; }
  }
  // the thread's three bias chunks (the chain's bias vectors live in LDS): read once per epilogue, not once per round
  float bs[IT][8];
  if (bias_lds) {
#pragma unroll
    for (int i = 0; i < IT; ++i) load8f(bias_lds + col[i], bs[i]);
  }
  // ONE register set for the chunks' prefetched global loads, refilled at the END of a round for the next one (they fly
  // during the two barriers and the staging traffic).  The first version kept two sets selected by the round's parity:
  // hipcc merged every load into both sets with v_cndmask right behind it, i.e. s_waitcnt vmcnt(0) after each of the
  // three loads of a round — no prefetch at all, and each wait also drained the previous round's stores.
  using P = decltype(pre(0, 0));
  P pv[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) pv[i] = pre(row0 + rl[i], col[i]);
  float* sw = &Stg[(lane & 15) * SPITCH + cg * 48 + (lane >> 4) * 4];
#pragma unroll 1
  for (int r = 0; r < ROUNDS; ++r) {
    FZ_PT(0);
    lds_barrier();  // staging free again; (r == 0) every wave is done reading the tile in the GEMM
    FZ_PT(1);
#ifndef FZ_NO_STAGING  // (measurement build: the accumulators never go to the staging buffer — the chunk bodies read whatever is there)
    if constexpr (NCG == 8) {
      auto put = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        if constexpr (R < ROUNDS) {
#pragma unroll
          for (int a2 = 0; a2 < FR; ++a2)
#pragma unroll
            for (int b = 0; b < 3; ++b) *reinterpret_cast<f32x4_t*>(sw + a2 * 16 * SPITCH + b * 16) = acc[R * FR + a2][b];
          // keep the stores inside their case: hipcc otherwise turns the switch into a register select (48 v_mov_b64 per
          // round to funnel the round's accumulators into one set of store operands) followed by one copy of the stores
          asm volatile("" ::: "memory");
        }
      };
      switch (r) {
        case 0: put(std::integral_constant<int, 0>{}); break;
        case 1: put(std::integral_constant<int, 1>{}); break;
        case 2: put(std::integral_constant<int, 2>{}); break;
        default: put(std::integral_constant<int, 3>{}); break;
      }
    } else if constexpr (RF == 8) {
      if (rh == r) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) *reinterpret_cast<f32x4_t*>(sw + (a & 1) * 16 * SPITCH + (a >> 1) * 192 + b * 16) = acc[a][b];
      }
    } else {  // RF == 4: one round, both 32-row halves side by side (row half rh at staging columns 192 rh)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) *reinterpret_cast<f32x4_t*>(sw + a * 16 * SPITCH + rh * 192 + b * 16) = acc[a][b];
    }
#else
    if (r == 0) {  // (the accumulators stay live: the GEMM pass is not dead code)
#pragma unroll
      for (int a = 0; a < RF * NCG / 8; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) asm volatile("" :: "v"(acc[a][b]));
    }
#endif
    const int rbase = row0 + r * RPR;
    FZ_PT(2);
    lds_barrier();
    FZ_PT(3);
    float v[IT][8];
#pragma unroll
    for (int i = 0; i < IT; ++i) load8f(&Stg[(rl[i] & 31) * SPITCH + col[i] + (rl[i] >= 32 ? 192 : 0)], v[i]);
    FZ_PT(4);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      if (bias_lds) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] += bs[i][j];
      }
      fn(rbase + rl[i], col[i], v[i], pv[i], i);
      // one chunk body at a time (8 independent elements give the VALU enough ILP): letting the scheduler interleave the
      // three bodies triples their temporaries while the accumulators of the later rounds are still live -> spills
      __builtin_amdgcn_sched_barrier(0);
    }
    FZ_PT(5);
    if (to_lds) {
#pragma unroll
      for (int i = 0; i < IT; ++i) *reinterpret_cast<u32x4_t*>(&As[(r * RPR + rl[i]) * APITCH + col[i]]) = pack8(v[i]);
    }
    if (r + 1 < ROUNDS) {
#pragma unroll
      for (int i = 0; i < IT; ++i) pv[i] = pre(rbase + RPR + rl[i], col[i]);
    }
    FZ_PT(6);
#ifdef FZ_PROFILE_EPI
    if (prof) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pacc[k] += pt[k + 1] - pt[k];
    }
#endif
  }
  lds_barrier();
#ifdef FZ_PROFILE_EPI
  if (profiling) {
#pragma unroll
    for (int k = 0; k < 6; ++k) prof[k] = pacc[k];
  }
#endif
#undef FZ_PT
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

// Dropout site with the device base seed folded in and the site key derived ONCE per kernel.  Reading *seed_ptr where the
// mask is drawn put a global load + s_waitcnt vmcnt(0) into every chunk body of the epilogues (the compiler cannot hoist
// it across the chunk's global stores; vmcnt(0) also waits for all of the previous chunk's stores), and deriving the key
// there cost ~55 scalar instructions per chunk.
struct DropK { unsigned key = 0, thr = 0; float inv_keep = 1.f; };
__device__ __forceinline__ DropK resolve_drop(const DropCfg& d, unsigned long long base) {
  DropK k; k.key = drop_key(d.seed + base, d.site); k.thr = d.thr; k.inv_keep = d.inv_keep; return k;
}

// dropout mask of one attention probability: the map of attention.hip (attn_drop), so that this file's in-tile attention and
// the attention kernels draw the same masks for the same (seed, site, sequence, head, query, key)
__device__ __forceinline__ float attn_drop_f(unsigned key, unsigned row, int k, unsigned lk_half, unsigned thr, float inv_keep) {
  return attn_drop_scale(key, row, k, lk_half, thr, inv_keep);
}

template <bool DROP>
__device__ __forceinline__ void apply_drop(const DropK& d, unsigned long long idx0, float (&v)[8]) {
#ifdef FZ_NO_EPI_MATH  // (measurement build: no dropout hash, identity instead of GELU — the epilogues keep their staging, loads and stores)
  return;
#endif
  if constexpr (DROP) {
    float sc[8];
    drop_scales_key<8>(d.key, idx0, d.thr, d.inv_keep, sc);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= sc[j];
  }
}

// Backward of a dropout site whose POST-dropout value the forward saved and the backward loads anyway (h1, hp: the pre-GELU values):
// an element was dropped iff the saved value is zero, so the mask needs no hash — dy * (saved != 0 ? 1 / keep : 0).  (A kept element
// that is exactly 0.0 after rounding — |x| < 1e-40 in bf16 — reads as dropped: its gradient goes to zero instead of dy GELU'(0) / keep;
// the forward's masks themselves are unchanged.)  The backward chain hashed 1 920 elements per token for its masks, 1 152 of them here.
template <bool DROP>
__device__ __forceinline__ void mask_from_saved(const DropK& d, const float (&saved)[8], float (&v)[8]) {
  if constexpr (DROP) {
#ifdef FZ_NO_EPI_MATH
    return;
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = saved[j] != 0.0f ? v[j] * d.inv_keep : 0.0f;
  }
}

// COOT LayerNorm (nntrainer/models/normalizations.py:98-101) of every tile row, in place.  Wave w owns rows
// [2 RF w, 2 RF w + 2 RF); 16 lanes share a row (4 rows per wave in flight): lane j holds the three 8-element chunks at
// columns 8 j, 128 + 8 j, 256 + 8 j (16-byte LDS / global accesses), reductions are 4 DPP steps.
template <int RF, bool DROP, bool OUT32, bool MASK = false>
__device__ __forceinline__ void ln_tile(bf16_t* As, const float* gain, const float* bias, int row0, int T, bf16_t* out, float* out32,
                                        long ld32, const DropK& drop) {  // MASK: rows >= T are not stored to `out` either (tiles that do not own whole 32-row blocks)
#ifdef FZ_NO_LN  // (measurement build: no LayerNorm on the tile)
  return;
#endif
  asm volatile("" : "+s"(row0));  // see ln_bwd_tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j16 = lane & 15, g = lane >> 4;
  constexpr int RW = 2 * RF, ITR = (RW + 3) / 4;
  float gn[3][8], bi[3][8];
#pragma unroll
  for (int m = 0; m < 3; ++m) { load8f(gain + m * 128 + j16 * 8, gn[m]); load8f(bias + m * 128 + j16 * 8, bi[m]); }
#pragma unroll
  for (int it = 0; it < ITR; ++it) {
    const int rw = (RW >= 4) ? it * 4 + g : (g < RW ? g : 0);  // RF = 1: two rows per wave, lanes 32..63 repeat row 0
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
      if constexpr (DROP) {
        float sc[8];
        drop_scales_key<8>(drop.key, (unsigned long long)row * FZ_D + m * 128 + j16 * 8, drop.thr, drop.inv_keep, sc);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] *= sc[e];
      }
      const u32x4_t o = pack8(y);
      *reinterpret_cast<u32x4_t*>(ar + m * 128) = o;
      if (!MASK || row < T) gst16(out, (unsigned)(row * FZ_D + m * 128 + j16 * 8) * 2u, o);
      if constexpr (OUT32) {
        if (row < T) {
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

// ---- the layer's self-attention on the tile's own rows (fused.h: FusedAttn) ---------------------------------------------------------
// Wave h computes HEAD h for every 16-query fragment of the tile.  Sequence by sequence (a tile touches <= 3): the wave stages K_h | V_h
// of the sequence's L rows into ITS OWN slice of the workgroup's LDS (the whole 160 KB is free here: the tile receives the result only
// at the end) — wave-private, so the phase needs no workgroup barrier and every wave runs its own stream of loads, MFMAs and softmax —
// then runs the fragments of that sequence exactly as attn_short_fwd_kernel does for one (sequence, head): S^T = K Q^T on the
// 16 x 16 x 16 MFMA (a lane owns one query and 4 keys per tile), single-pass softmax, O^T = V^T P^T with V^T from the LDS transpose read.
// The bf16 results wait in registers (6 per fragment); after one barrier they go into the LDS tile (columns 48 h ..), lse to global.
// (Round 6's first version — wave = fragment, one head per round, K / V of all the tile's sequences shared through LDS, two barriers per
// head — cost +37 us per 128-row launch: profiles/r06_ab_fused_attn.txt.)
template <int RF>
__device__ __forceinline__ void fused_self_attn(const PostAttnFwd& p, bf16_t* As, unsigned char* smem_base, int row0, unsigned long long sbase) {
  constexpr int DH = 48, RP = FZ_ATTN_RP(RF), NK = DH / 16, H = 8, CPR = DH / 8, LDQ = 3 * FZ_D, LMAX = FZ_ATTN_LMAX(RF);
  const FusedAttn& A = p.attn;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, h = tid >> 6, lq = lane & 15, lg = lane >> 4;
  // the tile's segment (a tile never straddles the two: launch_post_attn_fwd checks it)
  const int T0 = A.N0 * A.L0;
  const bool second = row0 >= T0;
  const int L = second ? A.L1 : A.L0, segbase = second ? T0 : 0, segend = second ? p.T : T0;
  const long long* lens = second ? A.lens1 : A.lens0;
  const int last = (row0 + 16 * RF < segend ? row0 + 16 * RF : segend) - 1;
  const int s_first = (row0 - segbase) / L, s_last = (last - segbase) / L;
  const int nkt = L >> 4;
  const unsigned dkey = A.drop.thr ? drop_key(A.drop.seed + (second ? A.seed2_delta : 0ull) + sbase, A.drop.site) : 0u;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_base) + h * (2 * LMAX * RP);  // this wave's slice
  bf16_t* Vs = Ks + LMAX * RP;
  u32x2_t out[RF][NK];
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int dt = 0; dt < NK; ++dt) out[f][dt] = u32x2_t{0u, 0u};
  const int half = L * CPR;  // chunks of K (then as many of V)
  // this head's query fragments of the whole tile, requested up front: they land during the first sequence's staging round trip
  // (64-row tiles only: on 128-row tiles the 48 extra registers spill — there a fragment's q is loaded when its turn comes)
  constexpr bool QALL = RF <= 4;
  s16x4_t qall[QALL ? RF : 1][NK];
  if constexpr (QALL) {
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        qall[f][ks] = s16x4_t{0, 0, 0, 0};
        if (row0 + 16 * f < segend) qall[f][ks] = *reinterpret_cast<const s16x4_t*>(A.qkv + (long)(row0 + 16 * f + lq) * LDQ + h * DH + ks * 16 + lg * 4);
      }
  }
  constexpr int NH = (LMAX * CPR + 63) / 64;  // 16-byte staging chunks per lane and matrix
  u32x4_t sk[NH], sv[NH];
  auto ld = [&](int kb0, int kv, u32x4_t (&st)[NH]) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int c = lane + 64 * i, r = c / CPR, ch = c - r * CPR;
      st[i] = u32x4_t{0u, 0u, 0u, 0u};
      if (c < half) st[i] = gld16(A.qkv, (unsigned)((kb0 + r) * LDQ + FZ_D * (1 + kv) + h * DH + ch * 8) * 2u);
    }
  };
  auto sto = [&](bf16_t* dst, const u32x4_t (&st)[NH]) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int c = lane + 64 * i, r = c / CPR, ch = c - r * CPR;
      if (c < half) *reinterpret_cast<u32x4_t*>(&dst[r * RP + ch * 8]) = st[i];
    }
  };
  // 64-row tiles request a sequence's rows one sequence ahead (they fly while the previous one's fragments are computed); on 128-row
  // tiles the 64 registers that would stay live across the compute spill (measured: the gain is gone) — there the rows are loaded
  // when the sequence's turn comes
  constexpr bool KVPRE = RF <= 4;
  if constexpr (KVPRE) { ld(segbase + s_first * L, 0, sk); ld(segbase + s_first * L, 1, sv); }
#pragma unroll 1
  for (int sq = s_first; sq <= s_last; ++sq) {
    const int kbase = segbase + sq * L;  // first row of the sequence
    if constexpr (!KVPRE) { ld(kbase, 0, sk); ld(kbase, 1, sv); }
    // K_h and V_h of the sequence -> the wave's LDS slice
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's fragment reads of the previous sequence are done)
    sto(Ks, sk);
    sto(Vs, sv);
    if constexpr (KVPRE) { if (sq < s_last) { ld(kbase + L, 0, sk); ld(kbase + L, 1, sv); } }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private data: DS operations of a wave execute in order, no barrier
    const int nvalid = (int)lens[sq];
    // the tile's fragments that lie in this sequence
    const int f_lo = kbase > row0 ? (kbase - row0) >> 4 : 0;
    const int f_hi_row = (kbase + L < row0 + 16 * RF ? kbase + L : row0 + 16 * RF);
    const int f_hi = ((f_hi_row < segend ? f_hi_row : segend) - row0) >> 4;  // exclusive
    // 128-row tiles: the query fragment one ahead is in flight while a fragment is computed (rows past the segment are allocated:
    // the token matrices come in whole 128-row tiles, api.hip: layout_saved)
    s16x4_t qnext[NK];
    auto qload = [&](int f, s16x4_t (&q)[NK]) {
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) q[ks] = *reinterpret_cast<const s16x4_t*>(A.qkv + (long)(row0 + 16 * f + lq) * LDQ + h * DH + ks * 16 + lg * 4);
    };
    if constexpr (!QALL) { if (f_lo < f_hi) qload(f_lo, qnext); }
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      if (f >= f_lo && f < f_hi) {
        const int g0 = row0 + 16 * f, qpos = g0 - kbase + lq;
        s16x4_t qf[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          if constexpr (QALL) qf[ks] = qall[f][ks];
          else qf[ks] = qnext[ks];
        }
        if constexpr (!QALL) { if (f + 1 < f_hi) qload(f + 1, qnext); }
        f32x4_t sc[LMAX / 16];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < LMAX / 16; ++kt) {
          if (kt < nkt) {
            sc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
              const s16x4_t kf = *reinterpret_cast<const s16x4_t*>(&Ks[(kt * 16 + lq) * RP + ks * 16 + lg * 4]);
              sc[kt] = COOT_MFMA_16x16x16(kf, qf[ks], sc[kt]);  // sc[i] = S^T[key kt * 16 + lg * 4 + i][query lq]
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float v = sc[kt][i] * A.scale;
              if (kt * 16 + lg * 4 + i >= nvalid) v = kMaskFill;  // masked_fill(mask, -INF) (transformer_legacy.py:544)
              sc[kt][i] = v;
              mx = fmaxf(mx, v);
            }
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        s16x4_t pf[LMAX / 16];
#pragma unroll
        for (int kt = 0; kt < LMAX / 16; ++kt) {
          if (kt < nkt) {
            float pr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              pr[i] = __expf(sc[kt][i] - mx);
              sum += pr[i];
              if (A.drop.thr)
                pr[i] *= attn_drop_f(dkey, (unsigned)((sq * H + h) * L + qpos), kt * 16 + lg * 4 + i, (unsigned)(L + 1) >> 1, A.drop.thr, A.drop.inv_keep);
            }
            const unsigned lo = pack2bf(pr[0], pr[1]), hi = pack2bf(pr[2], pr[3]);
            pf[kt] = s16x4_t{(short)(lo & 0xFFFF), (short)(lo >> 16), (short)(hi & 0xFFFF), (short)(hi >> 16)};
          }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int dt = 0; dt < NK; ++dt) {
          f32x4_t o = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < LMAX / 16; ++kt) {
            if (kt < nkt) {
              // A = V^T [dim][key]: LDS transpose read of the 16 x 16 block (keys 16 kt .., dims 16 dt ..)
              const bf16_t* addr = &Vs[(kt * 16) * RP + dt * 16] + (4 * lg + (lq >> 2)) * RP + (lq & 3) * 4;
              const s16x4_t vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(addr));
              o = COOT_MFMA_16x16x16(vt, pf[kt], o);
            }
          }
          out[f][dt] = u32x2_t{pack2bf(o[0] * inv, o[1] * inv), pack2bf(o[2] * inv, o[3] * inv)};  // O^T[dim 16 dt + 4 lg + i][query lq]
        }
        if (lg == 0) A.lse[(long)(g0 + lq) * H + h] = mx + __logf(sum);
      }
    }
  }
  __syncthreads();  // every wave is done with its staging slice (the slices overlap the tile)
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int dt = 0; dt < NK; ++dt) *reinterpret_cast<u32x2_t*>(&As[(16 * f + lq) * APITCH + h * DH + dt * 16 + lg * 4]) = out[f][dt];
  lds_barrier();
  // ctx for the backward pass (attention backward, out-projection weight gradient): 16-byte row-contiguous stores from the tile
  for (int c = tid; c < 16 * RF * 48; c += NTHR) {
    const int r = c / 48, ch = c - r * 48;
    if (row0 + r < p.T) gst16(p.ctx_w, (unsigned)((row0 + r) * FZ_D + ch * 8) * 2u, *reinterpret_cast<const u32x4_t*>(&As[r * APITCH + ch * 8]));
  }
}

template <int RF, bool DROP>
__global__ __launch_bounds__(512) void post_attn_fwd_kernel(PostAttnFwd p) {
  constexpr int BT = 16 * RF, RR = RF >= 2 ? 32 : 16;
  // LDS: token tile | fp32 staging | 7 parameter vectors of 384 floats (bo, b1, b2, ln1 gain/bias, ln2 gain/bias; the
  // first three slots are re-used for the pooling biases once the encoder layer is done) = 162,816 B of the 160 KiB
  constexpr int SMEM_CHAIN = BT * APITCH * 2 + RR * SPITCH * 4 + 7 * FZ_D * 4;
  constexpr int SMEM = SMEM_CHAIN > FZ_ATTN_SMEM(RF) ? SMEM_CHAIN : FZ_ATTN_SMEM(RF);  // (64-row tiles: the in-chain attention's slices need 3 KB more)
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  float* Stg = reinterpret_cast<float*>(smem + BT * APITCH * 2);
  float* Bsm = reinterpret_cast<float*>(smem + BT * APITCH * 2 + RR * SPITCH * 4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile_idx = fz_tile_index((p.T + BT - 1) / BT);
  if (tile_idx < 0) return;
  const int row0 = tile_idx * BT, T = p.T;
#ifdef FZ_TILE_LOG  // measurement build: where (XCD, shader engine, CU) every tile ran and how long it took (tools/tile_log.py)
  const unsigned long long fz_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  f32x4_t acc[RF][3];
  auto load_vectors = [&]() {
    const float* vecs[7] = {p.bo, p.b1, p.b2, p.ln1g, p.ln1b, p.ln2g, p.ln2b};
#pragma unroll
    for (int k = 0; k < 7; ++k)
      if (threadIdx.x < FZ_D / 4) reinterpret_cast<f32x4_t*>(Bsm + k * FZ_D)[threadIdx.x] = reinterpret_cast<const f32x4_t*>(vecs[k])[threadIdx.x];
  };
  bool fuse_attn = false;
  if constexpr (RF >= 4) fuse_attn = p.attn.on != 0;
  if (!fuse_attn) load_vectors();  // (with the attention inside, the vector area is part of its K / V staging first)
  int tsn = 0;
  // (the pointer is re-read from its SGPR pair at every stamp: as a VGPR address hoisted to the kernel's entry it was the last spill)
  auto stamp = [&]() {
    unsigned long long* ts = p.tstamps;
    asm volatile("" : "+s"(ts));
    if (ts && blockIdx.x == 0 && threadIdx.x == 0) ts[tsn] = __builtin_amdgcn_s_memtime();
    ++tsn;
  };
  unsigned long long sbase = 0;
  if constexpr (DROP) { if (p.d_ff1.seed_ptr) sbase = *p.d_ff1.seed_ptr; }
  const DropK d_postln = resolve_drop(p.d_postln, sbase), d_ff1 = resolve_drop(p.d_ff1, sbase), d_ff2 = resolve_drop(p.d_ff2, sbase),
                d_pool1 = resolve_drop(p.d_pool1, sbase), d_pool2 = resolve_drop(p.d_pool2, sbase);
  stamp();

  // ---- attention output projection + residual -> r1 ------------------------------------------------------------
  TileRing<RF> R;  // weight ring, carried from pass to pass (gemm_run)
  if constexpr (RF >= 4) {
    if (fuse_attn) {
      fused_self_attn<RF>(p, As, smem, row0, sbase);
      gemm_issue(R, p.wo + wave * GSZ, lane);
      load_vectors();
    } else {
      gemm_issue(R, p.wo + wave * GSZ, lane);  // (flies during the tile load)
      load_tile<RF>(As, p.ctx, FZ_D, 0, row0);
    }
  } else {
    gemm_issue(R, p.wo + wave * GSZ, lane);
    load_tile<RF>(As, p.ctx, FZ_D, 0, row0);
  }
  __syncthreads();
  stamp();
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, p.wo + wave * GSZ, p.w1 + wave * GSZ, acc, lane);
  stamp();
  epilogue<RF, 8>(acc, Stg, As, row0, Bsm + 0 * FZ_D,
      [&](int row, int col) { return PreRes{gld16(p.xres, (unsigned)(row * FZ_D + col) * 2u)}; },
      [&](int row, int col, float (&v)[8], const PreRes& pr, int) {
        float r[8];
        unpack8(pr.res, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        gst16(p.r1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, true);
  stamp();
  // ---- LN1 (+ dropout) -> z1 ---------------------------------------------------------------------------------------
  ln_tile<RF, DROP, false>(As, Bsm + 3 * FZ_D, Bsm + 4 * FZ_D, row0, T, p.z1, nullptr, 0, d_postln);
  __syncthreads();
  stamp();
  // ---- FF1: Linear -> Dropout -> GELU ------------------------------------------------------------------------------
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, p.w1 + wave * GSZ, p.w2 + wave * GSZ, acc, lane);
  stamp();
  epilogue<RF, 8>(acc, Stg, As, row0, Bsm + 1 * FZ_D, [&](int, int) { return PreNone{}; },
      [&](int row, int col, float (&v)[8], const PreNone&, int) {
        apply_drop<DROP>(d_ff1, (unsigned long long)row * FZ_D + col, v);
        gst16(p.h1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
        gst16(p.a1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, true
#ifdef FZ_PROFILE_EPI
      , p.tstamps ? p.tstamps + 32 : nullptr
#endif
      );
  stamp();
  // ---- FF2: Linear -> Dropout, + residual z1 -> r2 -----------------------------------------------------------------
  zero_acc<RF>(acc);
  {  // what follows FF2: the first pooling pass (or nothing: the ring then re-loads this pass's own first k-blocks, harmlessly)
    const bf16_t* nx = (RF >= 4 && p.do_pool) ? p.pw1 + wave * GSZ : p.w2 + wave * GSZ;
    gemm_run<RF, 12, true>(R, As, p.w2 + wave * GSZ, nx, acc, lane);
  }
  epilogue<RF, 8>(acc, Stg, As, row0, Bsm + 2 * FZ_D,
      [&](int row, int col) { return PreRes{gld16(p.z1, (unsigned)(row * FZ_D + col) * 2u)}; },
      [&](int row, int col, float (&v)[8], const PreRes& pr, int) {
        float r[8];
        unpack8(pr.res, r);
        apply_drop<DROP>(d_ff2, (unsigned long long)row * FZ_D + col, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        gst16(p.r2, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, true);
  stamp();
  // ---- LN2 -> z2 ------------------------------------------------------------------------------------------------------
  {
    const DropK none;
    if (p.z2_f32) ln_tile<RF, false, true>(As, Bsm + 5 * FZ_D, Bsm + 6 * FZ_D, row0, T, p.z2, p.z2_f32, p.ldz2_f32, none);
    else ln_tile<RF, false, false>(As, Bsm + 5 * FZ_D, Bsm + 6 * FZ_D, row0, T, p.z2, nullptr, 0, none);
  }
  __syncthreads();
  stamp();
  if constexpr (RF >= 4) {
    if (!p.do_pool) return;
    // pooling biases into the (now dead) bias slots 0..2: pb1 [768] | pb2 [384]
    if (threadIdx.x < 2 * FZ_D / 4) reinterpret_cast<f32x4_t*>(Bsm)[threadIdx.x] = reinterpret_cast<const f32x4_t*>(p.pb1)[threadIdx.x];
    else if (threadIdx.x < 3 * FZ_D / 4) reinterpret_cast<f32x4_t*>(Bsm)[threadIdx.x] = reinterpret_cast<const f32x4_t*>(p.pb2)[threadIdx.x - 2 * FZ_D / 4];
    __syncthreads();
    // ---- GenPool scores (poolers.py:171-181): per head h, a = GELU(dropout(z W1[h] + b1[h])), s = dropout(a W2[h] + b2[h]) ----
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      if (h == 1) {  // the tile holds a_0 now: bring z2 back (just written, L2 resident)
        load_tile<RF>(As, p.z2, FZ_D, 0, row0);
        __syncthreads();
      }
      zero_acc<RF>(acc);
      const bf16_t* w2h = p.pw2 + (h * 4 + (wave & 3)) * GSZ;
      gemm_run<RF, 12, true>(R, As, p.pw1 + (h * 8 + wave) * GSZ, w2h, acc, lane);
      epilogue<RF, 8>(acc, Stg, As, row0, Bsm + h * FZ_D, [&](int, int) { return PreNone{}; },
          [&](int row, int col, float (&v)[8], const PreNone&, int) {
            apply_drop<DROP>(d_pool1, (unsigned long long)row * (2 * FZ_D) + h * FZ_D + col, v);
            gst16(p.hp, (unsigned)(row * (2 * FZ_D) + h * FZ_D + col) * 2u, pack8(v));
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
            gst16(p.ap, (unsigned)(row * (2 * FZ_D) + h * FZ_D + col) * 2u, pack8(v));
          }, true);
      // FC2 of head h: 192 output columns = 4 column groups x 2 row halves
      f32x4_t acc2[RF / 2][3];
      zero_acc<RF / 2>(acc2);
      gemm_run<RF / 2, 12, true>(R, As + (wave >> 2) * (BT / 2) * APITCH, w2h, h == 0 ? p.pw1 + (8 + wave) * GSZ : w2h, acc2, lane);
      epilogue<RF, 4>(acc2, Stg, As, row0, Bsm + 2 * FZ_D + h * (FZ_D / 2), [&](int, int) { return PreNone{}; },
          [&](int row, int col, float (&v)[8], const PreNone&, int) {
            apply_drop<DROP>(d_pool2, (unsigned long long)row * FZ_D + h * (FZ_D / 2) + col, v);
            gst16(p.s, (unsigned)(row * FZ_D + h * (FZ_D / 2) + col) * 2u, pack8(v));
          }, false);
      stamp();
    }
  }
#ifdef FZ_TILE_LOG
  {
    unsigned long long* ts = p.tstamps;
    asm volatile("" : "+s"(ts));
    if (ts && threadIdx.x == 0) {
      unsigned long long* e = ts + 128 + 4 * tile_idx;
      e[0] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
      e[1] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
      e[2] = fz_t0; e[3] = __builtin_amdgcn_s_memrealtime();
    }
  }
#endif
}

// ---- backward chain --------------------------------------------------------------------------------------------------

// LayerNorm backward of every tile row, in place (dy -> dx), same thread mapping as ln_tile:
//   xc = x - mean, s = std + eps, h = dy * gain;  dx = (h - mean(h)) / s - (sum h xc) / s^2 * xc / ((n - 1) std)
// x = the saved LN input (global, bf16).  DROPY: dy is first multiplied by the dropout mask the forward applied to the LN
// output.  MASKX: dxm = dx * mask(drop_x) is what stays in the tile and goes to out_dxm (dx itself goes to out_dx).
// Column sums over the tile's valid rows (dgain = dy xc / s, dbias = dy, colsum of dxm or dx): lanes -> wave by two
// plain LDS stores into red[32][3 * 128] per 128-column chunk (LDS float atomics serialise per lane and cost ~1.5k cycles
// per wave-instruction here), summed by 384 threads into part_dst[3 * 384] (one partial row per tile).
template <int RF, bool DROPY, bool MASKX>
__device__ __forceinline__ void ln_bwd_tile(bf16_t* As, const float* gain_lds, const bf16_t* xsaved, int row0, int T, bf16_t* out_dx,
                                            bf16_t* out_dxm, const DropK& dy_drop, const DropK& dx_drop, float* red, float* part_dst) {
  static_assert(RF == 8 || RF == 4, "2 RF rows per wave, 4 per pass");
  constexpr int NIT = RF / 2, RW = 2 * RF;
  asm volatile("" : "+s"(row0));  // keep this call's per-thread offsets out of the other LayerNorm's live range (they were spilled across the chain)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));   // ... and everything derived from the thread index (see epilogue())
  const int lane = tid & 63, wave = tid >> 6, j16 = lane & 15, g = lane >> 4;
  u32x4_t xs[NIT][3];  // the saved LN input of this lane's rows, kept packed (all loads in flight together)
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int m = 0; m < 3; ++m) xs[it][m] = gld16(xsaved, (unsigned)((row0 + wave * RW + it * 4 + g) * FZ_D + m * 128 + j16 * 8) * 2u);
  // pass A: row statistics (4 scalars per row)
  float mean[NIT], rsv[NIT], hmean[NIT], k2[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int rl = wave * RW + it * 4 + g, row = row0 + rl;
    bf16_t* ar = As + rl * APITCH + j16 * 8;
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      float x[8];
      unpack8(xs[it][m], x);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[e];
    }
    const float mu = row16_sum(s) * (1.0f / 384.0f);
    float q = 0.f, hs = 0.f, hx = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      float x[8], dy[8], gn[8];
      unpack8(xs[it][m], x);
      unpack8(*reinterpret_cast<const u32x4_t*>(ar + m * 128), dy);
      load8f(gain_lds + m * 128 + j16 * 8, gn);
      if constexpr (DROPY) {
        // the masked gradient goes back into the tile (this lane's own chunk, bf16 like every gradient of the chain) and pass B reads
        // it from there: drawing the mask a second time in pass B cost 12 hash evaluations per lane and the registers that pushed
        // the row statistics into scratch
        float sc[8];
        drop_scales_key<8>(dy_drop.key, (unsigned long long)row * FZ_D + m * 128 + j16 * 8, dy_drop.thr, dy_drop.inv_keep, sc);
#pragma unroll
        for (int e = 0; e < 8; ++e) dy[e] *= sc[e];
        const u32x4_t md = pack8(dy);
        *reinterpret_cast<u32x4_t*>(ar + m * 128) = md;
        unpack8(md, dy);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xc = x[e] - mu;
        q += xc * xc;
        const float h = dy[e] * gn[e];
        hs += h; hx += h * xc;
      }
    }
    q = row16_sum(q); hs = row16_sum(hs); hx = row16_sum(hx);
    const float stdv = sqrtf(q * (1.0f / 383.0f));
    const float rs = 1.0f / (stdv + kLnEps);
    mean[it] = mu; rsv[it] = rs; hmean[it] = hs * (1.0f / 384.0f);
    k2[it] = stdv > 0.f ? hx * rs * rs / (383.0f * stdv) : 0.f;
  }
  // pass B: one 8-column chunk at a time (24 column-sum accumulators live instead of 72)
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    float gn[8], csg[8], csb[8], csx[8];
    load8f(gain_lds + m * 128 + j16 * 8, gn);
#pragma unroll
    for (int e = 0; e < 8; ++e) { csg[e] = 0.f; csb[e] = 0.f; csx[e] = 0.f; }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int rl = wave * RW + it * 4 + g, row = row0 + rl;
      bf16_t* ar = As + rl * APITCH + j16 * 8 + m * 128;
      float x[8], dy[8], dx[8], dm[8];
      unpack8(xs[it][m], x);
      unpack8(*reinterpret_cast<const u32x4_t*>(ar), dy);  // (DROPY: masked in pass A)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        x[e] -= mean[it];
        dx[e] = (dy[e] * gn[e] - hmean[it]) * rsv[it] - k2[it] * x[e];
      }
      if constexpr (MASKX) {
        float sc[8];
        drop_scales_key<8>(dx_drop.key, (unsigned long long)row * FZ_D + m * 128 + j16 * 8, dx_drop.thr, dx_drop.inv_keep, sc);
#pragma unroll
        for (int e = 0; e < 8; ++e) dm[e] = dx[e] * sc[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) dm[e] = dx[e];
      }
      if (row < T) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { csg[e] += dy[e] * x[e] * rsv[it]; csb[e] += dy[e]; csx[e] += dm[e]; }
      }
      const unsigned boff = (unsigned)(row * FZ_D + m * 128 + j16 * 8) * 2u;
      gst16(out_dx, boff, pack8(dx));
      const u32x4_t om = pack8(dm);
      if constexpr (MASKX) gst16(out_dxm, boff, om);
      *reinterpret_cast<u32x4_t*>(ar) = om;
    }
    // 32 (wave, row group) partial rows of this 128-column chunk -> LDS -> 384 threads add them up (the register route,
    // two xor-shuffles per value, is 144 ds_bpermute per wave and as slow as the LayerNorm itself)
    {
      float* rw = red + (wave * 4 + g) * (3 * 128) + j16 * 8;
      *reinterpret_cast<f32x4_t*>(rw) = f32x4_t{csg[0], csg[1], csg[2], csg[3]};
      *reinterpret_cast<f32x4_t*>(rw + 4) = f32x4_t{csg[4], csg[5], csg[6], csg[7]};
      *reinterpret_cast<f32x4_t*>(rw + 128) = f32x4_t{csb[0], csb[1], csb[2], csb[3]};
      *reinterpret_cast<f32x4_t*>(rw + 128 + 4) = f32x4_t{csb[4], csb[5], csb[6], csb[7]};
      *reinterpret_cast<f32x4_t*>(rw + 256) = f32x4_t{csx[0], csx[1], csx[2], csx[3]};
      *reinterpret_cast<f32x4_t*>(rw + 256 + 4) = f32x4_t{csx[4], csx[5], csx[6], csx[7]};
    }
    lds_barrier();
    if (tid < 3 * 128) {
      float v = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) v += red[r * (3 * 128) + tid];
      const int q = tid / 128, c = tid % 128;  // quantity (dgain, dbias, colsum dx), column within the chunk
      part_dst[q * FZ_D + m * 128 + c] = v;
    }
    lds_barrier();
  }
  __syncthreads();  // tile complete; the global stores (dx / dxm) are visible to the whole workgroup
}

// column sums a thread gathered over the epilogue rounds (its 3 fixed column chunks) -> one partial row of the tile.
// For chunk index i the threads t = ch + 48 k (k < 11) share column chunk (ch + 32 i) % 48: they park their sums in
// red[k][384] and 384 threads add the (up to) 11 rows.  `red` = the staging buffer (free between epilogues).
__device__ __forceinline__ void colsum_flush(const float (&cs)[3][8], float* red, float* dst) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // (see epilogue(): per-call thread offsets, not kernel-wide ones)
  const int k = tid / 48;
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int col = ((tid + NTHR * i) % 48) * 8;
    lds_barrier();
    *reinterpret_cast<f32x4_t*>(&red[k * FZ_D + col]) = f32x4_t{cs[i][0], cs[i][1], cs[i][2], cs[i][3]};
    *reinterpret_cast<f32x4_t*>(&red[k * FZ_D + col + 4]) = f32x4_t{cs[i][4], cs[i][5], cs[i][6], cs[i][7]};
    lds_barrier();
    if (tid < FZ_D) {
      // chunk c8 = tid / 8 is held (for this i) by the threads with (t + 32 i) % 48 == c8, i.e. t % 48 == (c8 - 32 i) mod 48
      const int ch = (tid / 8 + 48 * 2 - 32 * i) % 48;
      const int nk = (NTHR - ch + 47) / 48;  // threads ch, ch + 48, ... < 512
      for (int kk = 0; kk < nk; ++kk) tot += red[kk * FZ_D + tid];
    }
  }
  if (tid < FZ_D) dst[tid] = tot;
}

template <int RF, bool DROP>
__global__ __launch_bounds__(512) void pre_attn_bwd_kernel(PreAttnBwd p) {
  constexpr int BT = 16 * RF, RR = 32;
  constexpr long GSZ192 = 6L * 3 * 512;  // one 48-column weight group at K = 192
  // LDS: token tile | fp32 staging (also the LayerNorm column-sum buffer) | zero vector, ln2 gain, ln1 gain, column-sum buffer
  __shared__ __attribute__((aligned(16))) unsigned char smem[BT * APITCH * 2 + RR * SPITCH * 4 + 4 * FZ_D * 4];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  float* Stg = reinterpret_cast<float*>(smem + BT * APITCH * 2);
  float* Bsm = reinterpret_cast<float*>(smem + BT * APITCH * 2 + RR * SPITCH * 4);
  float* zero = Bsm, *g2 = Bsm + FZ_D, *g1 = Bsm + 2 * FZ_D, *cred = Bsm + 3 * FZ_D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * BT, T = p.T;
  float* part = p.part + (long)blockIdx.x * FZ_BWD_NCS;
  if (threadIdx.x < FZ_D / 4) {
    reinterpret_cast<f32x4_t*>(zero)[threadIdx.x] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    reinterpret_cast<f32x4_t*>(g2)[threadIdx.x] = reinterpret_cast<const f32x4_t*>(p.ln2g)[threadIdx.x];
    reinterpret_cast<f32x4_t*>(g1)[threadIdx.x] = reinterpret_cast<const f32x4_t*>(p.ln1g)[threadIdx.x];
  }
  f32x4_t acc[RF][3];
  float cs[3][8];
  int tsn = 0;
  // (the pointer is re-read from its SGPR pair at every stamp: as a VGPR address hoisted to the kernel's entry it was the last spill)
  auto stamp = [&]() {
    unsigned long long* ts = p.tstamps;
    asm volatile("" : "+s"(ts));
    if (ts && blockIdx.x == 0 && threadIdx.x == 0) ts[tsn] = __builtin_amdgcn_s_memtime();
    ++tsn;
  };
  unsigned long long sbase = 0;
  if constexpr (DROP) { if (p.d_ff1.seed_ptr) sbase = *p.d_ff1.seed_ptr; }
  const DropK d_postln = resolve_drop(p.d_postln, sbase), d_ff1 = resolve_drop(p.d_ff1, sbase), d_ff2 = resolve_drop(p.d_ff2, sbase),
                d_pool1 = resolve_drop(p.d_pool1, sbase);
  stamp();
  auto cs_zero = [&]() {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) cs[i][j] = 0.f;
  };

  TileRing<RF> R;  // weight ring, carried from pass to pass (gemm_run)
  // this wave's weight group of a P48 array, from a laundered group index: shared between the passes, `wave * GSZ` was the value hipcc
  // parked in scratch across the chain (one multiply-add to recompute)
  auto WG = [&](const bf16_t* base, int grp, long gsz) { asm volatile("" : "+v"(grp)); return base + grp * gsz; };
  if (p.do_pool) {
    // ---- GenPool score MLP backward: dhp_h = (ds_h . W2[h]^T) * GELU'(hp_h) * drop1;  dz = sum_h dhp_h . W1[h]^T + dzp ----
    gemm_issue(R, WG(p.pw2, wave, GSZ192), lane);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      load_tile<RF, 24>(As, p.ds, FZ_D, h * (FZ_D / 2), row0);
      __syncthreads();
      zero_acc<RF>(acc);
      gemm_run<RF, 6, true>(R, As, WG(p.pw2, h * 8 + wave, GSZ192), h == 0 ? WG(p.pw2, 8 + wave, GSZ192) : WG(p.pw1, 8 + wave, GSZ), acc, lane);
      cs_zero();
      epilogue<RF, 8>(acc, Stg, As, row0, nullptr,
          [&](int row, int col) { return PreRes{gld16(p.hp, (unsigned)(row * (2 * FZ_D) + h * FZ_D + col) * 2u)}; },
          [&](int row, int col, float (&v)[8], const PreRes& pr, int i) {
            float a[8];
            unpack8(pr.res, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(a[j]);
#ifdef FZ_HASH_BWD_MASKS  // (A/B build: the masks of these two sites re-drawn from the hash, as through round 5)
            apply_drop<DROP>(d_pool1, (unsigned long long)row * (2 * FZ_D) + h * FZ_D + col, v);
#else
            mask_from_saved<DROP>(d_pool1, a, v);  // hp = dropout(z W1 + b1): the mask is in the saved value
#endif
            if (row < T) {
#pragma unroll
              for (int j = 0; j < 8; ++j) cs[i][j] += v[j];
            }
            gst16(p.dhp, (unsigned)(row * (2 * FZ_D) + h * FZ_D + col) * 2u, pack8(v));
          }, true);
      colsum_flush(cs, Stg, part + h * FZ_D);  // pb1 gradient, head h
      stamp();
    }
    // the tile holds dhp_1 now
    zero_acc<RF>(acc);
    gemm_run<RF, 12, true>(R, As, WG(p.pw1, 8 + wave, GSZ), WG(p.pw1, wave, GSZ), acc, lane);
    __syncthreads();  // every wave is done with dhp_1; dhp_0 (stored by this workgroup above) is visible
    load_tile<RF>(As, p.dhp, 2 * FZ_D, 0, row0);
    __syncthreads();
    gemm_run<RF, 12, true>(R, As, WG(p.pw1, wave, GSZ), WG(p.w2, wave, GSZ), acc, lane);
    epilogue<RF, 8>(acc, Stg, As, row0, nullptr,
        [&](int row, int col) { return PreRes{gld16(p.dzp, (unsigned)(row * FZ_D + col) * 2u)}; },
        [&](int, int, float (&v)[8], const PreRes& pr, int) {
          float r[8];
          unpack8(pr.res, r);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += r[j];
        }, true);
  } else {
    gemm_issue(R, WG(p.w2, wave, GSZ), lane);
    load_tile<RF>(As, p.dz2, FZ_D, 0, row0);
    __syncthreads();
  }
  stamp();
  // ---- LN2 backward: dr2 (and dr2 * FF2 dropout mask = the gradient of the FF2 Linear output) ---------------------------
  ln_bwd_tile<RF, false, DROP>(As, g2, p.r2, row0, T, p.dr2, p.dr2m, d_ff2, d_ff2, Stg, part + 2 * FZ_D);  // ln2 dgain | ln2 dbias | b2
  stamp();
  // ---- dh1 = (df2 . W2) * GELU'(h1) * drop(FF1) ---------------------------------------------------------------------------
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, WG(p.w2, wave, GSZ), WG(p.w1, wave, GSZ), acc, lane);
  cs_zero();
  epilogue<RF, 8>(acc, Stg, As, row0, nullptr,
      [&](int row, int col) { return PreRes{gld16(p.h1, (unsigned)(row * FZ_D + col) * 2u)}; },
      [&](int row, int col, float (&v)[8], const PreRes& pr, int i) {
        float a[8];
        unpack8(pr.res, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(a[j]);
#ifdef FZ_HASH_BWD_MASKS
        apply_drop<DROP>(d_ff1, (unsigned long long)row * FZ_D + col, v);
#else
        mask_from_saved<DROP>(d_ff1, a, v);  // h1 = dropout(z1 W1 + b1)
#endif
        if (row < T) {
#pragma unroll
          for (int j = 0; j < 8; ++j) cs[i][j] += v[j];
        }
        gst16(p.dh1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, true);
  colsum_flush(cs, Stg, part + 5 * FZ_D);  // b1 gradient
  stamp();
  // ---- dz1 = dh1 . W1 + dr2 ----------------------------------------------------------------------------------------------------
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, WG(p.w1, wave, GSZ), WG(p.wo, wave, GSZ), acc, lane);
  epilogue<RF, 8>(acc, Stg, As, row0, nullptr,
      [&](int row, int col) { return PreRes{gld16(p.dr2, (unsigned)(row * FZ_D + col) * 2u)}; },
      [&](int, int, float (&v)[8], const PreRes& pr, int) {
        float r[8];
        unpack8(pr.res, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
      }, true);
  stamp();
  // ---- LN1 backward (through the post-LN dropout) -> dr1 -------------------------------------------------------------------
  ln_bwd_tile<RF, DROP, false>(As, g1, p.r1, row0, T, p.dr1, nullptr, d_postln, d_postln, Stg, part + 6 * FZ_D);  // ln1 dgain | ln1 dbias | bo
  stamp();
  // ---- dctx = dr1 . Wo ------------------------------------------------------------------------------------------------------------
  zero_acc<RF>(acc);
  gemm_run<RF, 12, false>(R, As, WG(p.wo, wave, GSZ), nullptr, acc, lane);
  epilogue<RF, 8>(acc, Stg, As, row0, nullptr, [&](int, int) { return PreNone{}; },
      [&](int row, int col, float (&v)[8], const PreNone&, int) { gst16(p.dctx, (unsigned)(row * FZ_D + col) * 2u, pack8(v)); }, false);
  stamp();
}

// ---- global network forward in one launch (fused.h: GlobFwd) ---------------------------------------------------------------

// Attention on LDS tiles, fp32 VALU (the problems are tiny: <= 32 keys of 48 channels per (query, head)).  Two threads per
// (query row r, head h): each takes 24 of the head's 48 channels.  Scores go through Sc[pair][32] (the staging buffer).
//   SELF: query r belongs to sequence r / Lk (position r % Lk), its keys are that sequence's rows; mask row id as attn_short
//   else: one query per sequence r (the context block), keys = rows [r Lk, r Lk + Lk); mask row id as attn_q1
// Output: bf16 into Os (zeros for rows >= nq), and for rows < nq into o_glob / lse_glob (global row = grow0 + r).
template <bool SELF, bool DROP>
__device__ __forceinline__ void tile_attn(const bf16_t* Qs, const bf16_t* Ks, const bf16_t* Vs, bf16_t* Os, float* Sc, int nq, int Lk,
                                          int seq0, const long long* lens, float scale, const DropK& dk, bf16_t* o_glob, float* lse_glob,
                                          int grow0) {
  const int tid = threadIdx.x, pair = tid >> 1, half = tid & 1, r = pair >> 3, h = pair & 7;
  const bool on = r < nq;
  const int sl = on ? (SELF ? r / Lk : r) : 0;
  const int kr0 = sl * Lk;
  const int nvalid = on ? (int)lens[seq0 + sl] : 0;
  const int coff = h * 48 + half * 24;
  float q[24];
#pragma unroll
  for (int c = 0; c < 3; ++c) unpack8(*reinterpret_cast<const u32x4_t*>(&Qs[(on ? r : 0) * APITCH + coff + c * 8]), &q[c * 8]);
  float m = -3.0e38f;
  for (int j = 0; j < Lk; ++j) {
    float kk[24];
#pragma unroll
    for (int c = 0; c < 3; ++c) unpack8(*reinterpret_cast<const u32x4_t*>(&Ks[(kr0 + j) * APITCH + coff + c * 8]), &kk[c * 8]);
    float sv = 0.f;
#pragma unroll
    for (int c = 0; c < 24; ++c) sv += q[c] * kk[c];
    sv += __shfl_xor(sv, 1, 64);
    sv *= scale;
    if (j >= nvalid) sv = kMaskFill;
    if (half == 0) Sc[pair * 32 + j] = sv;  // both threads of the pair hold the same value; the partner (the next lane) reads it below
    m = fmaxf(m, sv);
  }
  // the partner's reads below follow these writes in the same wave's LDS queue (DS operations of a wave execute in order);
  // the fence only keeps the compiler from moving them
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float sum = 0.f;
  for (int j = 0; j < Lk; ++j) sum += __expf(Sc[pair * 32 + j] - m);
  const float lse = m + __logf(sum);
  float o[24];
#pragma unroll
  for (int c = 0; c < 24; ++c) o[c] = 0.f;
  const unsigned mrow = SELF ? (unsigned)(((seq0 + sl) * 8 + h) * Lk + (r - kr0)) : (unsigned)((seq0 + sl) * 8 + h);
  for (int j = 0; j < Lk; ++j) {
    float pj = __expf(Sc[pair * 32 + j] - lse);
    if constexpr (DROP) pj *= attn_drop_f(dk.key, mrow, j, (unsigned)(Lk + 1) >> 1, dk.thr, dk.inv_keep);
    float vv[24];
#pragma unroll
    for (int c = 0; c < 3; ++c) unpack8(*reinterpret_cast<const u32x4_t*>(&Vs[(kr0 + j) * APITCH + coff + c * 8]), &vv[c * 8]);
#pragma unroll
    for (int c = 0; c < 24; ++c) o[c] += pj * vv[c];
  }
  if (r < 32) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const u32x4_t pk = on ? pack8(&o[c * 8]) : u32x4_t{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4_t*>(&Os[r * APITCH + coff + c * 8]) = pk;
      if (on) gst16(o_glob, (unsigned)((grow0 + r) * FZ_D + coff + c * 8) * 2u, pk);
    }
    if (on && half == 0) lse_glob[(long)(grow0 + r) * 8 + h] = lse;
  }
}

// L2 warm-up.  A 32-row tile chain streams all twelve 384 x 384 weight matrices (3.5 MB) through ONE CU; with only 8-64 such
// workgroups on the chip each is alone on its XCD's L2 and, cold, pulls them from HBM / Infinity Cache at the rate one CU can
// (measured ~30 GB/s: 115 us for the pass, whatever the launch count).  The otherwise idle CUs of the XCD fetch them instead:
// helper workgroup h of n on an XCD touches the 8 KB pieces h, h + n, ... of the eight P48 arrays in the order of use, so the
// chain finds them in L2 (block b runs on XCD b % 8: an affinity assumed for speed only, correctness does not depend on it).
__device__ __forceinline__ void l2_prefetch(const bf16_t* const (&ws)[8], int h, int n) {
  constexpr unsigned SZQKV = 3u * FZ_D * FZ_D * 2u, SZ = FZ_D * FZ_D * 2u, PIECE = 8192u;  // bytes; 512 threads x 16 B
  unsigned x = 0;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const unsigned bytes = (a & 3) == 0 ? SZQKV : SZ;
    for (unsigned piece = (unsigned)h; piece * PIECE < bytes; piece += (unsigned)n) {
      const u32x4_t v = gld16(ws[a], piece * PIECE + threadIdx.x * 16u);
      x ^= v[0];
    }
  }
  asm volatile("" :: "v"(x));  // the loads must be issued; their values are not needed
}

// block -> tile of the single-launch global passes.  Block b is expected on XCD b % 8 and serves slot b / 8 of that XCD: the first
// tiles_per_xcd slots of the XCDs [xcd_first, xcd_first + xcd_count) are chain workgroups (tile = slot * xcd_count + position of
// the XCD in the set), the following warm_per_xcd slots L2 warm-up helpers of that XCD; blocks on other XCDs leave at once (their
// L2 belongs to the other network).  Returns the tile, or -1 after doing a helper's work / nothing.
template <typename P>
__device__ __forceinline__ int glob_tile_or_help(const P& p, const bf16_t* const (&ws)[8]) {
  const int b = blockIdx.x, xr = (b & 7) - p.xcd_first, slot = b >> 3;
  if (xr < 0 || xr >= p.xcd_count) return -1;
  if (slot < p.tiles_per_xcd) {
    const int tile = slot * p.xcd_count + xr;
    return tile < p.tiles ? tile : -1;
  }
  l2_prefetch(ws, slot - p.tiles_per_xcd, p.warm_per_xcd);
  return -1;
}

// out-proj + residual + LN1 (+ dropout) + FF1 + GELU + FF2 + residual + LN2 on the 32-row tile in As (rows [row0, rowEnd) are
// real; nothing is stored for the others).  The chain of post_attn_fwd_kernel with row masks and global parameter vectors.
// R: the carried weight ring, holding the first k-blocks of L.wo on entry and those of `next` (the pass that follows the chain) on exit
template <bool DROP, bool OUT32>
__device__ __forceinline__ void glob_chain_fwd(bf16_t* As, float* Stg, const GlobLayerFwd& L, const bf16_t* xres, int row0, int rowEnd,
                                               float* z2_f32, long ldz2_f32, unsigned long long sbase, TileRing<2>& R, const bf16_t* next) {
  constexpr int RF = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const DropK d_postln = resolve_drop(L.d_postln, sbase), d_ff1 = resolve_drop(L.d_ff1, sbase), d_ff2 = resolve_drop(L.d_ff2, sbase);
  f32x4_t acc[RF][3];
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, L.wo + wave * GSZ, L.w1 + wave * GSZ, acc, lane);
  epilogue<RF, 8>(acc, Stg, As, row0, L.bo,
      [&](int row, int col) { return PreRes{row < rowEnd ? gld16(xres, (unsigned)(row * FZ_D + col) * 2u) : u32x4_t{0u, 0u, 0u, 0u}}; },
      [&](int row, int col, float (&v)[8], const PreRes& pr, int) {
        float r[8];
        unpack8(pr.res, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        if (row < rowEnd) gst16(L.r1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, true);
  ln_tile<RF, DROP, false, true>(As, L.ln1g, L.ln1b, row0, rowEnd, L.z1, nullptr, 0, d_postln);
  __syncthreads();
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, L.w1 + wave * GSZ, L.w2 + wave * GSZ, acc, lane);
  epilogue<RF, 8>(acc, Stg, As, row0, L.b1, [&](int, int) { return PreNone{}; },
      [&](int row, int col, float (&v)[8], const PreNone&, int) {
        apply_drop<DROP>(d_ff1, (unsigned long long)row * FZ_D + col, v);
        if (row < rowEnd) gst16(L.h1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
        if (row < rowEnd) gst16(L.a1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, true);
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, L.w2 + wave * GSZ, next, acc, lane);
  __syncthreads();  // z1 (stored by this workgroup in ln_tile) is read back as the residual
  epilogue<RF, 8>(acc, Stg, As, row0, L.b2,
      [&](int row, int col) { return PreRes{row < rowEnd ? gld16(L.z1, (unsigned)(row * FZ_D + col) * 2u) : u32x4_t{0u, 0u, 0u, 0u}}; },
      [&](int row, int col, float (&v)[8], const PreRes& pr, int) {
        float r[8];
        unpack8(pr.res, r);
        apply_drop<DROP>(d_ff2, (unsigned long long)row * FZ_D + col, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        if (row < rowEnd) gst16(L.r2, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, true);
  {
    const DropK none;
    ln_tile<RF, false, OUT32, true>(As, L.ln2g, L.ln2b, row0, rowEnd, L.z2, z2_f32, ldz2_f32, none);
  }
  __syncthreads();
}

template <bool DROP>
__global__ __launch_bounds__(512) void glob_fwd_kernel(GlobFwd p) {
  constexpr int RF = 2, BT = 32;
  // LDS: token tile | q | k | v tiles (bf16 [32][400]) | fp32 staging [32][388], also the attention score buffer [256][32]
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BT * APITCH * 2 + BT * SPITCH * 4];
  static_assert(BT * SPITCH * 4 >= 256 * 32 * 4, "the staging buffer holds the attention scores [256 pairs][32 keys]");
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Qs = As + BT * APITCH;
  bf16_t* Ks = Qs + BT * APITCH;
  bf16_t* Vs = Ks + BT * APITCH;
  float* Stg = reinterpret_cast<float*>(smem + 4 * BT * APITCH * 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cm = p.Cmax, G = 32 / Cm;
  int tile;
  {  // chain workgroup, L2 warm-up helper or bystander (see launch_glob_fwd)
    const bf16_t* ws[8] = {p.self.wqkv, p.self.wo, p.self.w1, p.self.w2, p.ctx.wqkv, p.ctx.wo, p.ctx.w1, p.ctx.w2};
    tile = glob_tile_or_help(p, ws);
    if (tile < 0) return;
  }
  const int v0 = tile * G, nv = (p.B - v0) < G ? (p.B - v0) : G;
  const int row0 = v0 * Cm, nrows = nv * Cm, rowEnd = row0 + nrows;
  unsigned long long sbase = 0;
  if constexpr (DROP) { if (p.self.d_ff1.seed_ptr) sbase = *p.self.d_ff1.seed_ptr; }
  const float scale = 0.14433756729740643f;  // 1 / sqrt(48)
  f32x4_t acc[RF][3];
  int tsn = 0;
  auto stamp = [&]() { if (p.tstamps && tile == 0 && threadIdx.x == 0) p.tstamps[tsn] = __builtin_amdgcn_s_memtime(); ++tsn; };
  stamp();
  TileRing<RF> R;  // weight ring, carried through all twelve passes (gemm_run)
  gemm_issue(R, p.self.wqkv + wave * GSZ, lane);  // (flies during the input LayerNorm)

  // ---- z0 = LayerNorm(x) + pe (transformer_legacy.py:222-241; padded items are zero rows: LN gives bias + pe there) ----
  {
    const int j16 = lane & 15, g = lane >> 4, rl = wave * 4 + g, row = row0 + rl;
    const bool on = rl < nrows;
    float x[3][8];
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      if (on) load8f(p.x + (long)row * FZ_D + m * 128 + j16 * 8, x[m]);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[m][e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[m][e];
    }
    const float mean = row16_sum(s) * (1.0f / 384.0f);
    float qq = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[m][e] -= mean; qq += x[m][e] * x[m][e]; }
    const float rs = 1.0f / (sqrtf(row16_sum(qq) * (1.0f / 383.0f)) + kLnEps);
    const int pos = rl % Cm;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int col = m * 128 + j16 * 8;
      float gn[8], bi[8], pe[8], y[8];
      load8f(p.n_gain + col, gn); load8f(p.n_bias + col, bi); load8f(p.pe + (long)pos * FZ_D + col, pe);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = on ? x[m][e] * rs * gn[e] + bi[e] + pe[e] : 0.f;
      const u32x4_t o = pack8(y);
      *reinterpret_cast<u32x4_t*>(&As[rl * APITCH + col]) = o;
      if (on) gst16(p.z0, (unsigned)(row * FZ_D + col) * 2u, o);
    }
  }
  __syncthreads();
  stamp();

  // ---- self-attention encoder layer -------------------------------------------------------------------------------------
#pragma unroll 1
  for (int q = 0; q < 3; ++q) {
    zero_acc<RF>(acc);
    gemm_run<RF, 12, true>(R, As, p.self.wqkv + (q * 8 + wave) * GSZ, q < 2 ? p.self.wqkv + ((q + 1) * 8 + wave) * GSZ : p.self.wo + wave * GSZ, acc, lane);
    bf16_t* dst = q == 0 ? Qs : (q == 1 ? Ks : Vs);
    epilogue<RF, 8>(acc, Stg, dst, row0, p.self.bqkv + q * FZ_D, [&](int, int) { return PreNone{}; },
        [&](int row, int col, float (&v)[8], const PreNone&, int) {
          if (row < rowEnd) gst16(p.self.q, (unsigned)(row * (3 * FZ_D) + q * FZ_D + col) * 2u, pack8(v));
        }, true);
    stamp();
  }
  __syncthreads();
  {
    const DropK dk = resolve_drop(p.self.d_attn, sbase);
    tile_attn<true, DROP>(Qs, Ks, Vs, As, Stg, nrows, Cm, v0, p.lens, scale, dk, p.self.ctx, p.self.lse, row0);
  }
  __syncthreads();
  stamp();
  {
    const bf16_t* nx = p.ctx.wqkv + (8 + wave) * GSZ;  // the context block's key projection follows
    if (p.per_token) glob_chain_fwd<DROP, true>(As, Stg, p.self, p.z0, row0, rowEnd, p.per_token, FZ_D, sbase, R, nx);
    else glob_chain_fwd<DROP, false>(As, Stg, p.self, p.z0, row0, rowEnd, nullptr, 0, sbase, R, nx);
  }
  stamp();

  // ---- avg_special pooling (poolers.py:237-238): the sum runs over ALL Cmax rows, padded ones included -----------------------
  if (tid < FZ_D) {
    for (int gv = 0; gv < nv; ++gv) {
      float a = 0.f;
      for (int j = 0; j < Cm; ++j) a += bf2f(As[(gv * Cm + j) * APITCH + tid]);
      p.pooled[(long)(v0 + gv) * (2 * FZ_D) + tid] = a / (float)p.lens[v0 + gv];
    }
  }

  // ---- context block (transformer_legacy.py:251-267): keys / values from the encoder output, one query per sequence ----------
#pragma unroll 1
  for (int q = 1; q < 3; ++q) {
    zero_acc<RF>(acc);
    gemm_run<RF, 12, true>(R, As, p.ctx.wqkv + (q * 8 + wave) * GSZ, q == 1 ? p.ctx.wqkv + (16 + wave) * GSZ : p.ctx.wqkv + wave * GSZ, acc, lane);
    bf16_t* dst = q == 1 ? Ks : Vs;
    bf16_t* gdst = q == 1 ? p.ctx.k : p.ctx.v;
    const int ldkv = (int)(q == 1 ? p.ctx.ldk : p.ctx.ldv);
    epilogue<RF, 8>(acc, Stg, dst, row0, p.ctx.bqkv + q * FZ_D, [&](int, int) { return PreNone{}; },
        [&](int row, int col, float (&v)[8], const PreNone&, int) {
          if (row < rowEnd) gst16(gdst, (unsigned)(row * ldkv + col) * 2u, pack8(v));
        }, true);
  }
  __syncthreads();  // every wave is done with the encoder output in As
  stamp();
  {  // query tile: row gv = bf16(hidden[v0 + gv]); 48 16-byte chunks per row
    for (int c = tid; c < BT * 48; c += 512) {
      const int rl = c / 48, ch = c - rl * 48;
      u32x4_t o = {0u, 0u, 0u, 0u};
      if (rl < nv) {
        float h[8];
        load8f(p.hidden + (long)(v0 + rl) * FZ_D + ch * 8, h);
        o = pack8(h);
        gst16(p.cq_in, (unsigned)((v0 + rl) * FZ_D + ch * 8) * 2u, o);
      }
      *reinterpret_cast<u32x4_t*>(&As[rl * APITCH + ch * 8]) = o;
    }
  }
  __syncthreads();
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, p.ctx.wqkv + wave * GSZ, p.ctx.wo + wave * GSZ, acc, lane);
  epilogue<RF, 8>(acc, Stg, Qs, v0, p.ctx.bqkv, [&](int, int) { return PreNone{}; },
      [&](int row, int col, float (&v)[8], const PreNone&, int) {
        if (row < v0 + nv) gst16(p.ctx.q, (unsigned)(row * (int)p.ctx.ldq + col) * 2u, pack8(v));
      }, true);
  __syncthreads();
  stamp();
  {
    const DropK dk = resolve_drop(p.ctx.d_attn, sbase);
    tile_attn<false, DROP>(Qs, Ks, Vs, As, Stg, nv, Cm, v0, p.lens, scale, dk, p.ctx.ctx, p.ctx.lse, v0);
  }
  __syncthreads();
  stamp();
  glob_chain_fwd<DROP, true>(As, Stg, p.ctx, p.cq_in, v0, v0 + nv, p.pooled + FZ_D, 2 * FZ_D, sbase, R, p.ctx.w2 + wave * GSZ /* nothing follows */);
  stamp();
}

// ---- global network backward in one launch (fused.h: GlobBwd) ---------------------------------------------------------------

// Column sums over the rows of a 32-row tile held one row per 16-lane group (lane j16 holds columns 128 m + 8 j16 .. + 8 of the
// three 128-column chunks): NQ quantities, dst[q][384] += (global atomics: at most 64 workgroups per launch add one row each).
// One 128-column chunk at a time through red[32 rows][NQ x 128] (plain LDS stores, NQ x 128 threads add the 32 rows): the
// register route (xor-shuffles over the 4 row groups of a wave = 48 NQ ds_bpermute per lane) made a LayerNorm backward of 32
// rows cost 17k cycles, twice a GEMM pass.
template <int NQ>
__device__ __forceinline__ void rows_colsum_flush(const float (&cs)[NQ][3][8], float* red, float* const (&dst)[NQ]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j16 = lane & 15, g = lane >> 4;
  const int rl = wave * 4 + g;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    lds_barrier();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float* w = red + rl * (NQ * 128) + q * 128 + j16 * 8;
      *reinterpret_cast<f32x4_t*>(w) = f32x4_t{cs[q][m][0], cs[q][m][1], cs[q][m][2], cs[q][m][3]};
      *reinterpret_cast<f32x4_t*>(w + 4) = f32x4_t{cs[q][m][4], cs[q][m][5], cs[q][m][6], cs[q][m][7]};
    }
    lds_barrier();
    if (threadIdx.x < NQ * 128) {
      float v = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) v += red[r * (NQ * 128) + threadIdx.x];
      float* d = dst[threadIdx.x / 128];
      if (d) acc_add(d + m * 128 + (threadIdx.x & 127), v);
    }
  }
  lds_barrier();
}

// LayerNorm backward of the 32 tile rows, one row per 16-lane group; rows [row0, rowEnd) are real.
//   dy: the tile (bf16) or fp32 global rows (DY32);  x: the saved LN input, bf16 global, or the fp32 network input (X32)
//   DROPY: dy is first multiplied by the dropout mask the forward applied to the LN output
//   MASKX: dxm = dx * mask(dx_drop) is what stays in the tile and goes to out_dxm; dx goes to out_dx (bf16) / out32 (fp32)
//   gradients: g_gain += sum dy xc / s, g_bias += sum dy (the column sums of dx / dxm = the bias gradient of the Linear in front
//   are taken by the weight-gradient GEMM that streams that tensor anyway: GemmTN::a_colsum)
template <bool DROPY, bool MASKX, bool DY32, bool X32>
__device__ __forceinline__ void glob_ln_bwd(bf16_t* As, const float* dy32, long lddy32, const void* xsrc, const float* gain, int row0, int rowEnd,
                                            bf16_t* out_dx, bf16_t* out_dxm, float* out32, const DropK& dy_drop, const DropK& dx_drop, float* red,
                                            float* g_gain, float* g_bias) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j16 = lane & 15, g = lane >> 4;
  const int rl = wave * 4 + g, row = row0 + rl;
  const bool on = row < rowEnd;
  float x[3][8], dy[3][8], gn[3][8];
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int col = m * 128 + j16 * 8;
    load8f(gain + col, gn[m]);
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[m][e] = 0.f; dy[m][e] = 0.f; }
    if (on) {
      if constexpr (X32) load8f(reinterpret_cast<const float*>(xsrc) + (long)row * FZ_D + col, x[m]);
      else unpack8(gld16(xsrc, (unsigned)(row * FZ_D + col) * 2u), x[m]);
      if constexpr (DY32) load8f(dy32 + (long)row * lddy32 + col, dy[m]);
      else unpack8(*reinterpret_cast<const u32x4_t*>(&As[rl * APITCH + col]), dy[m]);
      if constexpr (DROPY) {
        float sc[8];
        drop_scales_key<8>(dy_drop.key, (unsigned long long)row * FZ_D + col, dy_drop.thr, dy_drop.inv_keep, sc);
#pragma unroll
        for (int e = 0; e < 8; ++e) dy[m][e] *= sc[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s += x[m][e];
  }
  const float mu = row16_sum(s) * (1.0f / 384.0f);
  float q = 0.f, hs = 0.f, hx = 0.f;
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      x[m][e] -= mu;
      q += x[m][e] * x[m][e];
      const float h = dy[m][e] * gn[m][e];
      hs += h; hx += h * x[m][e];
    }
  q = row16_sum(q); hs = row16_sum(hs); hx = row16_sum(hx);
  const float stdv = sqrtf(q * (1.0f / 383.0f));
  const float rs = 1.0f / (stdv + kLnEps), hmean = hs * (1.0f / 384.0f);
  const float k2 = stdv > 0.f ? hx * rs * rs / (383.0f * stdv) : 0.f;
  float cs[2][3][8];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int col = m * 128 + j16 * 8;
    float dx[8], dm[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dx[e] = (dy[m][e] * gn[m][e] - hmean) * rs - k2 * x[m][e];
    if constexpr (MASKX) {
      float sc[8];
      drop_scales_key<8>(dx_drop.key, (unsigned long long)row * FZ_D + col, dx_drop.thr, dx_drop.inv_keep, sc);
#pragma unroll
      for (int e = 0; e < 8; ++e) dm[e] = dx[e] * sc[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dm[e] = dx[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cs[0][m][e] = on ? dy[m][e] * x[m][e] * rs : 0.f;
      cs[1][m][e] = on ? dy[m][e] : 0.f;
    }
    const u32x4_t om = on ? pack8(dm) : u32x4_t{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4_t*>(&As[rl * APITCH + col]) = om;
    if (on) {
      const unsigned boff = (unsigned)(row * FZ_D + col) * 2u;
      if (out_dx) gst16(out_dx, boff, pack8(dx));
      if constexpr (MASKX) gst16(out_dxm, boff, om);
      if (out32) {
        float* o32 = out32 + (long)row * FZ_D + col;
        *reinterpret_cast<f32x4_t*>(o32) = f32x4_t{dx[0], dx[1], dx[2], dx[3]};
        *reinterpret_cast<f32x4_t*>(o32 + 4) = f32x4_t{dx[4], dx[5], dx[6], dx[7]};
      }
    }
  }
  float* const dst[2] = {g_gain, g_bias};
  rows_colsum_flush<2>(cs, red, dst);
}

// rows [grow0, grow0 + nvalid) of a bf16 [., ld] matrix at column col0 -> tile rows 0.. (the rest of the 32 rows: zeros)
__device__ __forceinline__ void load_rows32(bf16_t* Ts, const bf16_t* src, long ld, int col0, int grow0, int nvalid) {
  for (int c = threadIdx.x; c < 32 * 48; c += NTHR) {
    const int rl = c / 48, ch = c - rl * 48;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (rl < nvalid) v = gld16(src, (unsigned)((grow0 + rl) * (int)ld + col0 + ch * 8) * 2u);
    *reinterpret_cast<u32x4_t*>(&Ts[rl * APITCH + ch * 8]) = v;
  }
}

// The chain of pre_attn_bwd_kernel on the 32-row tile: LN2 backward, FF2^T, GELU', FF1^T, LN1 backward (through the post-LN
// dropout), out-proj^T.  In: the gradient wrt the layer output in As (bf16) or dy32 (fp32 rows).  Out: the gradient wrt the
// attention output in As; dr2 / dr2m / dh1 / dr1 in global memory (the weight-gradient GEMMs read them), parameter gradients added.
// R: the carried weight ring — the first k-blocks of L.w2_kn on entry, those of `next` (the pass that follows the chain) on exit.
template <bool DROP, bool DY32>
__device__ __forceinline__ void glob_chain_bwd(bf16_t* As, float* Stg, const GlobLayerBwd& L, const float* dy32, long lddy32, int row0, int rowEnd,
                                               unsigned long long sbase, TileRing<2>& R, const bf16_t* next) {
  constexpr int RF = 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const DropK d_postln = resolve_drop(L.d_postln, sbase), d_ff1 = resolve_drop(L.d_ff1, sbase), d_ff2 = resolve_drop(L.d_ff2, sbase);
  f32x4_t acc[RF][3];
  glob_ln_bwd<false, DROP, DY32, false>(As, dy32, lddy32, L.r2, L.ln2g, row0, rowEnd, L.dr2, L.dr2m, nullptr, d_ff2, d_ff2, Stg, L.g_ln2g, L.g_ln2b);
  __syncthreads();  // the tile holds df2; dr2 (global) is read back below
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, L.w2_kn + wave * GSZ, L.w1_kn + wave * GSZ, acc, lane);
  epilogue<RF, 8>(acc, Stg, As, row0, nullptr,
      [&](int row, int col) { return PreRes{row < rowEnd ? gld16(L.h1, (unsigned)(row * FZ_D + col) * 2u) : u32x4_t{0u, 0u, 0u, 0u}}; },
      [&](int row, int col, float (&v)[8], const PreRes& pr, int) {
        float a[8];
        unpack8(pr.res, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(a[j]);
        apply_drop<DROP>(d_ff1, (unsigned long long)row * FZ_D + col, v);
        if (row < rowEnd) {
          gst16(L.dh1, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
      }, true);
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, L.w1_kn + wave * GSZ, L.wo_kn + wave * GSZ, acc, lane);
  epilogue<RF, 8>(acc, Stg, As, row0, nullptr,
      [&](int row, int col) { return PreRes{row < rowEnd ? gld16(L.dr2, (unsigned)(row * FZ_D + col) * 2u) : u32x4_t{0u, 0u, 0u, 0u}}; },
      [&](int, int, float (&v)[8], const PreRes& pr, int) {
        float r[8];
        unpack8(pr.res, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
      }, true);
  glob_ln_bwd<DROP, false, false, false>(As, nullptr, 0, L.r1, L.ln1g, row0, rowEnd, L.dr1, nullptr, nullptr, d_postln, d_postln, Stg, L.g_ln1g,
                                         L.g_ln1b);
  __syncthreads();
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, As, L.wo_kn + wave * GSZ, next, acc, lane);
  epilogue<RF, 8>(acc, Stg, As, row0, nullptr, [&](int, int) { return PreNone{}; }, [&](int, int, float (&)[8], const PreNone&, int) {}, true);
  __syncthreads();
}

template <bool DROP>
__global__ __launch_bounds__(512) void glob_bwd_kernel(GlobBwd p) {
  constexpr int RF = 2, BT = 32;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BT * APITCH * 2 + BT * SPITCH * 4];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Qs = As + BT * APITCH;
  bf16_t* Ks = Qs + BT * APITCH;
  bf16_t* Vs = Ks + BT * APITCH;
  float* Stg = reinterpret_cast<float*>(smem + 4 * BT * APITCH * 2);
  float* Sc = Stg;               // attention: probabilities [256 pairs][32 keys]
  float* Dl = Stg + 256 * 32;    // delta [256 pairs]
  float* Ls = Dl + 256;          // lse [256 pairs]
  static_assert(BT * SPITCH >= 256 * 32 + 512, "staging buffer holds the attention scratch");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Cm = p.Cmax, G = 32 / Cm;
  int tile;
  {
    const bf16_t* ws[8] = {p.ctx.wqkv_kn, p.ctx.wo_kn, p.ctx.w1_kn, p.ctx.w2_kn, p.self.wqkv_kn, p.self.wo_kn, p.self.w1_kn, p.self.w2_kn};
    tile = glob_tile_or_help(p, ws);
    if (tile < 0) return;
  }
  const int v0 = tile * G, nv = (p.B - v0) < G ? (p.B - v0) : G;
  const int row0 = v0 * Cm, nrows = nv * Cm, rowEnd = row0 + nrows;
  unsigned long long sbase = 0;
  if constexpr (DROP) { if (p.self.d_ff1.seed_ptr) sbase = *p.self.d_ff1.seed_ptr; }
  const float scale = 0.14433756729740643f;
  f32x4_t acc[RF][3];
  const int pair = tid >> 1, half = tid & 1, pr_ = pair >> 3, ph = pair & 7, coff = ph * 48 + half * 24;
  int tsn = 0;
  auto stamp = [&]() { if (p.tstamps && tile == 0 && threadIdx.x == 0) p.tstamps[tsn] = __builtin_amdgcn_s_memtime(); ++tsn; };
  stamp();

  // ---- context block backward: rows = sequences (tile row r = sequence v0 + r) -------------------------------------------------
  TileRing<RF> R;  // weight ring, carried through all twelve passes (gemm_run)
  const bf16_t* cqkv = p.ctx.wqkv_kn + (long)wave * (36L * 3 * 512);   // this wave's 48-column group at K = 1152: q | k | v slabs of 12 k-blocks
  const bf16_t* sqkv = p.self.wqkv_kn + (long)wave * (36L * 3 * 512);
  gemm_issue(R, p.ctx.w2_kn + wave * GSZ, lane);
  glob_chain_bwd<DROP, true>(As, Stg, p.ctx, p.dpooled + FZ_D, 2 * FZ_D, v0, v0 + nv, sbase, R, cqkv);   // As: gradient wrt the attention output
  stamp();
  load_rows32(Qs, p.ctx.q, p.ctx.ldq, 0, v0, nv);
  load_rows32(Ks, p.ctx.k, p.ctx.ldk, 0, row0, nrows);
  load_rows32(Vs, p.ctx.v, p.ctx.ldv, 0, row0, nrows);
  __syncthreads();
  {  // one-query attention backward, thread pair = (sequence r, head h), in place: Qs -> dq, Ks -> dk, Vs -> dv
    const DropK dk_ = resolve_drop(p.ctx.d_attn, sbase);
    const bool on = pr_ < nv;
    const int r = on ? pr_ : 0, kr0 = r * Cm;
    const int nvalid = on ? (int)p.lens[v0 + r] : 0;
    const float lse = on ? p.ctx.lse[(long)(v0 + r) * 8 + ph] : 0.f;
    float q[24], dO[24];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      unpack8(*reinterpret_cast<const u32x4_t*>(&Qs[r * APITCH + coff + c * 8]), &q[c * 8]);
      unpack8(*reinterpret_cast<const u32x4_t*>(&As[r * APITCH + coff + c * 8]), &dO[c * 8]);
    }
    const unsigned mrow = (unsigned)((v0 + r) * 8 + ph);
    float delta = 0.f;
    for (int j = 0; j < Cm; ++j) {
      float kk[24], vv[24];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        unpack8(*reinterpret_cast<const u32x4_t*>(&Ks[(kr0 + j) * APITCH + coff + c * 8]), &kk[c * 8]);
        unpack8(*reinterpret_cast<const u32x4_t*>(&Vs[(kr0 + j) * APITCH + coff + c * 8]), &vv[c * 8]);
      }
      float sv = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < 24; ++c) { sv += q[c] * kk[c]; dp += dO[c] * vv[c]; }
      sv += __shfl_xor(sv, 1, 64); dp += __shfl_xor(dp, 1, 64);
      sv = j < nvalid ? sv * scale : kMaskFill;
      const float pj = __expf(sv - lse);
      float dr = 1.f;
      if constexpr (DROP) dr = attn_drop_f(dk_.key, mrow, j, (unsigned)(Cm + 1) >> 1, dk_.thr, dk_.inv_keep);
      delta += pj * dr * dp;
      if (half == 0) { Sc[pair * 32 + j] = pj; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float dq[24];
#pragma unroll
    for (int c = 0; c < 24; ++c) dq[c] = 0.f;
    for (int j = 0; j < Cm; ++j) {
      float kk[24], vv[24];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        unpack8(*reinterpret_cast<const u32x4_t*>(&Ks[(kr0 + j) * APITCH + coff + c * 8]), &kk[c * 8]);
        unpack8(*reinterpret_cast<const u32x4_t*>(&Vs[(kr0 + j) * APITCH + coff + c * 8]), &vv[c * 8]);
      }
      float dp = 0.f;
#pragma unroll
      for (int c = 0; c < 24; ++c) dp += dO[c] * vv[c];
      dp += __shfl_xor(dp, 1, 64);
      const float pj = Sc[pair * 32 + j];
      float dr = 1.f;
      if constexpr (DROP) dr = attn_drop_f(dk_.key, mrow, j, (unsigned)(Cm + 1) >> 1, dk_.thr, dk_.inv_keep);
      const float ds = j < nvalid ? pj * (dr * dp - delta) * scale : 0.f;
      const float pd = pj * dr;
#pragma unroll
      for (int c = 0; c < 24; ++c) dq[c] += ds * kk[c];
      if (on) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float t8[8], u8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { t8[e] = ds * q[c * 8 + e]; u8[e] = pd * dO[c * 8 + e]; }
          const u32x4_t pk = pack8(t8), pv = pack8(u8);
          *reinterpret_cast<u32x4_t*>(&Ks[(kr0 + j) * APITCH + coff + c * 8]) = pk;
          *reinterpret_cast<u32x4_t*>(&Vs[(kr0 + j) * APITCH + coff + c * 8]) = pv;
          gst16(p.ctx.dk, (unsigned)((row0 + kr0 + j) * (int)p.ctx.lddk + coff + c * 8) * 2u, pk);
          gst16(p.ctx.dv, (unsigned)((row0 + kr0 + j) * (int)p.ctx.lddv + coff + c * 8) * 2u, pv);
        }
      }
    }
    if (on) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const u32x4_t pk = pack8(&dq[c * 8]);
        *reinterpret_cast<u32x4_t*>(&Qs[r * APITCH + coff + c * 8]) = pk;
        gst16(p.ctx.dq, (unsigned)((v0 + r) * (int)p.ctx.lddq + coff + c * 8) * 2u, pk);
      }
    }
  }
  __syncthreads();
  stamp();
  // ---- gradient wrt the context vectors: dhidden = dq . Wq + dr1 ------------------------------------------------------------------
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, Qs, cqkv, cqkv + 12L * 3 * 512, acc, lane);
  epilogue<RF, 8>(acc, Stg, As, v0, nullptr,
      [&](int row, int col) { return PreRes{row < v0 + nv ? gld16(p.ctx.dr1, (unsigned)(row * FZ_D + col) * 2u) : u32x4_t{0u, 0u, 0u, 0u}}; },
      [&](int row, int col, float (&v)[8], const PreRes& pr, int) {
        float r[8];
        unpack8(pr.res, r);
        if (row < v0 + nv) {
          float* o = p.dhidden + (long)row * FZ_D + col;
          *reinterpret_cast<f32x4_t*>(o) = f32x4_t{v[0] + r[0], v[1] + r[1], v[2] + r[2], v[3] + r[3]};
          *reinterpret_cast<f32x4_t*>(o + 4) = f32x4_t{v[4] + r[4], v[5] + r[5], v[6] + r[6], v[7] + r[7]};
        }
      }, false);
  stamp();
  // ---- gradient wrt the encoder output: dk . Wk + dv . Wv + avg_special backward (dpooled / len on ALL Cmax rows) ---------------------
  zero_acc<RF>(acc);
  gemm_run<RF, 12, true>(R, Ks, cqkv + 12L * 3 * 512, cqkv + 24L * 3 * 512, acc, lane);
  gemm_run<RF, 12, true>(R, Vs, cqkv + 24L * 3 * 512, p.self.w2_kn + wave * GSZ, acc, lane);
  epilogue<RF, 8>(acc, Stg, As, row0, nullptr, [&](int, int) { return PreNone{}; },
      [&](int row, int col, float (&v)[8], const PreNone&, int) {
        if (row < rowEnd) {
          const int vid = row / Cm;
          const float inv = 1.0f / (float)p.lens[vid];
          float g[8];
          load8f(p.dpooled + (long)vid * (2 * FZ_D) + col, g);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += g[j] * inv;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
      }, true);
  __syncthreads();
  stamp();

  // ---- encoder layer backward: rows = tokens -------------------------------------------------------------------------------------
  glob_chain_bwd<DROP, false>(As, Stg, p.self, nullptr, 0, row0, rowEnd, sbase, R, sqkv);   // As: gradient wrt the attention output
  stamp();
  load_rows32(Qs, p.self.q, p.self.ldq, 0, row0, nrows);
  load_rows32(Ks, p.self.k, p.self.ldk, 0, row0, nrows);
  load_rows32(Vs, p.self.v, p.self.ldv, 0, row0, nrows);
  __syncthreads();
  {
    const DropK dk_ = resolve_drop(p.self.d_attn, sbase);
    const bool on = pr_ < nrows;
    const int r = on ? pr_ : 0, sl = r / Cm, kr0 = sl * Cm, cpos = r - kr0;
    const int nvalid = on ? (int)p.lens[v0 + sl] : 0;
    // pass A, thread pair = (query row r, head h): delta, dq
    {
      const float lse = on ? p.self.lse[(long)(row0 + r) * 8 + ph] : 0.f;
      float q[24], dO[24];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        unpack8(*reinterpret_cast<const u32x4_t*>(&Qs[r * APITCH + coff + c * 8]), &q[c * 8]);
        unpack8(*reinterpret_cast<const u32x4_t*>(&As[r * APITCH + coff + c * 8]), &dO[c * 8]);
      }
      const unsigned mrow = (unsigned)(((v0 + sl) * 8 + ph) * Cm + cpos);
      float delta = 0.f;
      for (int j = 0; j < Cm; ++j) {
        float kk[24], vv[24];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          unpack8(*reinterpret_cast<const u32x4_t*>(&Ks[(kr0 + j) * APITCH + coff + c * 8]), &kk[c * 8]);
          unpack8(*reinterpret_cast<const u32x4_t*>(&Vs[(kr0 + j) * APITCH + coff + c * 8]), &vv[c * 8]);
        }
        float sv = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < 24; ++c) { sv += q[c] * kk[c]; dp += dO[c] * vv[c]; }
        sv += __shfl_xor(sv, 1, 64); dp += __shfl_xor(dp, 1, 64);
        sv = j < nvalid ? sv * scale : kMaskFill;
        const float pj = __expf(sv - lse);
        float dr = 1.f;
        if constexpr (DROP) dr = attn_drop_f(dk_.key, mrow, j, (unsigned)(Cm + 1) >> 1, dk_.thr, dk_.inv_keep);
        delta += pj * dr * dp;
        if (half == 0) Sc[pair * 32 + j] = pj;
      }
      if (half == 0) { Dl[pair] = delta; Ls[pair] = lse; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float dq[24];
#pragma unroll
      for (int c = 0; c < 24; ++c) dq[c] = 0.f;
      for (int j = 0; j < Cm; ++j) {
        float kk[24], vv[24];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          unpack8(*reinterpret_cast<const u32x4_t*>(&Ks[(kr0 + j) * APITCH + coff + c * 8]), &kk[c * 8]);
          unpack8(*reinterpret_cast<const u32x4_t*>(&Vs[(kr0 + j) * APITCH + coff + c * 8]), &vv[c * 8]);
        }
        float dp = 0.f;
#pragma unroll
        for (int c = 0; c < 24; ++c) dp += dO[c] * vv[c];
        dp += __shfl_xor(dp, 1, 64);
        const float pj = Sc[pair * 32 + j];
        float dr = 1.f;
        if constexpr (DROP) dr = attn_drop_f(dk_.key, mrow, j, (unsigned)(Cm + 1) >> 1, dk_.thr, dk_.inv_keep);
        const float ds = j < nvalid ? pj * (dr * dp - delta) * scale : 0.f;
#pragma unroll
        for (int c = 0; c < 24; ++c) dq[c] += ds * kk[c];
      }
      if (on) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gst16(p.self.dq, (unsigned)((row0 + r) * (int)p.self.lddq + coff + c * 8) * 2u, pack8(&dq[c * 8]));
      }
    }
    lds_barrier();  // every pair's delta / lse is in LDS
    // pass B, thread pair = (key row r, head h): dk, dv over the queries of the sequence
    {
      float kk[24], vv[24], dkr[24], dvr[24];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        unpack8(*reinterpret_cast<const u32x4_t*>(&Ks[r * APITCH + coff + c * 8]), &kk[c * 8]);
        unpack8(*reinterpret_cast<const u32x4_t*>(&Vs[r * APITCH + coff + c * 8]), &vv[c * 8]);
      }
#pragma unroll
      for (int c = 0; c < 24; ++c) { dkr[c] = 0.f; dvr[c] = 0.f; }
      const bool kvalid = on && cpos < nvalid;
      for (int i = 0; i < Cm; ++i) {
        float q[24], dO[24];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          unpack8(*reinterpret_cast<const u32x4_t*>(&Qs[(kr0 + i) * APITCH + coff + c * 8]), &q[c * 8]);
          unpack8(*reinterpret_cast<const u32x4_t*>(&As[(kr0 + i) * APITCH + coff + c * 8]), &dO[c * 8]);
        }
        float sv = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < 24; ++c) { sv += q[c] * kk[c]; dp += dO[c] * vv[c]; }
        sv += __shfl_xor(sv, 1, 64); dp += __shfl_xor(dp, 1, 64);
        const int qp = (kr0 + i) * 8 + ph;  // the query's pair index
        const float pj = kvalid ? __expf(sv * scale - Ls[qp]) : 0.f;
        float dr = 1.f;
        if constexpr (DROP) dr = attn_drop_f(dk_.key, (unsigned)(((v0 + sl) * 8 + ph) * Cm + i), cpos, (unsigned)(Cm + 1) >> 1, dk_.thr, dk_.inv_keep);
        const float ds = pj * (dr * dp - Dl[qp]) * scale;
        const float pd = pj * dr;
#pragma unroll
        for (int c = 0; c < 24; ++c) { dkr[c] += ds * q[c]; dvr[c] += pd * dO[c]; }
      }
      if (on) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          gst16(p.self.dk, (unsigned)((row0 + r) * (int)p.self.lddk + coff + c * 8) * 2u, pack8(&dkr[c * 8]));
          gst16(p.self.dv, (unsigned)((row0 + r) * (int)p.self.lddv + coff + c * 8) * 2u, pack8(&dvr[c * 8]));
        }
      }
    }
  }
  __syncthreads();  // dq | dk | dv (global, this workgroup's rows) are read back as the GEMM operand
  stamp();
  // ---- dz0 = dqkv . Wqkv + dr1, K = 1152 as three 384-wide tiles ---------------------------------------------------------------------
  zero_acc<RF>(acc);
  load_rows32(Qs, p.self.dq, p.self.lddq, 0, row0, nrows);  // q | k | v tiles are dead: all three operand tiles at once
  load_rows32(Ks, p.self.dk, p.self.lddk, 0, row0, nrows);
  load_rows32(Vs, p.self.dv, p.self.lddv, 0, row0, nrows);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    const bf16_t* Ts = c == 0 ? Qs : (c == 1 ? Ks : Vs);
    gemm_run<RF, 12, true>(R, Ts, sqkv + (long)c * 12 * 3 * 512, sqkv + (long)(c < 2 ? c + 1 : c) * 12 * 3 * 512, acc, lane);
  }
  epilogue<RF, 8>(acc, Stg, As, row0, nullptr,
      [&](int row, int col) { return PreRes{row < rowEnd ? gld16(p.self.dr1, (unsigned)(row * FZ_D + col) * 2u) : u32x4_t{0u, 0u, 0u, 0u}}; },
      [&](int, int, float (&v)[8], const PreRes& pr, int) {
        float r[8];
        unpack8(pr.res, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
      }, true);
  stamp();
  // ---- input LayerNorm backward (the positional encoding has no gradient) -----------------------------------------------------------
  {
    const DropK none;
    glob_ln_bwd<false, false, false, true>(As, nullptr, 0, p.x, p.n_gain, row0, rowEnd, nullptr, nullptr, p.dx, none, none, Stg, p.g_n_gain, p.g_n_bias);
  }
  stamp();
}

// ---- input FC (K = Din, streamed) + GELU + pe + QKV -----------------------------------------------------------------
constexpr int XPITCH = 80;  // bf16 elements per row of a 64-column slab (160 B: conflict-free fragment reads)

template <int RF>
__global__ __launch_bounds__(512) void infc_qkv_fwd_kernel(InfcQkvFwd p) {
  constexpr int BT = 16 * RF, RR = 32, CPT = RF / 4;  // CPT: 16-byte slab chunks per thread (BT rows x 64 columns over 512 threads)
  static_assert(RF == 8 || RF == 4, "tile height");
  __shared__ __attribute__((aligned(16))) unsigned char smem[BT * APITCH * 2 + RR * SPITCH * 4 + 4 * FZ_D * 4];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);          // during the K loop: two [128][80] slabs of xhat
  float* Stg = reinterpret_cast<float*>(smem + BT * APITCH * 2);
  float* Bsm = reinterpret_cast<float*>(smem + BT * APITCH * 2 + RR * SPITCH * 4);  // bin | bq | bk | bv
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_idx = fz_tile_index((p.T + BT - 1) / BT);
  if (tile_idx < 0) return;
  const int row0 = tile_idx * BT, Din = p.Din;
  int tsn = 48;
  // (the pointer is re-read from its SGPR pair at every stamp: as a VGPR address hoisted to the kernel's entry it was the last spill)
  auto stamp = [&]() {
    unsigned long long* ts = p.tstamps;
    asm volatile("" : "+s"(ts));
    if (ts && blockIdx.x == 0 && threadIdx.x == 0) ts[tsn] = __builtin_amdgcn_s_memtime();
    ++tsn;
  };
  stamp();
  if (tid < FZ_D / 4) reinterpret_cast<f32x4_t*>(Bsm)[tid] = reinterpret_cast<const f32x4_t*>(p.bin)[tid];
  else if (tid < 4 * FZ_D / 4) reinterpret_cast<f32x4_t*>(Bsm)[tid] = reinterpret_cast<const f32x4_t*>(p.bqkv)[tid - FZ_D / 4];
  bf16_t* Xs = As;  // [2][BT * XPITCH]
  // slab staging: BT rows x 64 columns = 8 BT 16-byte chunks, CPT per thread (rows r, r + 64)
  const int sr = tid >> 3, sc = (tid & 7) * 8;
  const unsigned g0 = (unsigned)((row0 + sr) * Din + sc) * 2u, g1 = (unsigned)((row0 + sr + 64) * Din + sc) * 2u;
  // Software pipeline of the K loop (one 64-column slab = two k-blocks = 48 MFMAs per wave, ~0.6 us per slab and CU):
  //   xhat: two register sets, each holding the slab TWO ahead (HBM latency > one slab); stored to the free LDS buffer at
  //         the end of the slab before its use;
  //   weights: a ring of three slabs (6 k-blocks), filled two slabs ahead of their use (L2 latency ~ one slab);
  //   one LDS-only barrier per slab (a __syncthreads() would also drain the prefetches: one memory round trip per slab).
  u32x4_t xr[2][2];
  auto gload = [&](int set, int k0) {
    xr[set][0] = gld16(p.xhat, g0 + (unsigned)k0 * 2u);
    if constexpr (CPT == 2) xr[set][1] = gld16(p.xhat, g1 + (unsigned)k0 * 2u);
  };
  auto sstore = [&](int set, int buf) {
    *reinterpret_cast<u32x4_t*>(&Xs[buf * BT * XPITCH + sr * XPITCH + sc]) = xr[set][0];
    if constexpr (CPT == 2) *reinterpret_cast<u32x4_t*>(&Xs[buf * BT * XPITCH + (sr + 64) * XPITCH + sc]) = xr[set][1];
  };
  f32x4_t acc[RF][3];
  zero_acc<RF>(acc);
  const int nkb = Din / 32, nslab = Din / 64;
  const bf16x8_t* wp = reinterpret_cast<const bf16x8_t*>(p.win + (long)wave * nkb * 3 * 512) + lane;
  bf16x8_t w[6][3];  // ring over k-blocks: slab s uses slots 2 (s % 3) + kk
  auto wload = [&](int s, auto slot3) {
    constexpr int S3 = decltype(slot3)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int b = 0; b < 3; ++b) w[2 * S3 + kk][b] = wp[((s * 2 + kk) * 3 + b) * 64];
  };
  wload(0, std::integral_constant<int, 0>{});
  if (nslab > 1) wload(1, std::integral_constant<int, 1>{});
  gload(0, 0);
  sstore(0, 0);
  if (nslab > 1) gload(1, 64);
  if (nslab > 2) gload(0, 128);
  __syncthreads();
  const bf16_t* arow = Xs + (lane & 15) * XPITCH + (lane >> 4) * 8;
  auto slab = [&](int s, auto j6) {
    constexpr int J = decltype(j6)::value, PAR = J & 1, S3 = J % 3;
    if (s + 2 < nslab) wload(s + 2, std::integral_constant<int, (S3 + 2) % 3>{});
    __builtin_amdgcn_sched_barrier(0);
    const bf16_t* ab = arow + PAR * BT * XPITCH;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int a = 0; a < RF; ++a) {
        const bf16x8_t xf = *reinterpret_cast<const bf16x8_t*>(ab + a * 16 * XPITCH + kk * 32);
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = COOT_MFMA_16x16x32(w[2 * S3 + kk][b], xf, acc[a][b]);
      }
    // slab s + 1 (register set 1 - PAR) -> the other LDS buffer (its readers finished before the last barrier), then that
    // set takes the loads of slab s + 3
    if (s + 1 < nslab) sstore(1 - PAR, 1 - PAR);
    if (s + 3 < nslab) gload(1 - PAR, (s + 3) * 64);
    lds_barrier();
  };
#pragma unroll 1
  for (int s = 0; s < nslab; s += 6) {
    slab(s, std::integral_constant<int, 0>{});
    if (s + 1 < nslab) slab(s + 1, std::integral_constant<int, 1>{});
    if (s + 2 < nslab) slab(s + 2, std::integral_constant<int, 2>{});
    if (s + 3 < nslab) slab(s + 3, std::integral_constant<int, 3>{});
    if (s + 4 < nslab) slab(s + 4, std::integral_constant<int, 4>{});
    if (s + 5 < nslab) slab(s + 5, std::integral_constant<int, 5>{});
  }
  stamp();
  TileRing<RF> R;  // weight ring of the three QKV passes; the first one's k-blocks fly during the input FC's epilogue
  gemm_issue(R, p.wqkv + wave * GSZ, lane);
  // ---- epilogue: + folded bias, save h0, GELU, + pe -> z0 (tile + global) ------------------------------------------
  struct PrePe { f32x4_t a, b; };
  epilogue<RF, 8>(acc, Stg, As, row0, Bsm,
      [&](int row, int col) {
        const int pos = p.pos ? (row < p.T ? p.pos[row] : 0) : (row < p.T0 ? row % p.L1 : (row - p.T0) % p.L2);
        const float* pp = p.pe + (long)pos * FZ_D + col;
        return PrePe{*reinterpret_cast<const f32x4_t*>(pp), *reinterpret_cast<const f32x4_t*>(pp + 4)};
      },
      [&](int row, int col, float (&v)[8], const PrePe& pr, int) {
        gst16(p.h0, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
        v[0] += pr.a[0]; v[1] += pr.a[1]; v[2] += pr.a[2]; v[3] += pr.a[3]; v[4] += pr.b[0]; v[5] += pr.b[1]; v[6] += pr.b[2]; v[7] += pr.b[3];
        gst16(p.z0, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, true);
  stamp();
  // ---- QKV ------------------------------------------------------------------------------------------------------------------
#pragma unroll 1
  for (int q = 0; q < 3; ++q) {
    zero_acc<RF>(acc);
    gemm_run<RF, 12, true>(R, As, p.wqkv + (q * 8 + wave) * GSZ, p.wqkv + ((q < 2 ? q + 1 : q) * 8 + wave) * GSZ, acc, lane);
    epilogue<RF, 8>(acc, Stg, As, row0, Bsm + (1 + q) * FZ_D, [&](int, int) { return PreNone{}; },
        [&](int row, int col, float (&v)[8], const PreNone&, int) { gst16(p.qkv, (unsigned)(row * (3 * FZ_D) + q * FZ_D + col) * 2u, pack8(v)); },
        false);
    stamp();
  }
}

// ---- QKV projection (forward) and its dX (backward) on full-width token tiles ---------------------------------------
__global__ __launch_bounds__(512) void qkv_fwd_kernel(QkvFwd p) {
  constexpr int RF = 8, BT = 128, RR = 32;
  __shared__ __attribute__((aligned(16))) unsigned char smem[BT * APITCH * 2 + RR * SPITCH * 4 + 3 * FZ_D * 4];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  float* Stg = reinterpret_cast<float*>(smem + BT * APITCH * 2);
  float* Bsm = reinterpret_cast<float*>(smem + BT * APITCH * 2 + RR * SPITCH * 4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * BT;
  if (threadIdx.x < 3 * FZ_D / 4) reinterpret_cast<f32x4_t*>(Bsm)[threadIdx.x] = reinterpret_cast<const f32x4_t*>(p.bias)[threadIdx.x];
  TileRing<RF> R;
  gemm_issue(R, p.wqkv + wave * GSZ, lane);
  load_tile<RF>(As, p.z, FZ_D, 0, row0);
  __syncthreads();
  f32x4_t acc[RF][3];
#pragma unroll 1
  for (int q = 0; q < 3; ++q) {
    zero_acc<RF>(acc);
    gemm_run<RF, 12, true>(R, As, p.wqkv + (q * 8 + wave) * GSZ, p.wqkv + ((q < 2 ? q + 1 : q) * 8 + wave) * GSZ, acc, lane);
    epilogue<RF, 8>(acc, Stg, As, row0, Bsm + q * FZ_D, [&](int, int) { return PreNone{}; },
        [&](int row, int col, float (&v)[8], const PreNone&, int) { gst16(p.qkv, (unsigned)(row * (3 * FZ_D) + q * FZ_D + col) * 2u, pack8(v)); },
        false);
  }
}

template <int RF, bool GELU>
__global__ __launch_bounds__(512) void qkv_bwd_kernel(QkvBwd p) {
  constexpr int BT = 16 * RF, RR = 32;
  __shared__ __attribute__((aligned(16))) unsigned char smem[BT * APITCH * 2 + RR * SPITCH * 4];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem);
  float* Stg = reinterpret_cast<float*>(smem + BT * APITCH * 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * BT, T = p.T;
  f32x4_t acc[RF][3];
  zero_acc<RF>(acc);
  const bf16_t* wg = p.wqkv + (long)wave * (36L * 3 * 512);  // one 48-column group at K = 1152
  TileRing<RF> R;
  gemm_issue(R, wg, lane);
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    if (c) __syncthreads();  // every wave is done with the previous 384-wide slab
    load_tile<RF>(As, p.dqkv, 3 * FZ_D, c * FZ_D, row0);
    __syncthreads();
    gemm_run<RF, 12, true>(R, As, wg + (long)c * 12 * 3 * 512, wg + (long)(c < 2 ? c + 1 : c) * 12 * 3 * 512, acc, lane);
  }
  float cs[3][8];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[i][j] = 0.f;
  struct PreTwo { u32x4_t res, aux; };
  epilogue<RF, 8>(acc, Stg, As, row0, nullptr,
      [&](int row, int col) {
        PreTwo pr;
        pr.res = gld16(p.res, (unsigned)(row * FZ_D + col) * 2u);
        if constexpr (GELU) pr.aux = gld16(p.aux, (unsigned)(row * FZ_D + col) * 2u);
        return pr;
      },
      [&](int row, int col, float (&v)[8], const PreTwo& pr, int i) {
        float r[8];
        unpack8(pr.res, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += r[j];
        if constexpr (GELU) {
          float a[8];
          unpack8(pr.aux, a);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(a[j]);
          if (row < T) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cs[i][j] += v[j];
          }
        }
        gst16(p.dz, (unsigned)(row * FZ_D + col) * 2u, pack8(v));
      }, false);
  if constexpr (GELU) colsum_flush(cs, Stg, p.part + (long)blockIdx.x * FZ_D);
}

// out[c] += sum_tile part[tile][c]
// 16 columns x 16 tile groups per workgroup: the loop over the tiles is a chain of dependent-latency loads, so the more
// groups share it the shorter it gets (4 groups over 200 tiles: 22 us; 16 groups, unrolled: the launch floor)
__device__ __forceinline__ float colsum_groups(const float* src, long ld, int ntiles, bool on) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, pg = threadIdx.x >> 4;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (on) {
    int t = pg;
    for (; t + 48 < ntiles; t += 64) {
      v0 += src[(long)t * ld]; v1 += src[(long)(t + 16) * ld]; v2 += src[(long)(t + 32) * ld]; v3 += src[(long)(t + 48) * ld];
    }
    for (; t < ntiles; t += 16) v0 += src[(long)t * ld];
  }
  red[pg][cl] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  float r = 0.f;
  if (pg == 0) {
#pragma unroll
    for (int g = 0; g < 16; ++g) r += red[g][cl];
  }
  return r;
}

__global__ __launch_bounds__(256) void colsum_tiles_kernel(const float* part, int ntiles, int n, float* out, int overwrite) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  const float r = colsum_groups(part + c, n, ntiles, c < n);
  if ((threadIdx.x >> 4) == 0 && c < n) out[c] = overwrite ? r : out[c] + r;
}

// out[seg][c] += sum_tile part[tile][off + c]: the bias / LayerNorm gradients of pre_attn_bwd_kernel
struct ScatterSeg { float* dst; int off, n; };
struct ScatterArgs { ScatterSeg s[8]; int nseg, ntiles; const float* part; };
__global__ __launch_bounds__(256) void colsum_scatter_kernel(ScatterArgs a) {
  const ScatterSeg& sg = a.s[blockIdx.y];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  const bool on = c < sg.n && sg.dst;
  const float r = colsum_groups(a.part + sg.off + c, FZ_BWD_NCS, a.ntiles, on);
  if ((threadIdx.x >> 4) == 0 && on) sg.dst[c] += r;
}

}  // namespace

// Tile height of the token-tile chains.  One workgroup per CU: a launch of <= 256 tiles is ONE wave of workgroups whose duration is
// the per-tile latency, however few tiles there are.  Up to 16 384 tokens (the text side of every shipped config, ragged batches
// with packed rows, small batches) 64-row tiles still fit in one wave and each takes about half as long; above, 128-row tiles
// (half the weight traffic per token) stay one wave up to 32 768 tokens.
// (round 5 measured the forward kernels of the 25 600-token video side on 64-row tiles too — 5 % slower, profiles/README.md; the
// per-kernel A/B switches "half_tiles_<k>" went in round 6: `kernel` only names the caller)
constexpr int kHalfTilesMax = 256 * 64;
bool half_tiles(int T, int /*kernel*/) { return T <= kHalfTilesMax; }

static int g_fused_attn = 1;  // coot_set_option("fused_attn", 0): the local networks' forward self-attention as launches of their own (attn_short_fwd).
// Round 6: inside post_attn_fwd_kernel for fixed-length sequences of 16 k <= 80 (64 on 64-row tiles) rows — ActivityNet's 80 frames and
// 64 / 16 words; other shapes, packed rows and every backward keep the attention kernels.  First version (wave = fragment, head by head
// through shared LDS, two barriers per head): 1.235 against 1.229 ms per step, not adopted.  Second (wave = head, wave-private K / V
// slices, no workgroup barrier): 1.197 against 1.210 ms, six of six alternating pairs (profiles/r06_ab_fused_attn.txt): on.
void set_fused_attn(int on) { g_fused_attn = on; }
static int g_fused_attn_launches = 0;  // launches that took the attention along (coot_get_option("fused_attn_launches"): tests)
int fused_attn_launches() { return g_fused_attn_launches; }
// tile height launch_post_attn_fwd picks for T tokens (0: the 32-row kernel of small launches, which never takes the attention along)
static int post_attn_tile_rows(int T, bool do_pool) {
  if (!do_pool && T < 1024) return 0;
  return half_tiles(T, 1) ? 64 : 128;
}
bool post_attn_can_fuse_attention(int T, bool do_pool, int N0, int L0, int N1, int L1) {
  if (!g_fused_attn) return false;
  const int BT = post_attn_tile_rows(T, do_pool);
  if (BT == 0) return false;
  const int lmax = BT == 128 ? FZ_ATTN_LMAX(8) : FZ_ATTN_LMAX(4);  // a wave's LDS slice holds K | V of one head of one sequence
  if (N0 <= 0 || L0 <= 0 || L0 % 16 != 0 || L0 > lmax || (long)N0 * L0 + (long)N1 * L1 != (long)T) return false;
  if (N1 > 0 && (L1 <= 0 || L1 % 16 != 0 || L1 > lmax || ((long)N0 * L0) % BT != 0)) return false;
  return true;
}

int launch_post_attn_fwd(const PostAttnFwd& p, hipStream_t st) {
  if (p.attn.on) {
    COOT_REQUIRE(p.attn.qkv && p.attn.lse && p.ctx_w && p.attn.lens0 && (p.attn.N1 == 0 || p.attn.lens1), "post_attn_fwd: fused attention pointers");
    COOT_REQUIRE(post_attn_can_fuse_attention(p.T, p.do_pool != 0, p.attn.N0, p.attn.L0, p.attn.N1, p.attn.L1),
                 "post_attn_fwd: these segments cannot take the attention along (post_attn_can_fuse_attention)");
    COOT_REQUIRE((long)p.T * 3 * FZ_D * 2 < (1l << 32), "post_attn_fwd: qkv beyond the 32-bit byte offsets of the fused attention");
    ++g_fused_attn_launches;
  }
  COOT_REQUIRE((p.ctx || p.attn.on) && p.xres && p.wo && p.w1 && p.w2 && p.bo && p.b1 && p.b2 && p.ln1g && p.ln1b && p.ln2g && p.ln2b && p.r1 && p.z1 &&
               p.h1 && p.a1 && p.r2 && p.z2, "post_attn_fwd: null pointer");
  COOT_REQUIRE(!p.do_pool || (p.pw1 && p.pw2 && p.pb1 && p.pb2 && p.hp && p.ap && p.s), "post_attn_fwd: pooling pointers");
  if (p.T <= 0) return 0;
  // GEMM passes of the chain: out-proj, FF1, FF2 (+ pool FC1 768 wide, FC2 384 wide): 2 * T * 384 * 384 each
  // (+ the in-chain attention's own products: Q K^T and P V, 4 L^2 d_model per sequence)
  const double attn_flops = p.attn.on ? 4.0 * 384.0 * ((double)p.attn.N0 * p.attn.L0 * p.attn.L0 + (double)p.attn.N1 * p.attn.L1 * p.attn.L1) : 0.0;
  void* ts = timing_begin(TIMING_FUSED, 2.0 * p.T * 384.0 * 384.0 * (p.do_pool ? 6.0 : 3.0) + attn_flops, 0, st);
  const bool drop = p.d_postln.thr || p.d_ff1.thr || p.d_ff2.thr || p.d_pool1.thr || p.d_pool2.thr;
  if (drop)
    COOT_REQUIRE(p.d_postln.thr && p.d_ff1.thr && p.d_ff2.thr && (!p.do_pool || (p.d_pool1.thr && p.d_pool2.thr)), "post_attn_fwd: dropout on some sites only");
  // Few tokens (the global networks: one row per clip): 32-token tiles.  The chain is then bound by every workgroup
  // streaming the three weight matrices from L2, not by the MFMAs — but it is ONE launch instead of five dependent ones.
  const bool small = !p.do_pool && p.T < 1024;
  if (small) {
    if (drop) hipLaunchKernelGGL((post_attn_fwd_kernel<2, true>), dim3((p.T + 31) / 32), dim3(NTHR), 0, st, p);
    else hipLaunchKernelGGL((post_attn_fwd_kernel<2, false>), dim3((p.T + 31) / 32), dim3(NTHR), 0, st, p);
  } else if (half_tiles(p.T, 1)) {
    if (drop) hipLaunchKernelGGL((post_attn_fwd_kernel<4, true>), dim3(fz_grid((p.T + 63) / 64)), dim3(NTHR), 0, st, p);
    else hipLaunchKernelGGL((post_attn_fwd_kernel<4, false>), dim3(fz_grid((p.T + 63) / 64)), dim3(NTHR), 0, st, p);
  } else if (drop) {
    hipLaunchKernelGGL((post_attn_fwd_kernel<8, true>), dim3(fz_grid((p.T + 127) / 128)), dim3(NTHR), 0, st, p);
  } else {
    hipLaunchKernelGGL((post_attn_fwd_kernel<8, false>), dim3(fz_grid((p.T + 127) / 128)), dim3(NTHR), 0, st, p);
  }
  timing_end(ts, st);
  COOT_CHECK_LAUNCH("post_attn_fwd");
  return 0;
}

}  // namespace coot

namespace coot {
int launch_pre_attn_bwd(const PreAttnBwd& p_in, hipStream_t st) {
  PreAttnBwd p = p_in;
  COOT_REQUIRE(p.h1 && p.r2 && p.r1 && p.w2 && p.w1 && p.wo && p.ln2g && p.ln1g && p.dr2 && p.dh1 && p.dr1 && p.dctx && p.part,
               "pre_attn_bwd: null pointer");
  COOT_REQUIRE(p.do_pool ? (p.ds && p.dzp && p.hp && p.pw2 && p.pw1 && p.dhp) : (p.dz2 != nullptr), "pre_attn_bwd: input gradient pointers");
  if (p.T <= 0) return 0;
  const bool half = half_tiles(p.T, 2);
  const int tiles = half ? (p.T + 63) / 64 : (p.T + 127) / 128;
  // the per-tile partial rows: kept until the pass's single reduction launch if the caller opened a deferred scope (rowops.h)
  float* top = colsum_defer_room() >= 8 ? partials_workspace_top((size_t)tiles * FZ_BWD_NCS) : nullptr;
  if (top) p.part = top;
  const bool drop = p.d_ff2.thr || p.d_ff1.thr || p.d_postln.thr || p.d_pool1.thr;
  void* ts = timing_begin(TIMING_FUSED, 2.0 * p.T * 384.0 * 384.0 * (p.do_pool ? 6.0 : 3.0), 0, st);
  if (drop) {
    COOT_REQUIRE(p.d_ff2.thr && p.d_ff1.thr && p.d_postln.thr && (!p.do_pool || p.d_pool1.thr) && p.dr2m, "pre_attn_bwd: dropout on some sites only");
    if (half) hipLaunchKernelGGL((pre_attn_bwd_kernel<4, true>), dim3(tiles), dim3(NTHR), 0, st, p);
    else hipLaunchKernelGGL((pre_attn_bwd_kernel<8, true>), dim3(tiles), dim3(NTHR), 0, st, p);
  } else {
    if (half) hipLaunchKernelGGL((pre_attn_bwd_kernel<4, false>), dim3(tiles), dim3(NTHR), 0, st, p);
    else hipLaunchKernelGGL((pre_attn_bwd_kernel<8, false>), dim3(tiles), dim3(NTHR), 0, st, p);
  }
  timing_end(ts, st);
  COOT_CHECK_LAUNCH("pre_attn_bwd");
  ScatterArgs a; a.ntiles = tiles; a.part = p.part; a.nseg = 8;
  a.s[0] = ScatterSeg{p.do_pool ? p.g_pb1 : nullptr, 0, 2 * FZ_D};
  a.s[1] = ScatterSeg{p.g_ln2g, 2 * FZ_D, FZ_D}; a.s[2] = ScatterSeg{p.g_ln2b, 3 * FZ_D, FZ_D}; a.s[3] = ScatterSeg{p.g_b2, 4 * FZ_D, FZ_D};
  a.s[4] = ScatterSeg{p.g_b1, 5 * FZ_D, FZ_D};
  a.s[5] = ScatterSeg{p.g_ln1g, 6 * FZ_D, FZ_D}; a.s[6] = ScatterSeg{p.g_ln1b, 7 * FZ_D, FZ_D}; a.s[7] = ScatterSeg{p.g_bo, 8 * FZ_D, FZ_D};
  if (top) {  // room for all eight was checked above
    for (int i = 0; i < 8; ++i)
      if (a.s[i].dst) colsum_defer_add(a.s[i].dst, p.part + a.s[i].off, FZ_BWD_NCS, tiles, a.s[i].n, 0);
    return 0;
  }
  hipLaunchKernelGGL(colsum_scatter_kernel, dim3(48, 8), dim3(256), 0, st, a);
  COOT_CHECK_LAUNCH("colsum_scatter");
  return 0;
}
}  // namespace coot

namespace coot {
int launch_qkv_fwd(const QkvFwd& p, hipStream_t st) {
  COOT_REQUIRE(p.z && p.wqkv && p.bias && p.qkv, "qkv_fwd: null pointer");
  if (p.T <= 0) return 0;
  void* ts = timing_begin(TIMING_FUSED, 2.0 * p.T * 384.0 * 1152.0, 0, st);
  hipLaunchKernelGGL(qkv_fwd_kernel, dim3((p.T + 127) / 128), dim3(NTHR), 0, st, p);
  timing_end(ts, st);
  COOT_CHECK_LAUNCH("qkv_fwd");
  return 0;
}
int launch_qkv_bwd(const QkvBwd& p_in, hipStream_t st) {
  QkvBwd p = p_in;
  COOT_REQUIRE(p.dqkv && p.wqkv && p.res && p.dz, "qkv_bwd: null pointer");
  COOT_REQUIRE(!p.aux || (p.colsum && p.part), "qkv_bwd: GELU' needs the column-sum buffers");
  if (p.T <= 0) return 0;
  const bool half = half_tiles(p.T, 3);
  const int tiles = half ? (p.T + 63) / 64 : (p.T + 127) / 128;
  bool deferred = false;
  if (p.aux) {  // per-tile column sums -> the pass's single deferred reduction, if a scope is open (rowops.h)
    float* top = partials_workspace_top((size_t)tiles * FZ_D);
    if (top && colsum_defer_add(p.colsum, top, FZ_D, tiles, FZ_D, p.colsum_overwrite)) { p.part = top; deferred = true; }
  }
  void* ts = timing_begin(TIMING_FUSED, 2.0 * p.T * 384.0 * 1152.0, 0, st);
  if (half) {
    if (p.aux) hipLaunchKernelGGL((qkv_bwd_kernel<4, true>), dim3(tiles), dim3(NTHR), 0, st, p);
    else hipLaunchKernelGGL((qkv_bwd_kernel<4, false>), dim3(tiles), dim3(NTHR), 0, st, p);
  } else {
    if (p.aux) hipLaunchKernelGGL((qkv_bwd_kernel<8, true>), dim3(tiles), dim3(NTHR), 0, st, p);
    else hipLaunchKernelGGL((qkv_bwd_kernel<8, false>), dim3(tiles), dim3(NTHR), 0, st, p);
  }
  timing_end(ts, st);
  COOT_CHECK_LAUNCH("qkv_bwd");
  if (p.aux) {
    if (deferred) return 0;
    hipLaunchKernelGGL(colsum_tiles_kernel, dim3((FZ_D + 15) / 16), dim3(256), 0, st, (const float*)p.part, tiles, FZ_D, p.colsum, p.colsum_overwrite);
    COOT_CHECK_LAUNCH("colsum_tiles");
  }
  return 0;
}
}  // namespace coot

namespace coot {
static thread_local int g_glob_xcd_first = 0, g_glob_xcd_count = 8;
void glob_xcd_set(int first, int count) {
  if (first < 0 || count < 1 || first + count > 8) { first = 0; count = 8; }
  g_glob_xcd_first = first; g_glob_xcd_count = count;
}
// grid of a single-launch global pass: slot s of XCD x is block 8 s + x (glob_tile_or_help)
template <typename P>
static int glob_grid(P& p, int tiles) {
  p.tiles = tiles;
  p.xcd_first = g_glob_xcd_first; p.xcd_count = g_glob_xcd_count;
  if (tiles > 8 * p.xcd_count) { p.xcd_first = 0; p.xcd_count = 8; }  // a pass with many tiles needs the CUs of every XCD, not L2 room
  p.tiles_per_xcd = (tiles + p.xcd_count - 1) / p.xcd_count;
  // helper workgroups per XCD (0: the chains fill the chip themselves); the same number of helpers overall on fewer XCDs
  p.warm_per_xcd = tiles >= 128 ? 0 : (tiles > 64 ? 8 : 16) * (8 / p.xcd_count);
  return 8 * (p.tiles_per_xcd + p.warm_per_xcd);
}
bool glob_fwd_supported(int Cmax) { return Cmax >= 1 && Cmax <= 32; }
int launch_glob_fwd(const GlobFwd& p_in, hipStream_t st) {
  COOT_REQUIRE(p_in.x && p_in.lens && p_in.hidden && p_in.pe && p_in.n_gain && p_in.n_bias && p_in.z0 && p_in.cq_in && p_in.pooled, "glob_fwd: null pointer");
  COOT_REQUIRE(glob_fwd_supported(p_in.Cmax), "glob_fwd: %d items per sequence (max 32)", p_in.Cmax);
  if (p_in.B <= 0) return 0;
  GlobFwd p = p_in;
  const int G = 32 / p.Cmax, tiles = (p.B + G - 1) / G;
  const int grid = glob_grid(p, tiles);
  const bool drop = p.self.d_attn.thr || p.self.d_postln.thr || p.self.d_ff1.thr || p.self.d_ff2.thr;
  if (drop) COOT_REQUIRE(p.self.d_attn.thr && p.self.d_postln.thr && p.self.d_ff1.thr && p.self.d_ff2.thr && p.ctx.d_attn.thr && p.ctx.d_postln.thr &&
                         p.ctx.d_ff1.thr && p.ctx.d_ff2.thr, "glob_fwd: dropout on some sites only");
  const double T = (double)p.B * p.Cmax;
  void* ts = timing_begin(TIMING_GLOB, 2.0 * 384.0 * 384.0 * (8.0 * T + 4.0 * p.B), 0, st);
  if (drop) hipLaunchKernelGGL(glob_fwd_kernel<true>, dim3(grid), dim3(NTHR), 0, st, p);
  else hipLaunchKernelGGL(glob_fwd_kernel<false>, dim3(grid), dim3(NTHR), 0, st, p);
  timing_end(ts, st);
  COOT_CHECK_LAUNCH("glob_fwd");
  return 0;
}
int launch_glob_bwd(const GlobBwd& p_in, hipStream_t st) {
  COOT_REQUIRE(p_in.x && p_in.lens && p_in.n_gain && p_in.dpooled && p_in.dhidden, "glob_bwd: null pointer");
  COOT_REQUIRE(glob_fwd_supported(p_in.Cmax), "glob_bwd: %d items per sequence (max 32)", p_in.Cmax);
  if (p_in.B <= 0) return 0;
  GlobBwd p = p_in;
  const int G = 32 / p.Cmax, tiles = (p.B + G - 1) / G;
  const int grid = glob_grid(p, tiles);
  const bool drop = p.self.d_attn.thr || p.self.d_postln.thr || p.self.d_ff1.thr || p.self.d_ff2.thr;
  if (drop) COOT_REQUIRE(p.self.d_attn.thr && p.self.d_postln.thr && p.self.d_ff1.thr && p.self.d_ff2.thr && p.ctx.d_attn.thr && p.ctx.d_postln.thr &&
                         p.ctx.d_ff1.thr && p.ctx.d_ff2.thr && p.self.dr2m && p.ctx.dr2m, "glob_bwd: dropout on some sites only");
  const double T = (double)p.B * p.Cmax;
  void* ts = timing_begin(TIMING_GLOB, 2.0 * 384.0 * 384.0 * (6.0 * T + 6.0 * T + 4.0 * p.B), 0, st);
  if (drop) hipLaunchKernelGGL(glob_bwd_kernel<true>, dim3(grid), dim3(NTHR), 0, st, p);
  else hipLaunchKernelGGL(glob_bwd_kernel<false>, dim3(grid), dim3(NTHR), 0, st, p);
  timing_end(ts, st);
  COOT_CHECK_LAUNCH("glob_bwd");
  return 0;
}
int launch_infc_qkv_fwd(const InfcQkvFwd& p, hipStream_t st) {
  COOT_REQUIRE(p.xhat && p.win && p.bin && p.pe && p.wqkv && p.bqkv && p.h0 && p.z0 && p.qkv, "infc_qkv_fwd: null pointer");
  COOT_REQUIRE(p.Din % 64 == 0 && p.Din >= 128 && p.L1 > 0 && p.L2 > 0, "infc_qkv_fwd: Din = %d must be a multiple of 64, >= 128", p.Din);
  if (p.T <= 0) return 0;
  void* ts = timing_begin(TIMING_FUSED, 2.0 * p.T * 384.0 * (p.Din + 1152.0), 0, st);
  if (half_tiles(p.T, 0)) hipLaunchKernelGGL(infc_qkv_fwd_kernel<4>, dim3(fz_grid((p.T + 63) / 64)), dim3(NTHR), 0, st, p);
  else hipLaunchKernelGGL(infc_qkv_fwd_kernel<8>, dim3(fz_grid((p.T + 127) / 128)), dim3(NTHR), 0, st, p);
  timing_end(ts, st);
  COOT_CHECK_LAUNCH("infc_qkv_fwd");
  return 0;
}
}  // namespace coot

namespace coot {
COOT_DET_DEFINE_SETTER(fused)
}  // namespace coot
