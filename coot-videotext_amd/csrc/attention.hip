// Attention kernels for gfx950 with v_mfma_f32_16x16x16_bf16 (d_head = 16*k, k <= 4).
//
// Forward / dQ kernels compute S^T = K.Q^T per 16x16 tile so that a lane owns ONE query
// (lane&15) and 4 consecutive keys per register quad: softmax statistics are lane-local plus two
// xor-shuffles, and the S^T accumulator layout IS the B-operand layout of the next MFMA
// (O^T = V^T.P^T, dQ^T = K^T.dS^T) — probabilities never leave registers.
// The dK/dV kernel uses the mirrored orientation (lane owns one key).
// K/V (resp. Q/dO) chunks of 64 rows are staged in LDS both row-major and transposed.
#include "attention.h"
#include "gemm.h"

namespace coot {

// dropout mask of one attention probability: common.h (attn_drop_scale)
__device__ __forceinline__ float attn_drop(unsigned key, unsigned row, int k, unsigned lk_half, unsigned thr, float inv_keep) {
  return attn_drop_scale(key, row, k, lk_half, thr, inv_keep);
}

constexpr int KC = 64;  // rows per staged chunk

template <int DH>
struct AttnSmem {
  static constexpr int RP = DH + 8;  // row-major pitch (elements)
  static constexpr int TP = KC + 8;  // transposed pitch
};

__device__ __forceinline__ f32x4_t mfma16(s16x4_t a, s16x4_t b, f32x4_t c) {
  return COOT_MFMA_16x16x16(a, b, c);
}
__device__ __forceinline__ s16x4_t pack4(float a, float b, float c, float d) {
  unsigned lo = pack2bf(a, b), hi = pack2bf(c, d);
  s16x4_t r;
  r[0] = (short)(lo & 0xFFFF); r[1] = (short)(lo >> 16); r[2] = (short)(hi & 0xFFFF); r[3] = (short)(hi >> 16);
  return r;
}

// stage `rows` rows (row index r0.., valid while < rmax) of a [*, ld] matrix at column c0 into
// row-major LDS tile R[KC][RP] and/or transposed tile T[DH][TP]; out-of-range rows are zero.
template <int DH, bool ROWMAJOR, bool TRANSPOSED>
__device__ __forceinline__ void stage_chunk(const bf16_t* src, long ld, long rowbase, int r0, int rmax, int c0,
                                            bf16_t* R, bf16_t* T) {
  constexpr int RP = AttnSmem<DH>::RP, TP = AttnSmem<DH>::TP;
  constexpr int CPR = DH / 8;  // 16-byte chunks per row
  for (int c = threadIdx.x; c < KC * CPR; c += 256) {
    const int r = c / CPR, cc = (c % CPR) * 8;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (r0 + r < rmax) v = *reinterpret_cast<const u32x4_t*>(src + (rowbase + r0 + r) * ld + c0 + cc);
    if (ROWMAJOR) {
      // RP*2 bytes is a multiple of 16 only when DH % 8 == 0 and (DH+8)*2 % 16 == 0 -> true for DH=16k
      *reinterpret_cast<u32x4_t*>(&R[r * RP + cc]) = v;
    }
    if (TRANSPOSED) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        T[(cc + 2 * j) * TP + r] = (bf16_t)(v[j] & 0xFFFFu);
        T[(cc + 2 * j + 1) * TP + r] = (bf16_t)(v[j] >> 16);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  const unsigned dkey = a.drop.thr ? drop_site_key(a.drop.seed, a.drop.seed_ptr, a.drop.site) : 0u;  // once per kernel, not per element
  constexpr int RP = AttnSmem<DH>::RP, TP = AttnSmem<DH>::TP, NK = DH / 16;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[KC * RP];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[DH * TP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int Lq = a.Lq, Lk = a.Lk;
  const int nvalid = (int)a.lens[n];
  const long qbase = (long)n * Lq, kbase = (long)n * Lk;
  const int lq = lane & 15, lg = lane >> 4;
  const int qrow = q0 + lq;
  const bool qok = qrow < Lq;

  s16x4_t qf[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    qf[ks] = s16x4_t{0, 0, 0, 0};
    if (qok) qf[ks] = *reinterpret_cast<const s16x4_t*>(a.q + (qbase + qrow) * a.ldq + h * DH + ks * 16 + lg * 4);
  }
  f32x4_t oacc[NK];
#pragma unroll
  for (int dt = 0; dt < NK; ++dt) oacc[dt] = f32x4_t{0, 0, 0, 0};
  float m_run = -INFINITY, l_run = 0.f;

  for (int kc = 0; kc < Lk; kc += KC) {
    __syncthreads();
    stage_chunk<DH, true, false>(a.k, a.ldk, kbase, kc, Lk, h * DH, Ks, nullptr);
    stage_chunk<DH, false, true>(a.v, a.ldv, kbase, kc, Lk, h * DH, nullptr, Vt);
    __syncthreads();
    f32x4_t s[4];
    float cmax = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4_t{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        s16x4_t kf = *reinterpret_cast<const s16x4_t*>(&Ks[(kt * 16 + lq) * RP + ks * 16 + lg * 4]);
        s[kt] = mfma16(kf, qf[ks], s[kt]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kidx = kc + kt * 16 + lg * 4 + i;
        float v = s[kt][i] * a.scale;
        if (kidx >= nvalid) v = kMaskFill;   // masked_fill(mask, -INF) (transformer_legacy.py:544)
        if (kidx >= Lk) v = -INFINITY;       // beyond the padded length: does not exist
        s[kt][i] = v;
        cmax = fmaxf(cmax, v);
      }
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 16, 64));
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
    const float m_new = fmaxf(m_run, cmax);
    const float alpha = __expf(m_run - m_new);
    float csum = 0.f;
    s16x4_t pf[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      float p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[i] = __expf(s[kt][i] - m_new);
        csum += p[i];
        if (a.drop.thr) {
          const int kidx = kc + kt * 16 + lg * 4 + i;
          p[i] *= attn_drop(dkey, (unsigned)((n * a.H + h) * Lq + qrow), kidx, (unsigned)(Lk + 1) >> 1, a.drop.thr, a.drop.inv_keep);
        }
      }
      pf[kt] = pack4(p[0], p[1], p[2], p[3]);
    }
    csum += __shfl_xor(csum, 16, 64);
    csum += __shfl_xor(csum, 32, 64);
    l_run = l_run * alpha + csum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < NK; ++dt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) oacc[dt][i] *= alpha;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s16x4_t vf = *reinterpret_cast<const s16x4_t*>(&Vt[(dt * 16 + lq) * TP + kt * 16 + lg * 4]);
        oacc[dt] = mfma16(vf, pf[kt], oacc[dt]);
      }
    }
  }
  if (qok) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int dt = 0; dt < NK; ++dt) {
      u32x2_t pk = {pack2bf(oacc[dt][0] * inv, oacc[dt][1] * inv), pack2bf(oacc[dt][2] * inv, oacc[dt][3] * inv)};
      *reinterpret_cast<u32x2_t*>(a.o + (qbase + qrow) * a.ldo + h * DH + dt * 16 + lg * 4) = pk;
    }
    if (lg == 0 && a.lse) a.lse[(qbase + qrow) * a.H + h] = m_run + __logf(l_run);
  }
}

// ---------------------------------------------------------------------------------------------
// backward, part 1: dQ (and delta = rowsum(dO * O)), one wave per 16 queries
// ---------------------------------------------------------------------------------------------
template <int DH>
__device__ __forceinline__ void attn_bwd_q_body(const AttnArgs& a, const int bx) {
  const unsigned dkey = a.drop.thr ? drop_site_key(a.drop.seed, a.drop.seed_ptr, a.drop.site) : 0u;  // once per kernel, not per element
  constexpr int RP = AttnSmem<DH>::RP, TP = AttnSmem<DH>::TP, NK = DH / 16;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[KC * RP];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[KC * RP];
  __shared__ __attribute__((aligned(16))) bf16_t Kt[DH * TP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.z, h = blockIdx.y;
  const int q0 = bx * 64 + wave * 16;
  const int Lq = a.Lq, Lk = a.Lk;
  const int nvalid = (int)a.lens[n];
  const long qbase = (long)n * Lq, kbase = (long)n * Lk;
  const int lq = lane & 15, lg = lane >> 4;
  const int qrow = q0 + lq;
  const bool qok = qrow < Lq;

  s16x4_t qf[NK], dof[NK];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    qf[ks] = s16x4_t{0, 0, 0, 0}; dof[ks] = qf[ks];
    if (qok) {
      const long off = h * DH + ks * 16 + lg * 4;
      qf[ks] = *reinterpret_cast<const s16x4_t*>(a.q + (qbase + qrow) * a.ldq + off);
      dof[ks] = *reinterpret_cast<const s16x4_t*>(a.dout + (qbase + qrow) * a.lddo + off);
      s16x4_t of = *reinterpret_cast<const s16x4_t*>(a.o + (qbase + qrow) * a.ldo + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) dl += bf2f((bf16_t)dof[ks][j]) * bf2f((bf16_t)of[j]);
    }
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  float lse = 0.f;
  if (qok) {
    lse = a.lse[(qbase + qrow) * a.H + h];
    if (lg == 0) a.delta[(qbase + qrow) * a.H + h] = dl;
  }
  f32x4_t dqacc[NK];
#pragma unroll
  for (int dt = 0; dt < NK; ++dt) dqacc[dt] = f32x4_t{0, 0, 0, 0};

  for (int kc = 0; kc < Lk; kc += KC) {
    __syncthreads();
    stage_chunk<DH, true, true>(a.k, a.ldk, kbase, kc, Lk, h * DH, Ks, Kt);
    stage_chunk<DH, true, false>(a.v, a.ldv, kbase, kc, Lk, h * DH, Vs, nullptr);
    __syncthreads();
    s16x4_t dsf[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4_t s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        s16x4_t kf = *reinterpret_cast<const s16x4_t*>(&Ks[(kt * 16 + lq) * RP + ks * 16 + lg * 4]);
        s16x4_t vf = *reinterpret_cast<const s16x4_t*>(&Vs[(kt * 16 + lq) * RP + ks * 16 + lg * 4]);
        s = mfma16(kf, qf[ks], s);
        dp = mfma16(vf, dof[ks], dp);
      }
      float ds[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kidx = kc + kt * 16 + lg * 4 + i;
        float sv = s[i] * a.scale;
        if (kidx >= nvalid) sv = kMaskFill;
        float p = (kidx < Lk && qok) ? __expf(sv - lse) : 0.f;
        float dpe = dp[i];
        if (a.drop.thr) {
          dpe *= attn_drop(dkey, (unsigned)((n * a.H + h) * Lq + qrow), kidx, (unsigned)(Lk + 1) >> 1, a.drop.thr, a.drop.inv_keep);
        }
        ds[i] = (kidx < nvalid) ? p * (dpe - dl) * a.scale : 0.f;  // masked_fill blocks the gradient
      }
      dsf[kt] = pack4(ds[0], ds[1], ds[2], ds[3]);
    }
#pragma unroll
    for (int dt = 0; dt < NK; ++dt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s16x4_t ktf = *reinterpret_cast<const s16x4_t*>(&Kt[(dt * 16 + lq) * TP + kt * 16 + lg * 4]);
        dqacc[dt] = mfma16(ktf, dsf[kt], dqacc[dt]);
      }
  }
  if (qok) {
#pragma unroll
    for (int dt = 0; dt < NK; ++dt) {
      u32x2_t pk = {pack2bf(dqacc[dt][0], dqacc[dt][1]), pack2bf(dqacc[dt][2], dqacc[dt][3])};
      *reinterpret_cast<u32x2_t*>(a.dq + (qbase + qrow) * a.lddq + h * DH + dt * 16 + lg * 4) = pk;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward, part 2: dK, dV, one wave per 16 keys, loop over query chunks
// ---------------------------------------------------------------------------------------------
template <int DH>
__device__ __forceinline__ void attn_bwd_kv_body(const AttnArgs& a, const int bx) {
  const unsigned dkey = a.drop.thr ? drop_site_key(a.drop.seed, a.drop.seed_ptr, a.drop.site) : 0u;  // once per kernel, not per element
  constexpr int RP = AttnSmem<DH>::RP, TP = AttnSmem<DH>::TP, NK = DH / 16;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[KC * RP];
  __shared__ __attribute__((aligned(16))) bf16_t dOs[KC * RP];
  __shared__ __attribute__((aligned(16))) bf16_t Qt[DH * TP];
  __shared__ __attribute__((aligned(16))) bf16_t dOt[DH * TP];
  __shared__ float lse_s[KC], delta_s[KC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.z, h = blockIdx.y;
  const int k0 = bx * 64 + wave * 16;
  const int Lq = a.Lq, Lk = a.Lk;
  const int nvalid = (int)a.lens[n];
  const long qbase = (long)n * Lq, kbase = (long)n * Lk;
  const int lk = lane & 15, lg = lane >> 4;
  const int krow = k0 + lk;
  const bool kin = krow < Lk;
  const bool kvalid = krow < nvalid;

  s16x4_t kf[NK], vf[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    kf[ks] = s16x4_t{0, 0, 0, 0}; vf[ks] = kf[ks];
    if (kin) {
      kf[ks] = *reinterpret_cast<const s16x4_t*>(a.k + (kbase + krow) * a.ldk + h * DH + ks * 16 + lg * 4);
      vf[ks] = *reinterpret_cast<const s16x4_t*>(a.v + (kbase + krow) * a.ldv + h * DH + ks * 16 + lg * 4);
    }
  }
  f32x4_t dkacc[NK], dvacc[NK];
#pragma unroll
  for (int dt = 0; dt < NK; ++dt) { dkacc[dt] = f32x4_t{0, 0, 0, 0}; dvacc[dt] = dkacc[dt]; }

  for (int qc = 0; qc < Lq; qc += KC) {
    __syncthreads();
    stage_chunk<DH, true, true>(a.q, a.ldq, qbase, qc, Lq, h * DH, Qs, Qt);
    stage_chunk<DH, true, true>(a.dout, a.lddo, qbase, qc, Lq, h * DH, dOs, dOt);
    if (threadIdx.x < KC) {
      const int qr = qc + threadIdx.x;
      lse_s[threadIdx.x] = qr < Lq ? a.lse[(qbase + qr) * a.H + h] : 0.f;
    }
    {  // delta_q = <dO_q, O_q> over this head, recomputed here (4 threads per query row) instead of read from the query-tile
       // blocks' output: the key tiles then depend on nothing those blocks write and both run in ONE launch
      const int ql = threadIdx.x >> 2, part = threadIdx.x & 3, qr = qc + ql;
      float dl = 0.f;
      if (qr < Lq) {
        const bf16_t* dop = a.dout + (qbase + qr) * a.lddo + h * DH + part * (DH / 4);
        const bf16_t* op = a.o + (qbase + qr) * a.ldo + h * DH + part * (DH / 4);
#pragma unroll
        for (int e = 0; e < DH / 4; e += 4) {
          const s16x4_t d4 = *reinterpret_cast<const s16x4_t*>(dop + e);
          const s16x4_t o4 = *reinterpret_cast<const s16x4_t*>(op + e);
#pragma unroll
          for (int j = 0; j < 4; ++j) dl += bf2f((bf16_t)d4[j]) * bf2f((bf16_t)o4[j]);
        }
      }
      dl += __shfl_xor(dl, 1, 64);
      dl += __shfl_xor(dl, 2, 64);
      if (part == 0) delta_s[ql] = dl;
    }
    __syncthreads();
    s16x4_t pf[4], dsf[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x4_t s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        s16x4_t qfr = *reinterpret_cast<const s16x4_t*>(&Qs[(qt * 16 + lk) * RP + ks * 16 + lg * 4]);
        s16x4_t dofr = *reinterpret_cast<const s16x4_t*>(&dOs[(qt * 16 + lk) * RP + ks * 16 + lg * 4]);
        s = mfma16(qfr, kf[ks], s);     // s[i]  = S[q = qt*16 + lg*4 + i][key = lk]
        dp = mfma16(dofr, vf[ks], dp);  // dp[i] = dP[q][key]
      }
      float p[4], ds[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ql = qt * 16 + lg * 4 + i, qr = qc + ql;
        float sv = s[i] * a.scale;
        float pv = (kvalid && qr < Lq) ? __expf(sv - lse_s[ql]) : 0.f;
        float dsc = 1.f;
        if (a.drop.thr) {
          dsc = attn_drop(dkey, (unsigned)((n * a.H + h) * Lq + qr), krow, (unsigned)(Lk + 1) >> 1, a.drop.thr, a.drop.inv_keep);
        }
        p[i] = pv * dsc;
        ds[i] = pv * (dp[i] * dsc - delta_s[ql]) * a.scale;
      }
      pf[qt] = pack4(p[0], p[1], p[2], p[3]);
      dsf[qt] = pack4(ds[0], ds[1], ds[2], ds[3]);
    }
#pragma unroll
    for (int dt = 0; dt < NK; ++dt)
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        s16x4_t dotf = *reinterpret_cast<const s16x4_t*>(&dOt[(dt * 16 + lk) * TP + qt * 16 + lg * 4]);
        s16x4_t qtf = *reinterpret_cast<const s16x4_t*>(&Qt[(dt * 16 + lk) * TP + qt * 16 + lg * 4]);
        dvacc[dt] = mfma16(pf[qt], dotf, dvacc[dt]);   // [key = lg*4+i][d = dt*16 + lk]
        dkacc[dt] = mfma16(dsf[qt], qtf, dkacc[dt]);
      }
  }
#pragma unroll
  for (int dt = 0; dt < NK; ++dt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = k0 + lg * 4 + i;
      if (kr < Lk) {
        a.dk[(kbase + kr) * a.lddk + h * DH + dt * 16 + lk] = f2bf(dkacc[dt][i]);
        a.dv[(kbase + kr) * a.lddv + h * DH + dt * 16 + lk] = f2bf(dvacc[dt][i]);
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Short self-attention (L <= 128 keys = queries; every sequence of the shipped configs: 80 frames, <= 64 words, <= 27
// clips): ONE workgroup per (sequence, head), one wave per 16 queries / keys.  K, V (and Q, dO in the backward) of the
// head are staged once, row-major, with 16-byte copies; the transposed MFMA operands (V^T, K^T, Q^T, dO^T) are
// formed by the LDS transpose read ds_read_b64_tr_b16 — no transposed LDS images, no 2-byte LDS writes.  All score
// tiles of a wave stay in registers: single-pass softmax, and the backward is one launch that produces dQ, dK and dV
// (dK^T / dV^T orientation: a lane owns 4 consecutive head channels of one key: 8-byte stores).
// The chunked kernels above (64 queries per workgroup, 64-key chunks, three launches for the backward) ran the
// 80-frame sequences at 2 workgroups per head with 3/8 of the waves idle and re-staged K/V per workgroup.
// ---------------------------------------------------------------------------------------------
constexpr int SH_MAXW = 8;  // waves per workgroup = ceil(L / 16)

__device__ __forceinline__ s16x4_t tr16(const bf16_t* tile, int pitch, int lane) {
  // tile -> element (row 0, col 0) of a 16 x 16 block; result lane l, element j = block[4 (l >> 4) + j][l & 15]
  const int g = lane >> 4, p = lane & 15;
  const bf16_t* addr = tile + (4 * g + (p >> 2)) * pitch + (p & 3) * 4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(addr));
}

// Workgroup -> (sequence, head), XCD aware: workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.
// The heads of one sequence read interleaved 96-byte runs of the same q | k | v rows (and write interleaved runs of dq | dk
// | dv), so they belong on ONE XCD, next to each other in time: id = 64 (n / 8) + 8 h + n % 8.  With (head, sequence) as
// the grid, id % 8 = h: every 128-byte line was fetched (and partially written) by up to 8 different L2s.
__device__ __forceinline__ bool short_wg(const AttnArgs& a, int& n, int& h) {
  const int id = blockIdx.x, g = id / (8 * a.H), r = id - g * (8 * a.H);
  if (a.xcd_order) { h = r >> 3; n = g * 8 + (r & 7); }
  else { h = id % a.H; n = id / a.H; }
  return n < a.Nseq + a.Nseq2;
}
static inline int short_grid(const AttnArgs& a) { return ((a.Nseq + a.Nseq2 + 7) / 8) * 8 * a.H; }
// sequence n of the launch -> its segment: length, valid keys, first token row, dropout key
struct ShortSeq { int n, L, nvalid; long base; unsigned dkey; };
__device__ __forceinline__ ShortSeq short_seq(const AttnArgs& a, int n) {
  ShortSeq s;
  if (a.cu) {  // packed rows
    const bool second = n >= a.Nseq;
    s.n = second ? n - a.Nseq : n;
    s.L = s.nvalid = (int)(second ? a.lens2[s.n] : a.lens[s.n]);
    s.base = a.cu[n];
    s.dkey = a.drop.thr ? drop_site_key(a.drop.seed + (second ? a.seed2_delta : 0ull), a.drop.seed_ptr, a.drop.site) : 0u;
    return s;
  }
  if (n >= a.Nseq) {
    s.n = n - a.Nseq; s.L = a.L2; s.nvalid = (int)a.lens2[s.n]; s.base = (long)a.Nseq * a.Lk + (long)s.n * a.L2;
    s.dkey = a.drop.thr ? drop_site_key(a.drop.seed + a.seed2_delta, a.drop.seed_ptr, a.drop.site) : 0u;
  } else {
    s.n = n; s.L = a.Lk; s.nvalid = (int)a.lens[n]; s.base = (long)n * a.Lk;
    s.dkey = a.drop.thr ? drop_site_key(a.drop.seed, a.drop.seed_ptr, a.drop.site) : 0u;
  }
  return s;
}

template <int DH>
__device__ __forceinline__ void stage_rows(const bf16_t* src, long ld, long rowbase, int L, int L16, int c0, bf16_t* R) {
  constexpr int RP = AttnSmem<DH>::RP, CPR = DH / 8;
  for (int c = threadIdx.x; c < L16 * CPR; c += blockDim.x) {
    const int r = c / CPR, cc = (c % CPR) * 8;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (r < L) v = *reinterpret_cast<const u32x4_t*>(src + (rowbase + r) * ld + c0 + cc);
    *reinterpret_cast<u32x4_t*>(&R[r * RP + cc]) = v;
  }
}

// The same in two halves — loads of every operand first, LDS stores after — so that a workgroup pays ONE memory round trip for
// its staging instead of one per loop iteration and operand (the launcher gives a workgroup 4 threads per padded row and a row
// has DH / 8 <= 8 chunks: at most two chunks per thread).
template <int DH>
__device__ __forceinline__ void stage_load(const bf16_t* src, long ld, long rowbase, int L, int L16, int c0, u32x4_t (&v)[2]) {
  constexpr int CPR = DH / 8;
  static_assert(CPR <= 8, "two chunks per thread cover a row block only up to 8 chunks per row");
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + i * blockDim.x, r = c / CPR, cc = (c % CPR) * 8;
    v[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (c < L16 * CPR && r < L) v[i] = *reinterpret_cast<const u32x4_t*>(src + (rowbase + r) * ld + c0 + cc);
  }
}
template <int DH>
__device__ __forceinline__ void stage_store(bf16_t* R, int L16, const u32x4_t (&v)[2]) {
  constexpr int RP = AttnSmem<DH>::RP, CPR = DH / 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + i * blockDim.x, r = c / CPR, cc = (c % CPR) * 8;
    if (c < L16 * CPR) *reinterpret_cast<u32x4_t*>(&R[r * RP + cc]) = v[i];
  }
}

template <int DH>
__global__ __launch_bounds__(512) void attn_short_fwd_kernel(AttnArgs a) {
  constexpr int RP = AttnSmem<DH>::RP, NK = DH / 16;
  extern __shared__ __attribute__((aligned(16))) bf16_t sh_lds[];  // K | V, L16 rows each (sized by the launcher)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int n, h;
  if (!short_wg(a, n, h)) return;
  const ShortSeq sq = short_seq(a, n);
  n = sq.n;
  const unsigned dkey = sq.dkey;
  const int L = sq.L, L16 = nw * 16;
  bf16_t* Ks = sh_lds;
  bf16_t* Vs = sh_lds + L16 * RP;
  const int nvalid = sq.nvalid;
  const long base = sq.base;
  const int lq = lane & 15, lg = lane >> 4;
  const int qrow = wave * 16 + lq;
  const bool qok = qrow < L;
  u32x4_t kst[2], vst[2];
  stage_load<DH>(a.k, a.ldk, base, L, L16, h * DH, kst);
  stage_load<DH>(a.v, a.ldv, base, L, L16, h * DH, vst);
  s16x4_t qf[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    qf[ks] = s16x4_t{0, 0, 0, 0};
    if (qok) qf[ks] = *reinterpret_cast<const s16x4_t*>(a.q + (base + qrow) * a.ldq + h * DH + ks * 16 + lg * 4);
  }
  stage_store<DH>(Ks, L16, kst);
  stage_store<DH>(Vs, L16, vst);
  __syncthreads();
  f32x4_t s[SH_MAXW];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < SH_MAXW; ++kt) {
    if (kt < nw) {
      s[kt] = f32x4_t{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const s16x4_t kf = *reinterpret_cast<const s16x4_t*>(&Ks[(kt * 16 + lq) * RP + ks * 16 + lg * 4]);
        s[kt] = mfma16(kf, qf[ks], s[kt]);  // s[i] = S^T[key kt*16 + lg*4 + i][query lq]
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kidx = kt * 16 + lg * 4 + i;
        float v = s[kt][i] * a.scale;
        if (kidx >= nvalid) v = kMaskFill;  // masked_fill(mask, -INF) (transformer_legacy.py:544)
        if (kidx >= L) v = -INFINITY;       // beyond the padded length: does not exist
        s[kt][i] = v;
        mx = fmaxf(mx, v);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
  s16x4_t pf[SH_MAXW];
#pragma unroll
  for (int kt = 0; kt < SH_MAXW; ++kt) {
    if (kt < nw) {
      float p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[i] = __expf(s[kt][i] - mx);
        sum += p[i];
        if (a.drop.thr) {
          const int kidx = kt * 16 + lg * 4 + i;
          p[i] *= attn_drop(dkey, (unsigned)((n * a.H + h) * L + qrow), kidx, (unsigned)(L + 1) >> 1, a.drop.thr, a.drop.inv_keep);
        }
      }
      pf[kt] = pack4(p[0], p[1], p[2], p[3]);
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  f32x4_t oacc[NK];
#pragma unroll
  for (int dt = 0; dt < NK; ++dt) {
    oacc[dt] = f32x4_t{0, 0, 0, 0};
#pragma unroll
    for (int kt = 0; kt < SH_MAXW; ++kt)
      if (kt < nw) oacc[dt] = mfma16(tr16(&Vs[kt * 16 * RP + dt * 16], RP, lane), pf[kt], oacc[dt]);  // A = V^T [dh][key]
  }
  if (qok) {
    const float inv = 1.0f / sum;
#pragma unroll
    for (int dt = 0; dt < NK; ++dt) {
      const u32x2_t pk = {pack2bf(oacc[dt][0] * inv, oacc[dt][1] * inv), pack2bf(oacc[dt][2] * inv, oacc[dt][3] * inv)};
      *reinterpret_cast<u32x2_t*>(a.o + (base + qrow) * a.ldo + h * DH + dt * 16 + lg * 4) = pk;
    }
    if (lg == 0 && a.lse) a.lse[(base + qrow) * a.H + h] = mx + __logf(sum);
  }
}

template <int DH>
__global__ __launch_bounds__(512) void attn_short_bwd_kernel(AttnArgs a) {
  constexpr int RP = AttnSmem<DH>::RP, NK = DH / 16;
  extern __shared__ __attribute__((aligned(16))) bf16_t sh_lds[];  // K | V | Q | dO (L16 rows each) | lse, delta (fp32)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int n, h;
  if (!short_wg(a, n, h)) return;
  const ShortSeq sq = short_seq(a, n);
  n = sq.n;
  const unsigned dkey = sq.dkey;
  const int L = sq.L, L16 = nw * 16;
  bf16_t* Ks = sh_lds;
  bf16_t* Vs = Ks + L16 * RP;
  bf16_t* Qs = Vs + L16 * RP;
  bf16_t* dOs = Qs + L16 * RP;
  float* lse_s = reinterpret_cast<float*>(dOs + L16 * RP);
  float* delta_s = lse_s + L16;
  const int nvalid = sq.nvalid;
  const long base = sq.base;
  const int l15 = lane & 15, lg = lane >> 4;
  // every global load of the workgroup is issued here, before the first wait: the four staged operands, this wave's query-side
  // fragments of q / dO / O and its lse (one memory round trip; staged one loop and one operand at a time they were nine)
  const int qrow = wave * 16 + l15;
  const bool qok = qrow < L;
  u32x4_t kst[2], vst[2], qst[2], dst[2];
  stage_load<DH>(a.k, a.ldk, base, L, L16, h * DH, kst);
  stage_load<DH>(a.v, a.ldv, base, L, L16, h * DH, vst);
  stage_load<DH>(a.q, a.ldq, base, L, L16, h * DH, qst);
  stage_load<DH>(a.dout, a.lddo, base, L, L16, h * DH, dst);
  s16x4_t qf[NK], dof[NK], of[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    qf[ks] = s16x4_t{0, 0, 0, 0}; dof[ks] = qf[ks]; of[ks] = qf[ks];
    if (qok) {
      const long off = h * DH + ks * 16 + lg * 4;
      qf[ks] = *reinterpret_cast<const s16x4_t*>(a.q + (base + qrow) * a.ldq + off);
      dof[ks] = *reinterpret_cast<const s16x4_t*>(a.dout + (base + qrow) * a.lddo + off);
      of[ks] = *reinterpret_cast<const s16x4_t*>(a.o + (base + qrow) * a.ldo + off);
    }
  }
  float lse = 0.f;
  if (qok) lse = a.lse[(base + qrow) * a.H + h];
  stage_store<DH>(Ks, L16, kst);
  stage_store<DH>(Vs, L16, vst);
  stage_store<DH>(Qs, L16, qst);
  stage_store<DH>(dOs, L16, dst);
  // ---- phase A: this wave's 16 queries: delta, dQ ------------------------------------------------------------
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < NK; ++ks)
#pragma unroll
    for (int j = 0; j < 4; ++j) dl += bf2f((bf16_t)dof[ks][j]) * bf2f((bf16_t)of[ks][j]);
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);
  if (lg == 0) { lse_s[wave * 16 + l15] = lse; delta_s[wave * 16 + l15] = dl; }
  __syncthreads();
  {
    s16x4_t dsf[SH_MAXW];
#pragma unroll
    for (int kt = 0; kt < SH_MAXW; ++kt) {
      if (kt < nw) {
        f32x4_t sv4 = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
          const s16x4_t kf = *reinterpret_cast<const s16x4_t*>(&Ks[(kt * 16 + l15) * RP + ks * 16 + lg * 4]);
          const s16x4_t vf = *reinterpret_cast<const s16x4_t*>(&Vs[(kt * 16 + l15) * RP + ks * 16 + lg * 4]);
          sv4 = mfma16(kf, qf[ks], sv4);
          dp = mfma16(vf, dof[ks], dp);
        }
        float ds[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kidx = kt * 16 + lg * 4 + i;
          float sv = sv4[i] * a.scale;
          if (kidx >= nvalid) sv = kMaskFill;
          const float p = (kidx < L && qok) ? __expf(sv - lse) : 0.f;
          float dpe = dp[i];
          if (a.drop.thr) {
            dpe *= attn_drop(dkey, (unsigned)((n * a.H + h) * L + qrow), kidx, (unsigned)(L + 1) >> 1, a.drop.thr, a.drop.inv_keep);
          }
          ds[i] = (kidx < nvalid) ? p * (dpe - dl) * a.scale : 0.f;  // masked_fill blocks the gradient
        }
        dsf[kt] = pack4(ds[0], ds[1], ds[2], ds[3]);
      }
    }
#pragma unroll
    for (int dt = 0; dt < NK; ++dt) {
      f32x4_t dq = {0, 0, 0, 0};
#pragma unroll
      for (int kt = 0; kt < SH_MAXW; ++kt)
        if (kt < nw) dq = mfma16(tr16(&Ks[kt * 16 * RP + dt * 16], RP, lane), dsf[kt], dq);  // A = K^T [dh][key]
      if (qok) {
        const u32x2_t pk = {pack2bf(dq[0], dq[1]), pack2bf(dq[2], dq[3])};
        *reinterpret_cast<u32x2_t*>(a.dq + (base + qrow) * a.lddq + h * DH + dt * 16 + lg * 4) = pk;
      }
    }
  }
  // ---- phase B: this wave's 16 keys: dK, dV (transposed products: a lane owns 4 channels of key l15) ---------------
  const int krow = wave * 16 + l15;
  const bool kin = krow < L, kvalid = krow < nvalid;
  s16x4_t kf[NK], vf[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    kf[ks] = *reinterpret_cast<const s16x4_t*>(&Ks[krow * RP + ks * 16 + lg * 4]);
    vf[ks] = *reinterpret_cast<const s16x4_t*>(&Vs[krow * RP + ks * 16 + lg * 4]);
  }
  f32x4_t dkacc[NK], dvacc[NK];
#pragma unroll
  for (int dt = 0; dt < NK; ++dt) { dkacc[dt] = f32x4_t{0, 0, 0, 0}; dvacc[dt] = dkacc[dt]; }
#pragma unroll
  for (int qt = 0; qt < SH_MAXW; ++qt) {
    if (qt < nw) {
      f32x4_t sv4 = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const s16x4_t qfr = *reinterpret_cast<const s16x4_t*>(&Qs[(qt * 16 + l15) * RP + ks * 16 + lg * 4]);
        const s16x4_t dofr = *reinterpret_cast<const s16x4_t*>(&dOs[(qt * 16 + l15) * RP + ks * 16 + lg * 4]);
        sv4 = mfma16(qfr, kf[ks], sv4);  // sv4[i] = S[q = qt*16 + lg*4 + i][key = l15]
        dp = mfma16(dofr, vf[ks], dp);
      }
      float p[4], ds[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int qr = qt * 16 + lg * 4 + i;
        const float pv = (kvalid && qr < L) ? __expf(sv4[i] * a.scale - lse_s[qr]) : 0.f;
        float dsc = 1.f;
        if (a.drop.thr) {
          dsc = attn_drop(dkey, (unsigned)((n * a.H + h) * L + qr), krow, (unsigned)(L + 1) >> 1, a.drop.thr, a.drop.inv_keep);
        }
        p[i] = pv * dsc;
        ds[i] = pv * (dp[i] * dsc - delta_s[qr]) * a.scale;
      }
      const s16x4_t pf = pack4(p[0], p[1], p[2], p[3]), dsf = pack4(ds[0], ds[1], ds[2], ds[3]);  // B operands: [k = query][n = key]
#pragma unroll
      for (int dt = 0; dt < NK; ++dt) {
        dvacc[dt] = mfma16(tr16(&dOs[qt * 16 * RP + dt * 16], RP, lane), pf, dvacc[dt]);   // dV^T[dh][key] = dO^T . P
        dkacc[dt] = mfma16(tr16(&Qs[qt * 16 * RP + dt * 16], RP, lane), dsf, dkacc[dt]);   // dK^T[dh][key] = Q^T . dS
      }
    }
  }
  if (kin) {
#pragma unroll
    for (int dt = 0; dt < NK; ++dt) {
      const u32x2_t pk = {pack2bf(dkacc[dt][0], dkacc[dt][1]), pack2bf(dkacc[dt][2], dkacc[dt][3])};
      const u32x2_t pv = {pack2bf(dvacc[dt][0], dvacc[dt][1]), pack2bf(dvacc[dt][2], dvacc[dt][3])};
      *reinterpret_cast<u32x2_t*>(a.dk + (base + krow) * a.lddk + h * DH + dt * 16 + lg * 4) = pk;
      *reinterpret_cast<u32x2_t*>(a.dv + (base + krow) * a.lddv + h * DH + dt * 16 + lg * 4) = pv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// One-query attention (the context block of the global networks, nntrainer/models/transformer_legacy.py:251-267: the local
// network's context vector attends to <= 64 clip / sentence tokens): ONE wave per (sequence, head), lane = key, everything in
// fp32 registers from the bf16 q / k / v.  The backward recomputes the probabilities and takes delta = sum_k P_k dP_k from the
// very products it differentiates, so sum_k dS_k = 0 to fp32 round-off.  (delta = <dO, O> with the bf16-rounded O leaves a
// residue eps; dq = sum_k dS_k K_k then carries eps * mean(K), and the keys of a global network share a large common component:
// measured cos 0.96 against the reference on the q / k projection gradients of tf_context at the benchmark shapes.)
// ---------------------------------------------------------------------------------------------
template <int DH>
__device__ __forceinline__ void load_row(const bf16_t* p, float (&r)[DH]) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) {
    const s16x4_t v = *reinterpret_cast<const s16x4_t*>(p + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[c + j] = bf2f((bf16_t)v[j]);
  }
}
template <int DH>
__device__ __forceinline__ void store_row(bf16_t* p, const float (&r)[DH]) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) {
    const u32x2_t pk = {pack2bf(r[c], r[c + 1]), pack2bf(r[c + 2], r[c + 3])};
    *reinterpret_cast<u32x2_t*>(p + c) = pk;
  }
}
template <int DH>
__global__ __launch_bounds__(256) void attn_q1_fwd_kernel(AttnArgs a) {
  const int lane = threadIdx.x & 63, pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= a.Nseq * a.H) return;
  const int n = pair / a.H, h = pair - n * a.H, Lk = a.Lk, nvalid = (int)a.lens[n];
  const unsigned dkey = a.drop.thr ? drop_site_key(a.drop.seed, a.drop.seed_ptr, a.drop.site) : 0u;
  float qv[DH], kv[DH];
  load_row<DH>(a.q + (long)n * a.ldq + h * DH, qv);
  float sv = 0.f;
  if (lane < Lk) {
    load_row<DH>(a.k + ((long)n * Lk + lane) * a.ldk + h * DH, kv);
#pragma unroll
    for (int c = 0; c < DH; ++c) sv += qv[c] * kv[c];
    sv *= a.scale;
    if (lane >= nvalid) sv = kMaskFill;
  }
  const float m = wave_max(lane < Lk ? sv : -3.0e38f);
  const float e = lane < Lk ? __expf(sv - m) : 0.f;
  const float sum = wave_sum(e);
  float p = e / sum;
  if (a.drop.thr && lane < Lk) p *= attn_drop(dkey, (unsigned)(n * a.H + h), lane, (unsigned)(Lk + 1) >> 1, a.drop.thr, a.drop.inv_keep);
  float o = 0.f;
  const bf16_t* vcol = a.v + (long)n * Lk * a.ldv + h * DH + (lane < DH ? lane : 0);
  for (int l = 0; l < Lk; ++l) o += __shfl(p, l, 64) * bf2f(vcol[(long)l * a.ldv]);
  if (lane < DH) a.o[(long)n * a.ldo + h * DH + lane] = f2bf(o);
  if (lane == 0) a.lse[(long)n * a.H + h] = m + __logf(sum);
}

template <int DH>
__global__ __launch_bounds__(256) void attn_q1_bwd_kernel(AttnArgs a) {
  const int lane = threadIdx.x & 63, pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= a.Nseq * a.H) return;
  const int n = pair / a.H, h = pair - n * a.H, Lk = a.Lk, nvalid = (int)a.lens[n];
  const unsigned dkey = a.drop.thr ? drop_site_key(a.drop.seed, a.drop.seed_ptr, a.drop.site) : 0u;
  float qv[DH], dov[DH], row[DH];
  load_row<DH>(a.q + (long)n * a.ldq + h * DH, qv);
  load_row<DH>(a.dout + (long)n * a.lddo + h * DH, dov);
  const float lse = a.lse[(long)n * a.H + h];
  float p = 0.f, dp = 0.f, dr = 1.f;
  if (lane < Lk) {
    load_row<DH>(a.k + ((long)n * Lk + lane) * a.ldk + h * DH, row);
    float sv = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) sv += qv[c] * row[c];
    sv *= a.scale;
    if (lane >= nvalid) sv = kMaskFill;
    p = __expf(sv - lse);
    load_row<DH>(a.v + ((long)n * Lk + lane) * a.ldv + h * DH, row);
#pragma unroll
    for (int c = 0; c < DH; ++c) dp += dov[c] * row[c];
    if (a.drop.thr) dr = attn_drop(dkey, (unsigned)(n * a.H + h), lane, (unsigned)(Lk + 1) >> 1, a.drop.thr, a.drop.inv_keep);
  }
  const float pd = p * dr;                       // dropped probability: what multiplied V in the forward
  const float delta = wave_sum(pd * dp);         // = <dO, O> in exact arithmetic
  const float ds = (lane < nvalid) ? p * (dr * dp - delta) * a.scale : 0.f;  // masked_fill blocks the gradient
  if (lane < Lk) {
#pragma unroll
    for (int c = 0; c < DH; ++c) row[c] = ds * qv[c];
    store_row<DH>(a.dk + ((long)n * Lk + lane) * a.lddk + h * DH, row);
#pragma unroll
    for (int c = 0; c < DH; ++c) row[c] = pd * dov[c];
    store_row<DH>(a.dv + ((long)n * Lk + lane) * a.lddv + h * DH, row);
  }
  float dq = 0.f;
  const bf16_t* kcol = a.k + (long)n * Lk * a.ldk + h * DH + (lane < DH ? lane : 0);
  for (int l = 0; l < Lk; ++l) dq += __shfl(ds, l, 64) * bf2f(kcol[(long)l * a.ldk]);
  if (lane < DH) a.dq[(long)n * a.lddq + h * DH + lane] = f2bf(dq);
}
static bool attn_q1_ok(const AttnArgs& a) { return a.Lq == 1 && a.Lk >= 1 && a.Lk <= 64 && a.Nseq2 == 0; }

static bool attn_short_ok(const AttnArgs& a) {
  return a.Lq == a.Lk && a.Lk <= 16 * SH_MAXW && a.Lk >= 1 && (a.Nseq2 == 0 || (a.L2 >= 1 && a.L2 <= 16 * SH_MAXW));
}
bool attn_short_path(int L) { return L >= 1 && L <= 16 * SH_MAXW; }

template <int DH>
static int attn_fwd_t(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  a.xcd_order = get_xcd_order() & 4;
  if (attn_q1_ok(a)) {
    hipLaunchKernelGGL(attn_q1_fwd_kernel<DH>, dim3((a.Nseq * a.H + 3) / 4), dim3(256), 0, st, a);
    COOT_CHECK_LAUNCH("attn_q1_fwd");
    return 0;
  }
  if (attn_short_ok(a)) {
    const int nw = ((a.Nseq2 > 0 && a.L2 > a.Lk ? a.L2 : a.Lk) + 15) / 16;
    hipLaunchKernelGGL(attn_short_fwd_kernel<DH>, dim3(short_grid(a)), dim3(64 * nw), (size_t)2 * nw * 16 * AttnSmem<DH>::RP * sizeof(bf16_t), st, a);
    COOT_CHECK_LAUNCH("attn_short_fwd");
    return 0;
  }
  dim3 grid((a.Lq + 63) / 64, a.H, a.Nseq);
  hipLaunchKernelGGL(attn_fwd_kernel<DH>, grid, dim3(256), 0, st, a);
  COOT_CHECK_LAUNCH("attn_fwd");
  return 0;
}
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a, int nq) {
  if ((int)blockIdx.x < nq) attn_bwd_q_body<DH>(a, (int)blockIdx.x);
  else attn_bwd_kv_body<DH>(a, (int)blockIdx.x - nq);
}
template <int DH>
static int attn_bwd_t(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  a.xcd_order = get_xcd_order() & 4;
  if (attn_q1_ok(a)) {
    hipLaunchKernelGGL(attn_q1_bwd_kernel<DH>, dim3((a.Nseq * a.H + 3) / 4), dim3(256), 0, st, a);
    COOT_CHECK_LAUNCH("attn_q1_bwd");
    return 0;
  }
  if (attn_short_ok(a)) {
    const int nw = ((a.Nseq2 > 0 && a.L2 > a.Lk ? a.L2 : a.Lk) + 15) / 16;
    hipLaunchKernelGGL(attn_short_bwd_kernel<DH>, dim3(short_grid(a)), dim3(64 * nw),
                       (size_t)4 * nw * 16 * AttnSmem<DH>::RP * sizeof(bf16_t) + (size_t)2 * nw * 16 * sizeof(float), st, a);
    COOT_CHECK_LAUNCH("attn_short_bwd");
    return 0;
  }
  // dQ tiles and dK/dV tiles in ONE launch (x < nq: query tiles, else key tiles; the key tiles recompute delta themselves) — on
  // the global networks' context layer (1 query per video) the two were dependent 7 us launches on the critical path
  const int nq = (a.Lq + 63) / 64, nk = (a.Lk + 63) / 64;
  hipLaunchKernelGGL(attn_bwd_kernel<DH>, dim3(nq + nk, a.H, a.Nseq), dim3(256), 0, st, a, nq);
  COOT_CHECK_LAUNCH("attn_bwd");
  return 0;
}

static int attn_check(const AttnArgs& a) {
  COOT_REQUIRE(a.q && a.k && a.v && a.o && a.lens, "attention: null pointer");
  COOT_REQUIRE(a.dh % 16 == 0 && a.dh >= 16 && a.dh <= 64, "attention: d_head=%d unsupported (16,32,48,64)", a.dh);
  COOT_REQUIRE(a.ldq % 4 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0, "attention: strides must be multiples of 8");
  return 0;
}

int launch_attn_fwd(const AttnArgs& a, hipStream_t st) {
  if (int rc = attn_check(a)) return rc;
  if (a.Nseq <= 0 || a.Lq <= 0) return 0;
  switch (a.dh) {
    case 16: return attn_fwd_t<16>(a, st);
    case 32: return attn_fwd_t<32>(a, st);
    case 48: return attn_fwd_t<48>(a, st);
    default: return attn_fwd_t<64>(a, st);
  }
}

int launch_attn_bwd(const AttnArgs& a, hipStream_t st) {
  if (int rc = attn_check(a)) return rc;
  COOT_REQUIRE(a.dout && a.lse && a.delta && a.dq && a.dk && a.dv, "attention bwd: null pointer");
  COOT_REQUIRE(a.ldq % 8 == 0 && a.lddo % 8 == 0, "attention bwd: strides must be multiples of 8");
  if (a.Nseq <= 0 || a.Lq <= 0) return 0;
  switch (a.dh) {
    case 16: return attn_bwd_t<16>(a, st);
    case 32: return attn_bwd_t<32>(a, st);
    case 48: return attn_bwd_t<48>(a, st);
    default: return attn_bwd_t<64>(a, st);
  }
}

}  // namespace coot
