// Common device helpers for the COOT retrieval hot path on gfx950 (CDNA4, wave64).
// No CUDA compatibility layer: this code is written for MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace coot {

// ---- the 16-bit operand type of the MFMA path ---------------------------------------------------------------------------------
// The library is built twice from these sources (csrc/build.sh): libcoot_hip.so with bfloat16 operands (the default, what bench.py
// times) and libcoot_hip_f16.so (-DCOOT_OPERAND_F16) with IEEE half operands — the arithmetic of the reference's GPU path (fp16
// autocast, coot/trainer_retrieval.py:264; BASELINE.json configs[3] "fp16 MFMA path").  Same MFMA rate (v_mfma_f32_16x16x32_f16 /
// _bf16), same storage size, fp32 accumulation in both; everything that knows the format is in this block: the conversions and the
// two MFMA wrappers.  The names keep "bf16" (they mean "the 16-bit operand"); raw bits travel as unsigned short either way.
typedef unsigned short bf16_t;  // raw 16-bit operand bits in HBM / LDS (bfloat16, or IEEE half in the f16 build)
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr float kMaskFill = -32752.0f;  // nntrainer/typext.py:24 (INF), used as -INF mask fill
constexpr float kLnEps = 1e-6f;         // nntrainer/models/normalizations.py:89

#ifndef COOT_OPERAND_F16
#define COOT_OPERAND_IS_F16 0
#define COOT_DTYPE_NATIVE 0  // COOT_DTYPE_BF16 (include/coot_hip.h)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// ---- bf16 <-> f32 (round-to-nearest-even, same as v_cvt_pk_bf16_f32) ----------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// fp32 -> bf16 pairs: the gfx950 hardware conversion (v_cvt_pk_bf16_f32, round to nearest even).  The software form
// (add 0x7FFF + lsb, shift) costs 5 VALU ops per element — with two or three bf16 stores per element in the fused
// epilogues that was a third of their instruction count.
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
// D = A (16 x 32) . B (32 x 16) + C on the matrix pipe, and the 16 x 16 x 16 form of the attention kernels (operands as raw 16-bit lanes)
#define COOT_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define COOT_MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)
#else
#define COOT_OPERAND_IS_F16 1
#define COOT_DTYPE_NATIVE 2  // COOT_DTYPE_F16
typedef _Float16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 bf16x2_t __attribute__((ext_vector_type(2)));
// ---- IEEE half <-> f32 (round-to-nearest-even: v_cvt_f16_f32 / v_cvt_pk_f16_f32; values beyond 65504 become Inf, as under autocast) ----
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ float bflo(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xFFFFu)); }
__device__ __forceinline__ float bfhi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
typedef _Float16 f16x4_op_t __attribute__((ext_vector_type(4)));
#define COOT_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define COOT_MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_op_t, (a)), __builtin_bit_cast(f16x4_op_t, (b)), (c), 0, 0, 0)
#endif
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.0f) & 0xFFFFu); }

// ---- erf GELU (nn.GELU(), nntrainer/models/activations.py:29-30) and derivative -----------------------
// Phi(x) = 0.5 erfc(-x / sqrt 2) with erfc by Abramowitz-Stegun 7.1.25 (|error| <= 2.5e-5 on erf, two orders below the
// bf16 resolution of the stored activations): for z = |x| / sqrt 2, erfc(z) = t (a1 + t (a2 + t a3)) exp(-z^2),
// t = 1 / (1 + 0.47047 z).  One v_rcp_f32 + one v_exp_f32 + 8 full-rate ops; exp(-x^2 / 2) is shared with the
// derivative's density term.  The elementwise work of a fused layer is VALU-issue bound (profiles/README.md), so
// these counts matter as much as the MFMA schedule.
struct GeluParts { float h; float e; };  // h = 0.5 erfc(|x| / sqrt 2) = Phi(-|x|),  e = exp(-x^2 / 2)
__device__ __forceinline__ GeluParts gelu_parts(float x) {
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044f);                 // exp(-x^2/2) = 2^(-x^2/2 * log2 e)
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.33267266f, 1.0f));        // 0.47047 / sqrt 2
  const float poly = t * (0.1740121f + t * (-0.0479399f + t * 0.3739278f));        // 0.5 * (a1, a2, a3)
  return GeluParts{poly * e, e};
}
__device__ __forceinline__ float gelu_f(float x) {
#ifdef FZ_NO_EPI_MATH  // (measurement build of fused.hip only: identity instead of GELU)
  return x;
#endif
  const GeluParts g = gelu_parts(x);
  // x Phi(x) = max(x, 0) - |x| Phi(-|x|): one max and one fma (|x| and the sign are source modifiers) instead of
  // mul + sub + compare + select
  return fmaf(-fabsf(x), g.h, fmaxf(x, 0.0f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const GeluParts g = gelu_parts(x);
  const float phi_cdf = x >= 0.0f ? 1.0f - g.h : g.h;
  return fmaf(x * g.e, 0.39894228040143268f, phi_cdf);
}

// ---- wave64 reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based RNG for dropout: one 32-bit draw per (seed, site, element) ---------------
// (cannot match torch's Philox stream; training parity is statistical)
__host__ __device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned rng_u32(unsigned long long seed, unsigned site, unsigned long long idx) {
  unsigned a = mix32((unsigned)idx ^ (unsigned)(seed));
  unsigned b = mix32((unsigned)(idx >> 32) + site * 0x9E3779B9u + (unsigned)(seed >> 32));
  return mix32(a ^ (b + 0x85ebca6bU + (a << 6) + (a >> 2)));
}
// effective seed = host salt + *device base seed (the device word advances once per training step, so a captured
// HIP graph draws fresh masks on every replay)
__device__ __forceinline__ unsigned long long eff_seed(unsigned long long salt, const unsigned long long* base) {
  return salt + (base ? *base : 0ull);
}
// Dropout masks.  The first version drew one rng_u32 (3 x mix32 = 6 integer multiplies) per element: ~100 SIMD cycles
// per 64 elements, which made the mask generation of one encoder layer cost more than its MFMAs (profiles/README.md).
// Now: key = f(seed, site) once per call site; ONE hash (drop_hash below) per PAIR of consecutive elements (2 q, 2 q + 1), 16 bits each,
// compared with the 16-bit threshold thr >> 16 (mkdrop() quantises p to 1/65536 and derives 1/(1-p) from the quantised
// value, so E[mask * inv_keep] = 1 exactly).  Element index < 2^33.
__host__ __device__ __forceinline__ unsigned drop_key(unsigned long long seed, unsigned site) {
  return mix32((unsigned)seed ^ mix32((unsigned)(seed >> 32) + site * 0x9E3779B9u));
}
// The per-pair mixer: two 24 x 24 -> 32 bit multiply-adds (v_mad_u32_u24) and xor-shifts, 9 issue slots per pair against 15 for
// mix32.  (Measured on gfx950, tools/micro/valu_rate.hip: v_mad_u32_u24 and v_mul_lo_u32 both issue at HALF rate, 4.85 cycles per
// wave-instruction per SIMD against 2.6 for shifts / xors / v_fma_f32 — the gain over mix32 is the instruction count, not the
// multiplier.)  Rate / correlation statistics over 4M consecutive pairs (neighbours, next row, the two halves, keys one bit
// apart) are as good as mix32's (tools: see DESIGN.md).
// (host + device: coot_debug_dropout_scales evaluates the same functions on the host, which is what pins the numpy restatement
// in oracle/dropout_masks.py to THIS source in the CPU suite)
__host__ __device__ __forceinline__ unsigned umul24_hd(unsigned a, unsigned b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return (a & 0xFFFFFFu) * (b & 0xFFFFFFu);
#endif
}
__host__ __device__ __forceinline__ unsigned drop_hash(unsigned t) {
  t = umul24_hd(t, 0xD2B54Bu) + (t >> 16);
  t ^= t >> 13;
  t = umul24_hd(t, 0x95A53Du) + (t >> 11);
  t ^= t >> 16;
  return t;
}
// Dropout mask of attention probability (sequence n, head h, query q, key k): 32-bit index arithmetic (it is only a hash
// input, wrap-around is harmless): the pair (k / 2) of row r = (n H + h) Lq + q shares one hash, k & 1 picks the half.
// The 64-bit element index of the first version cost ~8 VALU instructions per probability (v_mad_u64_u32, 64-bit shifts
// and compares) in kernels that are VALU bound.  All attention kernels (forward and the backward passes, attention.hip and the
// in-tile attention of fused.hip) use this map.
__host__ __device__ __forceinline__ float attn_drop_scale(unsigned key, unsigned row, int k, unsigned lk_half, unsigned thr, float inv_keep) {
  const unsigned h = drop_hash((row * lk_half + ((unsigned)k >> 1)) ^ key);
  const unsigned u = (k & 1) ? (h >> 16) : (h & 0xFFFFu);
  return u >= (thr >> 16) ? inv_keep : 0.0f;
}
__host__ __device__ __forceinline__ unsigned drop_pair(unsigned key, unsigned long long idx) { return drop_hash((unsigned)(idx >> 1) ^ key); }
// keep-scale of one element: 0 if dropped else 1/(1-p)
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned site, unsigned long long idx,
                                            unsigned thr, float inv_keep) {
  const unsigned h = drop_pair(drop_key(seed, site), idx);
  const unsigned u = (idx & 1ull) ? (h >> 16) : (h & 0xFFFFu);
  return u >= (thr >> 16) ? inv_keep : 0.0f;
}
// the same with the site key derived once per kernel (drop_site_key): the mask sites of a kernel share (seed, site), and
// reading *seed_ptr where the mask is drawn costs a global load + s_waitcnt vmcnt(0) per element group
__device__ __forceinline__ unsigned drop_site_key(unsigned long long salt, const unsigned long long* base, unsigned site) {
  return drop_key(eff_seed(salt, base), site);
}
__host__ __device__ __forceinline__ float drop_scale_key(unsigned key, unsigned long long idx, unsigned thr, float inv_keep) {
  const unsigned h = drop_pair(key, idx);
  const unsigned u = (idx & 1ull) ? (h >> 16) : (h & 0xFFFFu);
  return u >= (thr >> 16) ? inv_keep : 0.0f;
}
// keep-scales of N consecutive elements starting at an EVEN index (N even): N / 2 hashes
template <int N>
__device__ __forceinline__ void drop_scales_key(unsigned key, unsigned long long idx0, unsigned thr, float inv_keep, float* sc) {
  const unsigned t16 = thr >> 16;
  const unsigned q0 = (unsigned)(idx0 >> 1);
#pragma unroll
  for (int j = 0; j < N / 2; ++j) {
    const unsigned h = drop_hash((q0 + j) ^ key);
    sc[2 * j] = (h & 0xFFFFu) >= t16 ? inv_keep : 0.0f;
    sc[2 * j + 1] = (h >> 16) >= t16 ? inv_keep : 0.0f;
  }
}
template <int N>
__device__ __forceinline__ void drop_scales(unsigned long long seed, unsigned site, unsigned long long idx0, unsigned thr,
                                            float inv_keep, float* sc) {
  drop_scales_key<N>(drop_key(seed, site), idx0, thr, inv_keep, sc);
}

}  // namespace coot

// ---- host side error plumbing --------------------------------------------------------------
namespace coot {
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
}  // namespace coot

#define COOT_CHECK_LAUNCH(what)                                   \
  do {                                                            \
    int _rc = coot::check_hip(hipGetLastError(), what);           \
    if (_rc) return _rc;                                          \
  } while (0)

#define COOT_REQUIRE(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) { coot::set_error(__VA_ARGS__); return -2; }     \
  } while (0)
