// Row-wise kernels: COOT LayerNorm (unbiased std, eps on std) fwd/bwd, column sums, weight packing.
#pragma once
#include "common.h"

namespace coot {

struct DropCfg {
  unsigned thr = 0; float inv_keep = 1.0f; unsigned long long seed = 0; unsigned site = 0;
  const unsigned long long* seed_ptr = nullptr;  // device base seed added to `seed`
};

struct LnFwd {
  const void* x = nullptr; int x_f32 = 1; long ldx = 0;  // [R, D]
  const void* x2 = nullptr; int R0 = 0;                  // optional second source: rows [R0, R) are rows 0.. of x2 (two segments, one launch)
  int R = 0, D = 0;
  const float* gain = nullptr; const float* bias = nullptr;  // null gain => xhat only (affine folded elsewhere)
  const float* pe = nullptr; int pe_L = 1;                   // + pe[(row % pe_L) * D + col] after the affine
  bf16_t* y = nullptr; long ldy = 0;                         // bf16 out (optional)
  float* y32 = nullptr; long ldy32 = 0;                      // fp32 out (optional)
  DropCfg drop;                                              // dropout applied to the LN output
  // packed (varlen) output rows: row r of the output is token (r - cu[n]) of sequence n (cu: nseq + 1 packed row starts); the
  // source stays the padded [N, L, D] layout of the batch: sequences [0, N0) are rows n L0 + l of x, the others rows
  // (n - N0) L1 + l of x2.  pos_out[r] = position of row r within its sequence (for the positional encoding further down).
  const int* cu = nullptr; int nseq = 0, N0 = 0, L0 = 0, L1 = 0; int* pos_out = nullptr;
  int src_packed = 0;  // with cu: the SOURCE x is packed as well ([R, D] in cu order, coot_collate_packed): row r reads row r
  int max_wgs = 0;     // > 0: at most this many workgroups walk the rows (a launch that should trickle next to latency-bound kernels)
  int nt = 0;          // streaming (non-temporal) loads of an fp32 source and stores of the bf16 output: the launch runs NEXT TO kernels
                       // that live on L2-resident weights (the next batch's input LayerNorm beside the global networks) and must not evict them
};
int launch_ln_fwd(const LnFwd& p, hipStream_t stream);

struct LnBwd {
  const bf16_t* dy = nullptr; long lddy = 0;   // grad wrt LN output (bf16) ...
  const float* dy32 = nullptr; long lddy32 = 0;  // ... or fp32
  const bf16_t* dy_add = nullptr; long lddy_add = 0;  // optional second grad stream added to dy (bf16)
  const void* x = nullptr; int x_f32 = 0; long ldx = 0;  // saved LN input
  const float* gain = nullptr;
  int R = 0, D = 0;
  bf16_t* dx = nullptr; long lddx = 0;
  float* dx32 = nullptr; long lddx32 = 0;
  bf16_t* dxm = nullptr; long lddxm = 0;       // dx * dropmask(dxm_drop) (for a preceding Linear->Dropout)
  DropCfg dxm_drop; long dxm_drop_ld = 0;
  float* dgain = nullptr; float* dbias = nullptr;  // atomics, [D]
  float* dxcolsum = nullptr;                       // [D]: colsum of dxm (if set) else dx
  float* part_ws = nullptr;                        // internal: [blocks][3][D] partials (set by the launcher)
  DropCfg drop;                                    // dropout that was applied to the LN output in fwd
};
int launch_ln_bwd(const LnBwd& p, hipStream_t stream);

// colsum[c] += sum_r x[r][c]   (bf16 in, fp32 atomics)
int launch_colsum_bf16(const bf16_t* x, long ldx, int R, int C, float* out, hipStream_t stream);

// dst_bf16[r][c] = src_f32[r*lds + c] (optionally transposed: dst[c][r]), optional per-column scale
int launch_cast_weight(const float* src, long lds, int R, int C, bf16_t* dst, long ldd, int transpose,
                       const float* colscale, hipStream_t stream);

// many cast/transpose jobs in ONE launch (the whole bf16 weight pack of a network)
// p48 != 0: the destination is the fragment-major "P48" layout of the fused kernels (fused.h: p48_offset) of the logical
// matrix [n][k] = dst row, dst column (after the optional transpose); ldd is ignored.
struct PackJob { long src_off; long dst_byte_off; int R, C; long lds, ldd; int transpose; int p48; long colscale_off; };
// optional rider of a pack launch: out[n] = b[n] + sum_k W[n,k] v[k] (the folded input-FC bias), one wave per row in extra workgroups
struct PackMatvec { const float* W = nullptr; long ldw = 0; int N = 0, K = 0; const float* v = nullptr; const float* b = nullptr; float* out = nullptr; };
struct PackJobs { int n = 0; PackJob j[56]; PackMatvec mv; };
int launch_pack_jobs(const float* P, void* wpack, const PackJobs& jobs, hipStream_t stream);

// out[n] = b[n] + sum_k W[n,k] * v[k]   (tiny mat-vec; used for the folded input-FC bias)
int launch_matvec_bias(const float* W, long ldw, int N, int K, const float* v, const float* b, float* out,
                       hipStream_t stream);

// input-FC parameter grads from M = dh0^T . xhat (SURVEY/DESIGN: LN affine folded into the FC):
//   dW[n,k] += M[n,k]*g[k] + c[n]*b0[k];  dg[k] += sum_n W[n,k]*M[n,k];  db0[k] += sum_n c[n]*W[n,k]
//   db_in[n] += c[n] (optional: the input-FC bias gradient, c = colsum of dh0)
int launch_infc_param_grads(const float* M, const float* W, const float* g0, const float* b0, const float* c,
                            int N, int K, float* dW, float* dg0, float* db0, float* db_in, hipStream_t stream, int overwrite = 0);

// ---- contention-free column reductions ---------------------------------------------------------------
// Producers write per-workgroup partial sums into a stream-ordered scratch region instead of issuing hundreds of
// fp32 atomics onto the same few addresses (cross-XCD atomics serialise at the memory side: a 40 us GEMM became
// 265 us, profiles/README.md); out[c] += sum_p ws[p*ld + c] is then one tiny kernel.
void set_partials_workspace(float* ws, size_t floats, size_t low_floats = 0);  // thread-local, set by the orchestrator per call;
                                                                              // the first low_floats stay with the immediate users
float* partials_workspace(size_t need_floats);            // nullptr if not available / too small
int launch_reduce_partials(const float* ws, int nparts, long ld, int C, float* out, hipStream_t stream);

// Deferred reductions: inside colsum_defer_begin() .. colsum_defer_flush() a producer keeps its partial rows in a region taken
// from the TOP of the partials workspace (partials_workspace_top) and only RECORDS out[c] (+)= sum_p src[p * ld + c]; flush
// runs all recorded reductions as ONE launch.  A reduction launch between two large kernels costs far more than its 5 us
// (drain, launch, ramp-up): the local backward had three of them.  Outside a scope (or with the table / workspace full)
// colsum_defer_add returns false and the producer launches its reduction itself.
constexpr int DEFER_MAX = 24;
struct DeferSeg { float* dst; const float* src; long ld; int nparts, n, overwrite; };
struct DeferArgs { DeferSeg s[DEFER_MAX]; int nseg; };
#ifdef __HIPCC__
// one workgroup (256 threads) of the deferred column sums: segment `seg`, columns 16 bx .. 16 bx + 15 — 16 columns x 16 partial-row
// groups (a serial walk over the partial rows is a chain of load latencies).  Body of colsum_defer_kernel; also the extra grid rows
// of the weight-gradient split reduction (gemm.hip: tn_batch_flush with take_colsums), which saves the launch.
__device__ __forceinline__ void colsum_defer_block(const DeferArgs& a, int seg, int bx) {
  __shared__ float red[16][17];
  const DeferSeg& sg = a.s[seg];
  const int cl = threadIdx.x & 15, pg = threadIdx.x >> 4, c = bx * 16 + cl;
  if (bx * 16 >= sg.n) return;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (c < sg.n) {
    const float* src = sg.src + c;
    int t = pg;
    for (; t + 48 < sg.nparts; t += 64) {
      v0 += src[(long)t * sg.ld]; v1 += src[(long)(t + 16) * sg.ld]; v2 += src[(long)(t + 32) * sg.ld]; v3 += src[(long)(t + 48) * sg.ld];
    }
    for (; t < sg.nparts; t += 16) v0 += src[(long)t * sg.ld];
  }
  red[pg][cl] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (pg == 0 && c < sg.n) {
    float r = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) r += red[g][cl];
    sg.dst[c] = sg.overwrite ? r : sg.dst[c] + r;
  }
}
#endif
void colsum_defer_begin();
// hands the recorded reductions to a caller that runs them inside a launch of its own (resets the table like a flush); false: none
// recorded, or one of them has more than max_cols columns (nothing is taken then)
bool colsum_defer_take(DeferArgs* out, int max_cols);
float* partials_workspace_top(size_t need_floats);
bool colsum_defer_add(float* dst, const float* src, long ld, int nparts, int n, int overwrite);
int colsum_defer_room();  // free slots of the table (0 outside a scope)
int colsum_defer_flush(hipStream_t stream);
void colsum_defer_end();

int launch_fill_f32(float* p, long n, float v, hipStream_t stream);
// y += a * x
int launch_axpy_f32(float* y, const float* x, long n, float a, hipStream_t stream);

}  // namespace coot
