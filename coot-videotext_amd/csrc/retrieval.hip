// Retrieval ranking on the device: validate_epoch's normalisation (coot/trainer_retrieval.py:397-402),
// the similarity matrix d = emb1 . emb2^T and compute_retrieval_cosine (nntrainer/retrieval.py:57-98) for BOTH directions,
// without materialising d and without leaving the GPU.  The reference does this on the host: a D2H copy per batch, an
// fp32 sgemm and one numpy argsort per row (O(N^2 log N), ~30 s per epoch for the 4 917 ActivityNet validation videos).
//
//   rank of item i in row i of d = position of i in argsort(d[i])[::-1] = #{j : d_ij > d_ii}            (no exact ties)
//   ties: counted as ahead of i when j > i (= the reversal of a stable ascending sort; numpy's introsort leaves the order
//         of exact ties unspecified, so no rule can be "the" reference one — exact fp32 ties between different
//         embeddings do not occur on real data)
//
// Everything is integer counting on top of fp32 similarities, so the similarity arithmetic is fixed: one fp32 FMA chain
// per element in k order (chunks of 32), identical in the diagonal pass and in the full pass, and identical to what
// the optional sim output holds — the ranks are bit-exact functions of that matrix (tests/test_retrieval_device.py
// checks them against numpy's argsort on the matrix the kernel wrote).
//
// HBM / VALU bound integer + fp32 work: not reshaped into bf16 MFMA GEMMs (bf16 similarities would reorder near-ties).
#include "../../include/coot_hip.h"
#include "common.h"

namespace coot {
namespace {

constexpr int RT = 64;   // tile: 64 rows of emb1 x 64 rows of emb2
constexpr int RK = 32;   // k chunk
constexpr int RP = RK + 1;

// x / sqrt(sum x^2) per row (coot/trainer_retrieval.py:400-402: no eps), one wave per row; rows [0, N) of a, then of b
__global__ __launch_bounds__(256) void rt_normalize_kernel(const float* a, const float* b, int N, int d, float* na, float* nb) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * N) return;
  const float* src = row < N ? a + (long)row * d : b + (long)(row - N) * d;
  float* dst = row < N ? na + (long)row * d : nb + (long)(row - N) * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) { const float x = src[c]; s += x * x; }
  s = wave_sum(s);
  const float nrm = sqrtf(s);
  for (int c = lane; c < d; c += 64) dst[c] = src[c] / nrm;
}

// One 64 x 64 tile of d: thread (ty, tx) of a 16 x 16 grid owns the 4 x 4 block rows 4 ty .., columns 4 tx ..
// acc[r][c] accumulates in k order, chunk by chunk: the same chain for every element in every pass.
struct Tile { float acc[4][4]; };
__device__ __forceinline__ void tile_dot(const float* A, const float* B, int N, int d, int i0, int j0, float (*As)[RP], float (*Bs)[RP], Tile& t) {
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) t.acc[r][c] = 0.f;
  for (int k0 = 0; k0 < d; k0 += RK) {
    // stage 64 x 32 of each operand: 2048 floats each, 8 per thread; rows beyond N and columns beyond d are zero
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q, r = e >> 5, k = e & 31;
      const bool kin = k0 + k < d;
      As[r][k] = (kin && i0 + r < N) ? A[(long)(i0 + r) * d + k0 + k] : 0.f;
      Bs[r][k] = (kin && j0 + r < N) ? B[(long)(j0 + r) * d + k0 + k] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < RK; ++k) {
      float av[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = As[4 * ty + r][k];
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = Bs[4 * tx + c][k];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) t.acc[r][c] = fmaf(av[r], bv[c], t.acc[r][c]);
    }
    __syncthreads();
  }
}

// pass 1: the diagonal entries d_ii (tile (I, I) only)
__global__ __launch_bounds__(256) void rt_diag_kernel(const float* A, const float* B, int N, int d, float* diag) {
  __shared__ float As[RT][RP], Bs[RT][RP];
  const int i0 = blockIdx.x * RT;
  Tile t;
  tile_dot(A, B, N, d, i0, i0, As, Bs, t);
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  if (ty == tx) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (i0 + 4 * ty + r < N) diag[i0 + 4 * ty + r] = t.acc[r][r];
  }
}

// pass 2: every tile; row counts -> ranks_ab (emb1 -> emb2), column counts -> ranks_ba
__global__ __launch_bounds__(256) void rt_rank_kernel(const float* A, const float* B, int N, int d, const float* diag, float* sim,
                                                      int* ranks_ab, int* ranks_ba) {
  __shared__ float As[RT][RP], Bs[RT][RP];
  __shared__ int rowc[RT], colc[RT];
  const int i0 = blockIdx.y * RT, j0 = blockIdx.x * RT;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  if (tid < RT) { rowc[tid] = 0; colc[tid] = 0; }
  Tile t;
  tile_dot(A, B, N, d, i0, j0, As, Bs, t);  // ends with a barrier: the counters are zeroed
  float di[4], dj[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) di[r] = i0 + 4 * ty + r < N ? diag[i0 + 4 * ty + r] : 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) dj[c] = j0 + 4 * tx + c < N ? diag[j0 + 4 * tx + c] : 0.f;
  int rc[4] = {0, 0, 0, 0}, cc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + 4 * ty + r, j = j0 + 4 * tx + c;
      if (i < N && j < N) {
        const float s = t.acc[r][c];
        if (sim) sim[(long)i * N + j] = s;
        if (i != j) {
          // row i of d: is j ahead of i?   column j of d (= row j of d^T): is i ahead of j?
          if (s > di[r] || (s == di[r] && j > i)) ++rc[r];
          if (s > dj[c] || (s == dj[c] && i > j)) ++cc[c];
        }
      }
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) if (rc[r]) atomicAdd(&rowc[4 * ty + r], rc[r]);
#pragma unroll
  for (int c = 0; c < 4; ++c) if (cc[c]) atomicAdd(&colc[4 * tx + c], cc[c]);
  __syncthreads();
  if (tid < RT) {
    if (i0 + tid < N && rowc[tid]) atomicAdd(ranks_ab + i0 + tid, rowc[tid]);
    if (j0 + tid < N && colc[tid]) atomicAdd(ranks_ba + j0 + tid, colc[tid]);
  }
}

// R@1/5/10/50 (fractions), medr = floor(median) + 1, meanr = mean + 1, sum = r1 + r5 + r50 (nntrainer/retrieval.py:88-97)
// for one direction per workgroup.  hist: [2][N] ints, zeroed.  The median of N integers from their histogram.
__global__ __launch_bounds__(1024) void rt_metrics_kernel(const int* ranks_ab, const int* ranks_ba, int N, int* hist, float* out) {
  __shared__ unsigned long long s_sum;
  __shared__ int s_cnt[4];
  __shared__ int s_scan[1024];
  __shared__ int s_med[2];
  const int* ranks = blockIdx.x == 0 ? ranks_ab : ranks_ba;
  int* h = hist + (long)blockIdx.x * N;
  const int tid = threadIdx.x;
  if (tid == 0) { s_sum = 0ull; s_med[0] = -1; s_med[1] = -1; }
  if (tid < 4) s_cnt[tid] = 0;
  __syncthreads();
  unsigned long long ls = 0ull;
  int lc[4] = {0, 0, 0, 0};
  for (int i = tid; i < N; i += 1024) {
    const int r = ranks[i];
    ls += (unsigned long long)r;
    lc[0] += r < 1; lc[1] += r < 5; lc[2] += r < 10; lc[3] += r < 50;
    atomicAdd(h + r, 1);
  }
  atomicAdd(&s_sum, ls);
#pragma unroll
  for (int q = 0; q < 4; ++q) if (lc[q]) atomicAdd(&s_cnt[q], lc[q]);
  __threadfence();
  __syncthreads();
  // order statistics (N - 1) / 2 and N / 2 (0-based) of the sorted ranks: np.median averages them
  const int per = (N + 1023) / 1024, lo = tid * per, hi = min(N, lo + per);
  int part = 0;
  for (int v = lo; v < hi; ++v) part += __hip_atomic_load(h + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the counts were made with device atomics
  s_scan[tid] = part;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int q = 0; q < 1024; ++q) { const int c = s_scan[q]; s_scan[q] = run; run += c; }
  }
  __syncthreads();
  const int k0 = (N - 1) / 2, k1 = N / 2;
  int run = s_scan[tid];
  for (int v = lo; v < hi; ++v) {
    const int c = __hip_atomic_load(h + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c) {
      if (k0 >= run && k0 < run + c) s_med[0] = v;
      if (k1 >= run && k1 < run + c) s_med[1] = v;
      run += c;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float* o = out + 7 * blockIdx.x;
    const float n = (float)N;
    const float r1 = s_cnt[0] / n, r5 = s_cnt[1] / n, r10 = s_cnt[2] / n, r50 = s_cnt[3] / n;
    const double med = 0.5 * ((double)s_med[0] + (double)s_med[1]);
    o[0] = r1; o[1] = r5; o[2] = r10; o[3] = r50;
    o[4] = (float)(floor(med) + 1.0);
    o[5] = (float)((double)s_sum / (double)N + 1.0);
    o[6] = r1 + r5 + r50;
  }
}

struct Ws { float *na, *nb, *diag; int* hist; size_t bytes; };
Ws layout(void* base, int N, int d) {
  Ws w; size_t off = 0;
  auto take = [&](size_t n) { char* p = base ? (char*)base + off : nullptr; off += (n + 255) & ~(size_t)255; return (void*)p; };
  w.na = (float*)take((size_t)N * d * 4); w.nb = (float*)take((size_t)N * d * 4); w.diag = (float*)take((size_t)N * 4);
  w.hist = (int*)take((size_t)2 * N * 4);
  w.bytes = off;
  return w;
}

}  // namespace
}  // namespace coot

using namespace coot;

extern "C" {

size_t coot_retrieval_workspace_bytes(int N, int d) { return layout(nullptr, N, d).bytes + 256; }

int coot_retrieval_ranks(const float* emb1, const float* emb2, int N, int d, int normalize, int32_t* ranks_12, int32_t* ranks_21,
                         float* metrics, float* sim_out, void* workspace, size_t workspace_bytes, coot_stream_t stream) {
  COOT_REQUIRE(emb1 && emb2 && ranks_12 && ranks_21 && workspace, "retrieval: null pointer");
  COOT_REQUIRE(N >= 1 && d >= 1, "retrieval: N = %d, d = %d", N, d);
  hipStream_t st = (hipStream_t)stream;
  Ws w = layout(workspace, N, d);
  COOT_REQUIRE(w.bytes <= workspace_bytes, "retrieval: workspace too small (%zu < %zu)", workspace_bytes, w.bytes);
  const float *A = emb1, *B = emb2;
  if (normalize) {
    hipLaunchKernelGGL(rt_normalize_kernel, dim3((2 * N + 3) / 4), dim3(256), 0, st, emb1, emb2, N, d, w.na, w.nb);
    COOT_CHECK_LAUNCH("rt_normalize");
    A = w.na; B = w.nb;
  }
  if (int rc = check_hip(hipMemsetAsync(ranks_12, 0, (size_t)N * 4, st), "memset ranks")) return rc;
  if (int rc = check_hip(hipMemsetAsync(ranks_21, 0, (size_t)N * 4, st), "memset ranks")) return rc;
  const int nt = (N + RT - 1) / RT;
  hipLaunchKernelGGL(rt_diag_kernel, dim3(nt), dim3(256), 0, st, A, B, N, d, w.diag);
  COOT_CHECK_LAUNCH("rt_diag");
  hipLaunchKernelGGL(rt_rank_kernel, dim3(nt, nt), dim3(256), 0, st, A, B, N, d, (const float*)w.diag, sim_out, (int*)ranks_12, (int*)ranks_21);
  COOT_CHECK_LAUNCH("rt_rank");
  if (metrics) {
    if (int rc = check_hip(hipMemsetAsync(w.hist, 0, (size_t)2 * N * 4, st), "memset hist")) return rc;
    hipLaunchKernelGGL(rt_metrics_kernel, dim3(2), dim3(1024), 0, st, (const int*)ranks_12, (const int*)ranks_21, N, w.hist, metrics);
    COOT_CHECK_LAUNCH("rt_metrics");
  }
  return 0;
}

}  // extern "C"
