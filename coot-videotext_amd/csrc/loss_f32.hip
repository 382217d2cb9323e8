// fp32 REFERENCE MODE of the contrastive loss (include/coot_hip.h: coot_contrastive_fwd_bwd_f32): the op sequence of
// compute_total_constrastive_loss (coot/trainer_retrieval.py:148-182) — F.normalize of the six sets, ContrastiveLoss.forward
// (coot/loss_fn.py:63-100) for up to nine (X, Y) problems, and their derivatives — in plain fp32 FMA kernels: no MFMA, no bf16
// rounding, nothing fused, every sum in a fixed order (no atomics).  A checker like csrc/ref_f32.hip, never what bench.py times:
// the fast path (loss_fused.hip) computes the similarities on bf16 MFMA operands and puts ~1e-3 into every gradient; the
// difference between the two IS that rounding (tests/test_gpu_f32_mode.py).
//
//   xn = x / max(||x||, 1e-12)                                   F.normalize (torch/nn/functional.py: eps = 1e-12)
//   S = Xn Yn^T, d_i = S_ii
//   cost_s[i][j]  = max(0, m + S_ij - d_i), cost_im[i][j] = max(0, m + S_ij - d_j), both 0 on the diagonal
//   L = w (sum cost_s + sum cost_im) / N^2
//   dL/dS_ij = w / N^2 ([cost_s_ij > 0] + [cost_im_ij > 0])  (i != j)
//   dL/dS_ii = -w / N^2 (#{j != i: cost_s_ij > 0} + #{k != i: cost_im_ki > 0})
//   dXn = G Yn, dYn = G^T Xn;  dx = (dxn - xn <dxn, xn>) / max(||x||, 1e-12)
#include "loss.h"

namespace coot {
namespace {

constexpr float kNormEps = 1e-12f;

// one wave per row: xn, 1 / max(||x||, eps)
__global__ __launch_bounds__(256) void clf_norm_kernel(const float* x, int n, int d, float* xn, float* inv, float* dxn_zero) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const float* xr = x + (long)row * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s = fmaf(xr[c], xr[c], s);
  s = wave_sum(s);
  const float nrm = sqrtf(s), iv = 1.0f / fmaxf(nrm, kNormEps);
  for (int c = lane; c < d; c += 64) {
    xn[(long)row * d + c] = xr[c] * iv;
    if (dxn_zero) dxn_zero[(long)row * d + c] = 0.f;  // the gradient accumulator of this set's normalised rows
  }
  if (lane == 0) inv[row] = iv;
}

// S[i][j] = <X_i, Y_j>, k ascending, one thread per entry (16 x 16 tiles through LDS)
__global__ __launch_bounds__(256) void clf_sim_kernel(const float* X, const float* Y, int n, int d, float* S) {
  __shared__ float xs[16][17], ys[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < d; k0 += 16) {
    const int xi = blockIdx.y * 16 + ty, yj = blockIdx.x * 16 + ty;
    xs[ty][tx] = (xi < n && k0 + tx < d) ? X[(long)xi * d + k0 + tx] : 0.f;
    ys[ty][tx] = (yj < n && k0 + tx < d) ? Y[(long)yj * d + k0 + tx] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(xs[ty][k], ys[tx][k], acc);
    __syncthreads();
  }
  if (i < n && j < n) S[(long)i * n + j] = acc;
}

// per row i: hinge sums and the diagonal gradient; G off the diagonal.  One workgroup (256 threads) per row.
__global__ __launch_bounds__(256) void clf_hinge_kernel(const float* S, int n, float margin, float wn /* w / N^2 */, float* G, float* row_loss) {
  __shared__ float red[256];
  __shared__ int cnt[256];
  const int i = blockIdx.x;
  const float di = S[(long)i * n + i];
  float ls = 0.f;
  int c = 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    if (j == i) continue;
    const float sij = S[(long)i * n + j], sji = S[(long)j * n + i], dj = S[(long)j * n + j];
    const float cs = margin + sij - di, ci = margin + sij - dj;  // entry (i, j): against the row's / the column's own pair
    ls += fmaxf(cs, 0.f) + fmaxf(ci, 0.f);
    G[(long)i * n + j] = wn * ((cs > 0.f ? 1.f : 0.f) + (ci > 0.f ? 1.f : 0.f));
    // the diagonal entry (i, i) collects -1 for every violated cost_s of ROW i and every violated cost_im of COLUMN i
    c += (cs > 0.f ? 1 : 0) + ((margin + sji - di) > 0.f ? 1 : 0);
  }
  red[threadIdx.x] = ls; cnt[threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { red[threadIdx.x] += red[threadIdx.x + s]; cnt[threadIdx.x] += cnt[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    row_loss[i] = red[0];
    G[(long)i * n + i] = -wn * (float)cnt[0];
  }
}

// *loss += wn * sum_i row_loss[i] (one thread, fixed order, double accumulator)
__global__ void clf_loss_kernel(const float* row_loss, int n, float wn, float* loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += (double)row_loss[i];
  *loss += (float)(s * (double)wn);
}

// dA[i][c] += sum_j (TRANS ? G[j][i] : G[i][j]) B[j][c], j ascending; one thread per (i, c)
template <bool TRANS>
__global__ __launch_bounds__(256) void clf_grad_kernel(const float* G, const float* B, int n, int d, float* dA) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= n || c >= d) return;
  float acc = 0.f;
  for (int j = 0; j < n; ++j) acc = fmaf(TRANS ? G[(long)j * n + i] : G[(long)i * n + j], B[(long)j * d + c], acc);
  dA[(long)i * d + c] += acc;
}

// dx += (dxn - xn <dxn, xn>) * inv; one wave per row
__global__ __launch_bounds__(256) void clf_norm_bwd_kernel(const float* xn, const float* dxn, const float* inv, int n, int d, float* dx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s = fmaf(dxn[(long)row * d + c], xn[(long)row * d + c], s);
  s = wave_sum(s);
  const float iv = inv[row];
  for (int c = lane; c < d; c += 64) dx[(long)row * d + c] += (dxn[(long)row * d + c] - xn[(long)row * d + c] * s) * iv;
}

struct F32Layout {
  float* xn[6]; float* dxn[6]; float* inv[6]; float* S; float* G; float* row_loss;
  size_t bytes;
};
F32Layout f32_layout(void* base, int n_high, int n_low, int d_high, int d_low) {
  F32Layout L;
  size_t off = 0;
  auto get = [&](size_t floats) { off = (off + 255) & ~(size_t)255; float* p = base ? (float*)((char*)base + off) : nullptr; off += floats * 4; return p; };
  const int n[6] = {n_high, n_high, n_low, n_low, n_high, n_high}, d[6] = {d_high, d_high, d_low, d_low, d_low, d_low};
  for (int s = 0; s < 6; ++s) { L.xn[s] = get((size_t)n[s] * d[s]); L.dxn[s] = get((size_t)n[s] * d[s]); L.inv[s] = get(n[s]); }
  const size_t nm = (size_t)(n_high > n_low ? n_high : n_low);
  L.S = get(nm * nm); L.G = get(nm * nm); L.row_loss = get(nm);
  L.bytes = off;
  return L;
}

}  // namespace

size_t contrastive_f32_scratch_bytes(int n_high, int n_low, int d_high, int d_low) { return f32_layout(nullptr, n_high, n_low, d_high, d_low).bytes; }

int launch_contrastive_f32(const float* const v[6], float* const dv[6], int n_high, int n_low, int d_high, int d_low, const float w_pair[3],
                           const float w_self[3], float margin, float* loss, void* scratch, size_t scratch_bytes, hipStream_t st) {
  COOT_REQUIRE(n_high >= 1 && n_low >= 1 && d_high >= 1 && d_low >= 1, "contrastive_f32: empty set");
  const F32Layout L = f32_layout(scratch, n_high, n_low, d_high, d_low);
  COOT_REQUIRE(scratch_bytes >= L.bytes, "contrastive_f32: scratch too small (%zu < %zu: coot_contrastive_f32_scratch_bytes)", scratch_bytes, L.bytes);
  const bool bwd = dv[0] != nullptr;
  const int n[6] = {n_high, n_high, n_low, n_low, n_high, n_high}, d[6] = {d_high, d_high, d_low, d_low, d_low, d_low};
  for (int s = 0; s < 6; ++s) {
    hipLaunchKernelGGL(clf_norm_kernel, dim3((n[s] + 3) / 4), dim3(256), 0, st, v[s], n[s], d[s], L.xn[s], L.inv[s], bwd ? L.dxn[s] : (float*)nullptr);
  }
  // the (X, Y, weight) problems in the order of the reference's sum (trainer_retrieval.py:168-182): three alignment terms, then
  // the cluster terms as two self problems each (w_self carries the 1/2 of compute_cluster_loss)
  struct Prob { int a, b; float w; };
  Prob probs[9];
  int np = 0;
  for (int p = 0; p < 3; ++p) if (w_pair[p] != 0.f) probs[np++] = Prob{2 * p, 2 * p + 1, w_pair[p]};
  for (int p = 0; p < 3; ++p) if (w_self[p] != 0.f) { probs[np++] = Prob{2 * p, 2 * p, w_self[p]}; probs[np++] = Prob{2 * p + 1, 2 * p + 1, w_self[p]}; }
  for (int q = 0; q < np; ++q) {
    const Prob& pr = probs[q];
    const int N = n[pr.a], D = d[pr.a];
    const float wn = pr.w / ((float)N * (float)N);
    hipLaunchKernelGGL(clf_sim_kernel, dim3((N + 15) / 16, (N + 15) / 16), dim3(256), 0, st, L.xn[pr.a], L.xn[pr.b], N, D, L.S);
    hipLaunchKernelGGL(clf_hinge_kernel, dim3(N), dim3(256), 0, st, L.S, N, margin, wn, L.G, L.row_loss);
    hipLaunchKernelGGL(clf_loss_kernel, dim3(1), dim3(64), 0, st, L.row_loss, N, wn, loss);
    if (bwd) {
      const dim3 grid((D + 63) / 64, (N + 3) / 4);
      hipLaunchKernelGGL(clf_grad_kernel<false>, grid, dim3(256), 0, st, L.G, L.xn[pr.b], N, D, L.dxn[pr.a]);
      hipLaunchKernelGGL(clf_grad_kernel<true>, grid, dim3(256), 0, st, L.G, L.xn[pr.a], N, D, L.dxn[pr.b]);
    }
  }
  if (bwd) {
    for (int s = 0; s < 6; ++s)
      hipLaunchKernelGGL(clf_norm_bwd_kernel, dim3((n[s] + 3) / 4), dim3(256), 0, st, L.xn[s], L.dxn[s], L.inv[s], n[s], d[s], dv[s]);
  }
  COOT_CHECK_LAUNCH("contrastive_f32");
  return 0;
}

}  // namespace coot
