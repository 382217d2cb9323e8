// Cycle-consistency loss kernel for gfx950 (the contrastive loss lives in loss_fused.hip).
#include "loss.h"
#include "det.h"

namespace coot {

// ---- CycleConsistencyLoss (coot/loss_fn.py:143-387), one workgroup per (video, direction) -------
constexpr int CC_MAXC = 64;   // max clips / sentences per video
constexpr int CC_NVD = 16;    // D <= 1024

__global__ __launch_bounds__(256) void cyclecons_kernel(CycleArgs a) {
  __shared__ float dist[CC_MAXC][CC_MAXC + 1];
  __shared__ float alpha[CC_MAXC][CC_MAXC + 1];
  __shared__ float beta_sel[CC_MAXC], alpha_sel[CC_MAXC], dd2[CC_MAXC], ddist[CC_MAXC], dalpha[CC_MAXC];
  __shared__ float wred[4][CC_MAXC];
  __shared__ float lrow[CC_MAXC];
  const int b = blockIdx.x, dir = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = a.D;
  const float* src = dir == 0 ? a.clip + (long)b * a.Cc * D : a.sent + (long)b * a.Cs * D;
  const float* tgt = dir == 0 ? a.sent + (long)b * a.Cs * D : a.clip + (long)b * a.Cc * D;
  const int Csrc = dir == 0 ? a.Cc : a.Cs, Ctgt = dir == 0 ? a.Cs : a.Cc;
  const int ns = (int)(dir == 0 ? a.clip_lens[b] : a.sent_lens[b]);
  const int nt = (int)(dir == 0 ? a.sent_lens[b] : a.clip_lens[b]);
  const float invD = 1.0f / (float)D;

  // (1) proximity = -mean_d (src_i - tgt_j)^2, masked with -INF where source or target is padding
  for (int pidx = wave; pidx < Csrc * Ctgt; pidx += 4) {
    const int i = pidx / Ctgt, j = pidx % Ctgt;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) { float t = src[(long)i * D + d] - tgt[(long)j * D + d]; s += t * t; }
    s = wave_sum(s);
    if (lane == 0) dist[i][j] = (i < ns && j < nt) ? -s * invD : kMaskFill;
  }
  __syncthreads();
  // (2) alpha = softmax_j
  if (threadIdx.x < Csrc) {
    const int i = threadIdx.x;
    float m = -INFINITY;
    for (int j = 0; j < Ctgt; ++j) m = fmaxf(m, dist[i][j]);
    float z = 0.f;
    for (int j = 0; j < Ctgt; ++j) { float e = __expf(dist[i][j] - m); alpha[i][j] = e; z += e; }
    const float iz = 1.f / z;
    for (int j = 0; j < Ctgt; ++j) alpha[i][j] *= iz;
  }
  __syncthreads();
  // (3) soft NN s~_i = sum_j alpha_ij tgt_j (registers), dist2[i][k] = -mean_d (s~_i - src_k)^2 -> dist
  for (int i = wave; i < Csrc; i += 4) {
    float nn[CC_NVD];
#pragma unroll
    for (int v = 0; v < CC_NVD; ++v) {
      const int d = lane + 64 * v;
      float t = 0.f;
      if (d < D) for (int j = 0; j < Ctgt; ++j) t += alpha[i][j] * tgt[(long)j * D + d];
      nn[v] = t;
    }
    for (int k = 0; k < Csrc; ++k) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < CC_NVD; ++v) {
        const int d = lane + 64 * v;
        if (d < D) { float t = nn[v] - src[(long)k * D + d]; s += t * t; }
      }
      s = wave_sum(s);
      if (lane == 0) dist[i][k] = (i < ns && k < ns) ? -s * invD : kMaskFill;
    }
  }
  __syncthreads();
  // (4) beta = softmax_k, mu_i = sum_k beta_ik k, l_i = (mu_i - i)^2 (0 for padded i)
  const int isel = (int)(dir == 0 ? a.idx_clip[b] : a.idx_sent[b]);
  if (threadIdx.x < Csrc) {
    const int i = threadIdx.x;
    float m = -INFINITY;
    for (int k = 0; k < Csrc; ++k) m = fmaxf(m, dist[i][k]);
    float z = 0.f;
    for (int k = 0; k < Csrc; ++k) z += __expf(dist[i][k] - m);
    float mu = 0.f;
    const float iz = 1.f / z;
    for (int k = 0; k < Csrc; ++k) {
      const float bk = __expf(dist[i][k] - m) * iz;
      mu += bk * (float)k;
      if (i == isel) beta_sel[k] = bk;
    }
    const float l = i < ns ? (mu - (float)i) * (mu - (float)i) : 0.f;
    lrow[i] = l;
    float* rows = dir == 0 ? a.rows_clip : a.rows_sent;
    if (rows) rows[(long)b * Csrc + i] = l;
    if (i == isel) {
      acc_add(a.loss, a.weight * a.inv_batch * l);
      // d loss / d mu
      dd2[CC_MAXC - 1] = 0.f;
      wred[0][0] = a.weight * a.inv_batch * 2.f * (mu - (float)i);
    }
    if (i == isel) for (int j = 0; j < Ctgt; ++j) alpha_sel[j] = alpha[i][j];
  }
  if (!a.dclip) return;
  __syncthreads();
  float* dsrc = dir == 0 ? a.dclip + (long)b * a.Cc * D : a.dsent + (long)b * a.Cs * D;
  float* dtgt = dir == 0 ? a.dsent + (long)b * a.Cs * D : a.dclip + (long)b * a.Cc * D;
  const float dmu = wred[0][0];
  __syncthreads();
  // ddist2_k = beta_k (dbeta_k - sum beta dbeta), dbeta_k = dmu * k ; zero for masked k
  if (threadIdx.x == 0) {
    float sb = 0.f;
    for (int k = 0; k < Csrc; ++k) sb += beta_sel[k] * dmu * (float)k;
    for (int k = 0; k < Csrc; ++k) dd2[k] = k < ns ? beta_sel[k] * (dmu * (float)k - sb) : 0.f;
  }
  __syncthreads();
  // per-thread feature slice d = tid + 256*v (D <= 1024)
  float nn1[4], dnn1[4], cvec[4], ksel[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int d = threadIdx.x + 256 * v;
    nn1[v] = 0.f; dnn1[v] = 0.f; cvec[v] = 0.f;
    if (d < D) {
      cvec[v] = src[(long)isel * D + d];
      float t = 0.f;
      for (int j = 0; j < Ctgt; ++j) t += alpha_sel[j] * tgt[(long)j * D + d];
      nn1[v] = t;
      float g = 0.f;
      for (int k = 0; k < Csrc; ++k) {
        const float diff = t - src[(long)k * D + d];
        g += dd2[k] * (-2.f * invD) * diff;
        // ONE add per gradient word and workgroup (the selected row's share waits for its second term below): with the other
        // direction's workgroup a word then receives exactly two addends on top of its zero — a + b = b + a, the sum does not depend
        // on who comes first (three or more addends in arrival order would)
        if (k == isel) ksel[v] = dd2[k] * (2.f * invD) * diff;
        else if (dd2[k] != 0.f) atomicAdd(dsrc + (long)k * D + d, dd2[k] * (2.f * invD) * diff);
      }
      dnn1[v] = g;
    }
  }
  // dalpha_j = <tgt_j, dnn1>
  for (int j = 0; j < Ctgt; ++j) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) { const int d = threadIdx.x + 256 * v; if (d < D) s += tgt[(long)j * D + d] * dnn1[v]; }
    s = wave_sum(s);
    if (lane == 0) wred[wave][j] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sa = 0.f;
    for (int j = 0; j < Ctgt; ++j) { dalpha[j] = wred[0][j] + wred[1][j] + wred[2][j] + wred[3][j]; sa += alpha_sel[j] * dalpha[j]; }
    for (int j = 0; j < Ctgt; ++j) ddist[j] = j < nt ? alpha_sel[j] * (dalpha[j] - sa) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int d = threadIdx.x + 256 * v;
    if (d < D) {
      float gsel = 0.f;
      for (int j = 0; j < Ctgt; ++j) {
        const float diff = cvec[v] - tgt[(long)j * D + d];
        float gt = alpha_sel[j] * dnn1[v] + ddist[j] * (2.f * invD) * diff;
        gsel += ddist[j] * (-2.f * invD) * diff;
        if (gt != 0.f) atomicAdd(dtgt + (long)j * D + d, gt);
      }
      atomicAdd(dsrc + (long)isel * D + d, gsel + ksel[v]);
    }
  }
}

int launch_cyclecons(const CycleArgs& a, hipStream_t st) {
  COOT_REQUIRE(a.clip && a.sent && a.clip_lens && a.sent_lens && a.idx_clip && a.idx_sent && a.loss, "cyclecons: null pointer");
  COOT_REQUIRE(a.Cc <= CC_MAXC && a.Cs <= CC_MAXC && a.D <= 1024, "cyclecons: at most %d clips/sentences per video and D<=1024 (Cc=%d Cs=%d D=%d)", CC_MAXC, a.Cc, a.Cs, a.D);
  COOT_REQUIRE((a.dclip == nullptr) == (a.dsent == nullptr), "cyclecons: dclip/dsent must both be set or both null");
  if (a.B <= 0) return 0;
  hipLaunchKernelGGL(cyclecons_kernel, dim3(a.B, 2), dim3(256), 0, st, a);
  COOT_CHECK_LAUNCH("cyclecons");
  return 0;
}

}  // namespace coot

namespace coot {
COOT_DET_DEFINE_SETTER(loss)
}  // namespace coot
