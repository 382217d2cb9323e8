// Loss kernels: fused contrastive loss (L2 normalisation + bidirectional max-margin ranking + gradients, loss_fused.hip),
// cycle-consistency (CycleConsistencyLoss, loss.hip).  coot/loss_fn.py, coot/trainer_retrieval.py:148-233.
#pragma once
#include "common.h"

namespace coot {

struct CycleArgs {
  const float* clip = nullptr;  // [B, Cc, D] zero padded
  const float* sent = nullptr;  // [B, Cs, D]
  const long long* clip_lens = nullptr;  // [B]
  const long long* sent_lens = nullptr;
  const long long* idx_clip = nullptr;   // [B] sampled position per video (th.multinomial, loss_fn.py:312)
  const long long* idx_sent = nullptr;
  int B = 0, Cc = 0, Cs = 0, D = 0;
  float weight = 0.f;       // loss_cycle_cons
  float inv_batch = 0.f;    // 1 / (global batch size)
  float* loss = nullptr;    // += weight * (L_clip + L_sent) contribution of these B videos
  float* rows_clip = nullptr;  // optional [B, Cc] per-position losses (deterministic parity target)
  float* rows_sent = nullptr;  // optional [B, Cs]
  float* dclip = nullptr;   // [B, Cc, D] += grad (may be null: forward only)
  float* dsent = nullptr;   // [B, Cs, D]
};
int launch_cyclecons(const CycleArgs& a, hipStream_t st);

// compute_total_constrastive_loss in three launches (loss_fused.hip).  v / dv: the six sets in the order vid_emb, par_emb,
// clip_emb, sent_emb, vid_ctx, par_ctx (dv all null = forward only; gradients are ACCUMULATED).  w_pair[p] / w_self[p]:
// alignment / cluster weight of pair p (high, low, context); w_self already carries the 1/2 of compute_cluster_loss.
size_t contrastive_fused_scratch_bytes(int n_high, int n_low, int d_high, int d_low);
void set_cl_col_split(int k);  // 0: column splits of a strip by batch size; k >= 1: forced (set BEFORE sizing the scratch)
void set_cl_small(int on);     // 0: sets that fit the LDS take the three-launch path too (default 1: cl_small_kernel + cl_finish)
// ldv (optional): row strides of the six input sets (default dense).  window (optional) = {high row0, high rows, low row0, low
// rows}: only these rows of the per-video sets (0, 1, 4, 5) / per-clip sets (2, 3) receive gradients, dv[] are compact
// [rows, d] arrays (data parallel: the loss is over the gathered batch, a rank keeps the rows of its own videos).
// blk (optional): the input rows live in the BLOCKS of an all-gather (one block per rank): row i of set s belongs to the rank r
// with row0[level][r] <= i < row0[level][r + 1] (level 0: per-video sets, 1: per-clip sets) and starts at
// blocks + base[s][r] + (i - row0[level][r]) * ldv[s]; v[s] then only serves the window's rows (this rank's own block).
constexpr int CL_MAX_RANKS = 16;
struct ClBlocks { const float* blocks; int world; int row0[2][CL_MAX_RANKS + 1]; long base[6][CL_MAX_RANKS]; };
int launch_contrastive_fused(const float* const v[6], float* const dv[6], int n_high, int n_low, int d_high, int d_low, const float w_pair[3],
                             const float w_self[3], float margin, float* loss, void* scratch, size_t scratch_bytes, hipStream_t st,
                             const long* ldv = nullptr, const int* window = nullptr, int pair_mask = 7, const ClBlocks* blk = nullptr);

// fp32 reference mode of the same loss (loss_f32.hip): plain FMA kernels, fixed summation order, no bf16 anywhere; full batch only
size_t contrastive_f32_scratch_bytes(int n_high, int n_low, int d_high, int d_low);
int launch_contrastive_f32(const float* const v[6], float* const dv[6], int n_high, int n_low, int d_high, int d_low, const float w_pair[3],
                           const float w_self[3], float margin, float* loss, void* scratch, size_t scratch_bytes, hipStream_t st);

}  // namespace coot
