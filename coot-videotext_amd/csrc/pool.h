// Pooling / packing kernels: GenPool column softmax ("atn"), TemporalAvgPool ("avg_special"),
// clip->video packing, small casts.
#pragma once
#include "common.h"
#include "rowops.h"

namespace coot {

struct PoolArgs {
  const bf16_t* s = nullptr; long lds = 0;   // [N*L, D] pre-softmax scores (dropout2 already applied)
  const bf16_t* z = nullptr; long ldz = 0;   // [N*L, D] features being pooled
  const long long* lens = nullptr;           // [N]
  const int* cu = nullptr;                   // packed rows: sequence n is rows [cu[n], cu[n] + lens[n]) of s / z / ds / dz (no padding rows)
  int N = 0, L = 0, D = 0;
  float* pooled = nullptr; long ldp = 0;     // [N, D] fp32 (may be a column slice of a wider matrix)
  float* pooled_copy = nullptr;              // optional dense [N, D] copy kept for the backward
  float* smax = nullptr; float* ssum = nullptr;  // [N, D] saved softmax statistics
  DropCfg drop_w;                            // dropout3 on the softmax weights
  // backward
  const float* dpooled = nullptr; long lddp = 0;
  bf16_t* ds = nullptr; long ldds = 0;       // grad wrt the pre-dropout2 scores (bf16)
  bf16_t* dz = nullptr; long lddz = 0;       // partial grad wrt z (direct path)
  float* ds_colsum = nullptr;                // [D] (bias grad of the 2nd pooling FC)
  float* part_ws = nullptr;                  // internal: [N][D] partials
  DropCfg drop_s; long drop_s_ld = 0;        // dropout2 mask (same indexing as the GEMM epilogue)
  long drop_s_row0 = 0;                      // global row index of this segment's first token
  // Hand-over from the ITEM segment (clips / sentences: item n = video b's c-th item, n = sum_{i < b} counts[i] + c) to the global network's
  // padded [B, Cmax, D] layout by the pooling kernel itself (= launch_pack_fwd, which was a 7-us launch plus a launch gap on the latency
  // chain between the local and the global forward: 1.191 -> 1.184 ms per step): pooled rows also go to pk_out [B, Cmax, D] (padding rows
  // zero), pk_mask [B, Cmax] (1 = padding), pk_lens [B].  (The backward's counterpart — the global network's gradients joining dpooled inside
  // pool_bwd_kernel — was built and measured 1 % slower: the prefix search and the dependent loads at the head of every workgroup cost what
  // the launch saved; profiles/r06_ab_handover.txt.)
  const long long* pk_counts = nullptr; int pk_B = 0, pk_Cmax = 0;
  float* pk_out = nullptr; unsigned char* pk_mask = nullptr; long long* pk_lens = nullptr;
};
int launch_pool_fwd(const PoolArgs& p, hipStream_t stream);
int launch_pool_bwd(const PoolArgs& p, hipStream_t stream);
// the same for up to two segments of sequences (e.g. 64 videos and 256 clips through the same network) in ONE launch:
// grid = all sequences; the backward's bias-gradient partials of both segments take one reduction (same D, same ds_colsum)
int launch_pool_fwd2(const PoolArgs* segs, int nseg, hipStream_t stream);
int launch_pool_bwd2(const PoolArgs* segs, int nseg, hipStream_t stream);

// TemporalAvgPool (poolers.py:232-241): sum over ALL L rows / len
int launch_avgpool_fwd(const bf16_t* z, long ldz, const long long* lens, int N, int L, int D, float* out, long ldo, hipStream_t st);
// dz[n,l,:] = (dz_add ? dz_add : 0) + dpooled[n,:]/len[n]
int launch_avgpool_bwd(const float* dpooled, long lddp, const long long* lens, int N, int L, int D, bf16_t* dz, long lddz, hipStream_t st);

// pack loop of encode_visual (coot/model_retrieval.py:121-136)
int launch_pack_fwd(const float* emb, const long long* counts, int B, int Cmax, int D, float* out, unsigned char* mask,
                    long long* lens, hipStream_t st);
// d_emb[ptr+c,:] += d_out[b,c,:]
int launch_pack_bwd(const float* dout, const long long* counts, int B, int Cmax, int D, float* demb, hipStream_t st);
int launch_pack_bwd_join(const float* dout1, const float* dout2, const float* dctx, const long long* counts, int B, int Cmax, int D,
                         float* demb_items, float* demb_ctx, hipStream_t st);

int launch_cast_bf16_f32(const bf16_t* src, long lds, int R, int C, float* dst, long ldd, hipStream_t st);
int launch_cast_f32_bf16(const float* src, long lds, int R, int C, bf16_t* dst, long ldd, hipStream_t st);
// out_f32[r][c] = a_bf16[r][c] + b_bf16[r][c]  (used to merge two grad streams into an fp32 output)
int launch_add_bf16_to_f32(const bf16_t* a, long lda, const bf16_t* b, long ldb, int R, int C, float* dst, long ldd,
                           int accumulate, hipStream_t st);

}  // namespace coot
