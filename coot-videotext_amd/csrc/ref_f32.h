// fp32 reference mode of a network (ref_f32.hip): the op sequence of TransformerLegacy.forward in eval mode and its backward, all fp32.
#pragma once
#include <vector>

#include "common.h"

namespace coot {

struct RefLayerP { long wqkv, bq, bk, bv, wo, bo, ln1g, ln1b, w1, b1, w2, b2, ln2g, ln2b; };  // float offsets into the parameter arena
struct RefNetDesc {
  int Din = 0, D = 0, H = 0, F = 0, num_layers = 1, use_input_fc = 0, use_context = 0, ctx_num_layers = 1, pooler = 0, pool_hidden = 0, pool_heads = 1;
  int Nmax = 0;  // sequences of the call (sizes the per-sequence buffers of the context block)
  long n_gain = -1, n_bias = -1, in_w = -1, in_b = -1, pw1 = -1, pb1 = -1, pw2 = -1, pb2 = -1;
  std::vector<RefLayerP> layers, ctx;
};
struct RefSegs { int n = 1; int N[2] = {0, 0}; int L[2] = {0, 0}; const long long* lens[2] = {nullptr, nullptr}; };

// bytes of the `saved` buffer (every intermediate of the forward: the backward reads them there) and of the backward's `scratch`
size_t ref_f32_workspace_bytes(const RefNetDesc& d, long T);
size_t ref_f32_scratch_bytes(const RefNetDesc& d, long T);
// feats / feats2: the (up to two) padded fp32 segments [N, L, Din]; pooled [Ntot, out_dim]; per_token [N0 L0, D] or null
int ref_f32_forward(const RefNetDesc& d, const float* P, const float* pe, const float* feats, const float* feats2, const RefSegs& sg,
                    const float* hidden, float* pooled, float* per_token, void* ws, size_t ws_bytes, hipStream_t st);
// backward of the forward that filled `ws`: ACCUMULATES the parameter gradients into G (same offsets as P), writes dhidden [N, D]
// (context networks) and dfeats [T, Din] (networks without input FC), both optional
int ref_f32_backward(const RefNetDesc& d, const float* P, float* G, const float* feats, const float* feats2, const RefSegs& sg,
                     const float* hidden, const float* dpooled, float* dhidden, float* dfeats, void* ws, size_t ws_bytes, void* scratch,
                     size_t scratch_bytes, hipStream_t st);

}  // namespace coot
