// Multi-head attention over short padded sequences (L <= a few hundred), gfx950 MFMA.
// Semantics: MultiHeadAttention of nntrainer/models/transformer_legacy.py:492-579 —
// all Lq query rows are computed, keys >= len[n] are filled with -32752 before the softmax.
#pragma once
#include "common.h"
#include "rowops.h"

namespace coot {

struct AttnArgs {
  const bf16_t* q = nullptr; long ldq = 0;  // [Nseq*Lq, >=H*dh], head h at column h*dh
  const bf16_t* k = nullptr; long ldk = 0;  // [Nseq*Lk, ...]
  const bf16_t* v = nullptr; long ldv = 0;
  bf16_t* o = nullptr; long ldo = 0;        // fwd: output; bwd: saved output
  float* lse = nullptr;                     // [Nseq*Lq, H]
  const long long* lens = nullptr;          // [Nseq] valid keys
  int Nseq = 0, Lq = 0, Lk = 0, H = 0, dh = 0;
  float scale = 1.0f;
  int xcd_order = 1;                        // short kernels: XCD-aware (sequence, head) order (set by the launcher)
  DropCfg drop;                             // dropout on the attention probabilities
  // backward only
  const bf16_t* dout = nullptr; long lddo = 0;
  float* delta = nullptr;                   // [Nseq*Lq, H] scratch
  bf16_t* dq = nullptr; long lddq = 0;
  bf16_t* dk = nullptr; long lddk = 0;
  bf16_t* dv = nullptr; long lddv = 0;
};

int launch_attn_fwd(const AttnArgs& a, hipStream_t stream);
int launch_attn_bwd(const AttnArgs& a, hipStream_t stream);
// 1 (default): self-attention with L <= 128 uses the one-workgroup-per-(sequence, head) kernels; 0: chunked kernels (A/B switch)
void set_attn_short(int on);

}  // namespace coot
