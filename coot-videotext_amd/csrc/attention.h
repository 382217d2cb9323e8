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
  // short self-attention kernels only: an optional SECOND segment of sequences behind the first one in the same token
  // matrices (rows Nseq * Lk ..): Nseq2 sequences of length L2 (e.g. 64 videos x 80 frames, then 256 clips x 80 frames) —
  // one launch for both (launch_attn_fwd/bwd fall back to one launch per segment when a segment is too long)
  int Nseq2 = 0, L2 = 0; const long long* lens2 = nullptr; unsigned long long seed2_delta = 0;
  const int* cu = nullptr;                  // short kernels, packed rows: sequence n (both segments, n < Nseq + Nseq2) starts at row
                                            // cu[n] and has lens[n] rows, all of them valid (no padding rows exist)
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
bool attn_short_path(int L);  // true if self-attention over sequences of length L takes the short kernels

}  // namespace coot
