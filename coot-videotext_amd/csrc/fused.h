// Fused token-tile chains of the COOT encoder layer for d_model = 384 (every shipped config).
//
// One workgroup (8 waves) owns a tile of BT tokens.  The tile lives in LDS as a bf16 [BT, 384] matrix (800-byte rows:
// conflict-free ds_read_b128 fragment reads on gfx950) and is the MFMA "token" operand of a CHAIN of 384-wide Linear
// layers; the weights are streamed straight from L2 into registers in a fragment-major layout ("P48": the 1 KB a
// wave needs for one 16x32 weight fragment is contiguous, the fragments of the 48 output columns a wave owns follow
// each other).  Between two GEMMs the fp32 accumulators go through a small fp32 LDS staging buffer so that every
// elementwise neighbour (bias, dropout, saved pre-activations, GELU, residual, LayerNorm) runs on 16-byte row-contiguous
// chunks — all global traffic of the chain is coalesced, nothing is re-read from HBM between the Linear layers.
#pragma once
#include "common.h"
#include "rowops.h"

namespace coot {

constexpr int FZ_D = 384;

// element offset of logical W[n][k] (n = output feature, k = reduction index, K = reduction length) in the P48 layout
__host__ __device__ inline long p48_offset(int n, int k, int K) {
  const int g = n / 48, b = (n % 48) / 16, r = n % 16, kb = k / 32, q = (k % 32) / 8, e = k % 8;
  return ((((long)g * (K / 32) + kb) * 3 + b) * 64 + (q * 16 + r)) * 8 + e;
}

// out-proj + residual + LN1 + FF1 + GELU + FF2 + residual + LN2 (+ GenPool FC1 + GELU + FC2) of one encoder layer,
// forward (nntrainer/models/transformer_legacy.py:420-467, :582-605; poolers.py:156-190).  Writes exactly the tensors
// the unfused path saves for the backward pass.
// The layer's self-attention INSIDE post_attn_fwd_kernel (nntrainer/models/transformer_legacy.py:536-563): with fixed-length
// sequences whose length is a multiple of 16 every 16-row fragment of a token tile lies in ONE sequence, so the tile computes
// softmax(Q K^T / sqrt(dh)) V for its own rows from the q | k | v rows infc_qkv_fwd / qkv_fwd wrote (wave h = head h: K_h, V_h of one
// sequence at a time in the wave's own LDS slice, no workgroup barrier; S^T = K Q^T so that a lane owns one query) and the result
// lands in the LDS tile that is the out-projection's operand — no attention launch, no ctx read.  ctx and lse are still
// written (the backward reads them).  Same masks, mask fill, dropout map and summation order per (query, head) as attn_short_fwd.
struct FusedAttn {
  int on = 0;
  const bf16_t* qkv = nullptr;      // [T, 1152]: q | k | v, head h at columns 48 h of each third
  float* lse = nullptr;             // [T, 8]
  int N0 = 0, L0 = 0, N1 = 0, L1 = 0;                // up to two segments of N sequences x L rows (padded layout)
  const long long *lens0 = nullptr, *lens1 = nullptr;  // valid keys per sequence
  DropCfg drop; unsigned long long seed2_delta = 0;  // dropout on the probabilities; the second segment's seed offset
  float scale = 0.f;
};
// in-chain attention: LDS row pitch (bf16 elements) of a wave's K | V slice and the longest sequence it holds, by tile height — the
// workgroup's LDS (162 816 B at 128 rows, 111 616 B at 64) is cut into 8 slices: 2 L RP 2 bytes each
constexpr int FZ_ATTN_RP(int) { return 56; }                       // 112-byte rows: 16-byte aligned, conflict-free 8-byte fragment reads
constexpr int FZ_ATTN_LMAX(int RF) { return RF == 8 ? 80 : 64; }   // 8 x 2 x 80 x 112 = 143 360 B; 8 x 2 x 64 x 112 = 114 688 B (the 64-row kernel declares that much)
constexpr int FZ_ATTN_SMEM(int RF) { return RF >= 4 ? 8 * 2 * FZ_ATTN_LMAX(RF) * FZ_ATTN_RP(RF) * 2 : 0; }
struct PostAttnFwd {
  int T = 0;
  FusedAttn attn;                // attn.on: ctx is an OUTPUT (ctx_w) computed by the kernel
  bf16_t* ctx_w = nullptr;
  const bf16_t* ctx = nullptr;   // [T, 384] attention output (heads concatenated)
  const bf16_t* xres = nullptr;  // [T, 384] sublayer input (residual)
  const bf16_t *wo = nullptr, *w1 = nullptr, *w2 = nullptr, *pw1 = nullptr, *pw2 = nullptr;  // P48 packs
  const float *bo = nullptr, *ln1g = nullptr, *ln1b = nullptr, *b1 = nullptr, *b2 = nullptr, *ln2g = nullptr, *ln2b = nullptr,
              *pb1 = nullptr, *pb2 = nullptr;
  bf16_t *r1 = nullptr, *z1 = nullptr, *h1 = nullptr, *a1 = nullptr, *r2 = nullptr, *z2 = nullptr;
  float* z2_f32 = nullptr; long ldz2_f32 = 0;
  int do_pool = 0;
  bf16_t *hp = nullptr, *ap = nullptr, *s = nullptr;  // [T, 768], [T, 768], [T, 384]
  DropCfg d_postln, d_ff1, d_ff2, d_pool1, d_pool2;
  unsigned long long* tstamps = nullptr;  // profiling aid: s_memtime stamps of block 0 at the phase boundaries
};
int launch_post_attn_fwd(const PostAttnFwd& p, hipStream_t st);
// whether launch_post_attn_fwd(T, do_pool) can take the self-attention of these segments along (tile height of the launch, sequence
// lengths multiples of 16 and <= FZ_ATTN_LMAX, segment boundary on a tile boundary); coot_set_option
// ("fused_attn", 0) switches it off
bool post_attn_can_fuse_attention(int T, bool do_pool, int N0, int L0, int N1, int L1);
void set_fused_attn(int on);
int fused_attn_launches();

// Backward of the same chain, from the GenPool score gradients (or the gradient wrt the layer output) down to the
// gradient wrt the attention output: pooling MLP dX, LN2 backward, FF2 / FF1 dX (GELU'), LN1 backward, out-proj dX.
// Every bias / LayerNorm parameter gradient (column sums over the tokens) leaves the kernel as one partial row per
// tile; launch_pre_attn_bwd() reduces them into the gradient arena with one more launch.  The weight gradients stay
// with the batched TN GEMM (gemm.h), which reads the dY tensors this kernel writes.
struct PreAttnBwd {
  int T = 0;
  int do_pool = 0;
  const bf16_t *ds = nullptr, *dzp = nullptr;  // [T, 384] pooling: score gradients (head h at columns 192 h), direct pooled-feature gradient
  const bf16_t* dz2 = nullptr;                 // [T, 384] gradient wrt the layer output (do_pool == 0)
  const bf16_t *hp = nullptr, *h1 = nullptr, *r2 = nullptr, *r1 = nullptr;  // saved: [T, 768], [T, 384] x 3
  const bf16_t *pw2 = nullptr, *pw1 = nullptr, *w2 = nullptr, *w1 = nullptr, *wo = nullptr;  // P48 packs of the dX orientation
  const float *ln2g = nullptr, *ln1g = nullptr;
  bf16_t *dhp = nullptr, *dr2 = nullptr, *dr2m = nullptr, *dh1 = nullptr, *dr1 = nullptr, *dctx = nullptr;
  float* part = nullptr;                       // [tiles, 3456] column-sum partials
  float *g_pb1 = nullptr, *g_ln2g = nullptr, *g_ln2b = nullptr, *g_b2 = nullptr, *g_b1 = nullptr, *g_ln1g = nullptr, *g_ln1b = nullptr,
        *g_bo = nullptr;                       // gradient arena destinations (+=)
  DropCfg d_pool1, d_ff2, d_ff1, d_postln;
  unsigned long long* tstamps = nullptr;  // profiling aid (block 0 phase stamps)
};
// Input FC + GELU + positional encoding + QKV projection of a local network in one launch: the normalised input xhat
// [T, Din] (bf16) streams through LDS in 64-column slabs, each read exactly once (the 128 x 128 tiles of the LDS-staged
// GEMM read it three times), the 128 x 384 result tile never leaves the CU before the QKV passes.
struct InfcQkvFwd {
  int T = 0, Din = 0;
  const bf16_t* xhat = nullptr;   // [T, Din]
  const bf16_t* win = nullptr;    // P48 [384 x Din] (LayerNorm gain folded)
  const float* bin = nullptr;     // [384] folded bias
  const float* pe = nullptr; int T0 = 0, L1 = 1, L2 = 1;  // pe[pos][384]; pos = row < T0 ? row % L1 : (row - T0) % L2
  const int* pos = nullptr;       // packed rows: pos[row] (written by the input LayerNorm) instead
  const bf16_t* wqkv = nullptr; const float* bqkv = nullptr;
  bf16_t *h0 = nullptr, *z0 = nullptr, *qkv = nullptr;
  unsigned long long* tstamps = nullptr;  // profiling aid: block 0 phase stamps at slots 48.. (tools/fused_stamps.py)
};
int launch_infc_qkv_fwd(const InfcQkvFwd& p, hipStream_t st);

// qkv = z . Wqkv^T + (bq | bk | bv): three full-width passes over one LDS-resident token tile (the LDS-staged GEMM re-reads
// the token slab once per 128-column block: 9 times)
struct QkvFwd { int T = 0; const bf16_t* z = nullptr; const bf16_t* wqkv = nullptr; const float* bias = nullptr; bf16_t* qkv = nullptr; };
int launch_qkv_fwd(const QkvFwd& p, hipStream_t st);

// dz = dqkv . Wqkv + dr1 [* GELU'(h0), column sums -> colsum (input-FC bias gradient)]: K = 1152 as three 384-wide token tiles
struct QkvBwd {
  int T = 0;
  const bf16_t* dqkv = nullptr;  // [T, 1152]
  const bf16_t* wqkv = nullptr;  // P48, logical [384][1152]
  const bf16_t* res = nullptr;   // [T, 384] dr1
  const bf16_t* aux = nullptr;   // [T, 384] h0 (pre-GELU of the input FC) or null
  bf16_t* dz = nullptr;          // [T, 384]
  float* colsum = nullptr;       // [384] column sums of dz (only with aux): += , or = when colsum_overwrite
  int colsum_overwrite = 0;
  float* part = nullptr;         // [tiles, 384] workspace (only with aux)
};
int launch_qkv_bwd(const QkvBwd& p, hipStream_t st);

// ---- global (video / paragraph level) network, the whole forward in ONE launch --------------------------------------------
// TransformerLegacy.forward of a context network (nntrainer/models/transformer_legacy.py:200-288 with use_input_fc off,
// use_context on, pooler avg_special; d_model 384, 8 heads, one encoder layer, one context layer — every shipped config):
// LayerNorm + positional encoding, self-attention encoder layer, context block (one query per sequence, :251-267),
// avg_special pooling (poolers.py:232-241).  A workgroup owns G = 32 / Cmax whole sequences (<= 32 token rows): their tokens
// stay in LDS from the input LayerNorm to the pooled output, the twelve 384-wide weight matrices stream from L2, attention
// runs on the LDS tiles in fp32.  Was 10 dependent launches of 5-30 us on 64-256 rows with the chip idle.
// Writes exactly the tensors the per-op path saves for the backward pass (same buffers, same layout).
struct GlobLayerFwd {
  const bf16_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;  // P48 packs ([1152 x 384], 3 x [384 x 384])
  const float *bqkv = nullptr, *bo = nullptr, *ln1g = nullptr, *ln1b = nullptr, *b1 = nullptr, *b2 = nullptr, *ln2g = nullptr, *ln2b = nullptr;
  bf16_t* q = nullptr; long ldq = 0; bf16_t* k = nullptr; long ldk = 0; bf16_t* v = nullptr; long ldv = 0;  // saved projections
  bf16_t *ctx = nullptr, *r1 = nullptr, *z1 = nullptr, *h1 = nullptr, *a1 = nullptr, *r2 = nullptr, *z2 = nullptr; float* lse = nullptr;
  DropCfg d_attn, d_postln, d_ff1, d_ff2;
};
struct GlobFwd {
  int B = 0, Cmax = 0;            // sequences (videos), padded items (clips) per sequence; Cmax <= 32
  const float* x = nullptr;       // [B, Cmax, 384] fp32, zero padded
  const long long* lens = nullptr;  // [B] valid items
  const float* hidden = nullptr;  // [B, 384] context vectors of the local network
  const float* pe = nullptr;      // [>= Cmax, 384]
  const float *n_gain = nullptr, *n_bias = nullptr;
  GlobLayerFwd self, ctx;
  bf16_t *z0 = nullptr, *cq_in = nullptr;  // saved: LN(x) + pe [B Cmax, 384], bf16(hidden) [B, 384]
  float* pooled = nullptr;        // [B, 768] = avg_special | context block output
  float* per_token = nullptr;     // optional [B Cmax, 384] fp32 copy of the encoder output
  int train = 0;
  int tiles = 0, warm_per_xcd = 0, xcd_first = 0, xcd_count = 8, tiles_per_xcd = 0;  // set by launch_glob_fwd (glob_xcd_set)
  unsigned long long* tstamps = nullptr;  // profiling aid: block 0 phase stamps (tools/glob_stamps.py)
};
int launch_glob_fwd(const GlobFwd& p, hipStream_t st);
bool glob_fwd_supported(int Cmax);

// ---- the same network, backward, in ONE launch (+ the batched weight-gradient GEMM the caller records) ------------------------
// From the gradient of the pooled output [B, 768] down to the gradient wrt the packed input [B, Cmax, 384] and the context
// vectors [B, 384]: context-block chain backward (LN2, FF2^T, GELU', FF1^T, LN1, out-proj^T), one-query attention backward,
// q / k / v projection dX, avg_special backward, encoder-layer chain backward, self-attention backward, QKV dX, input LayerNorm
// backward — on the same 32-row tiles as glob_fwd_kernel.  Writes every dY tensor the weight-gradient GEMMs read (the layout
// of the per-op path's scratch buffers) and adds the bias / LayerNorm parameter gradients (column sums) into the gradient arena.
struct GlobLayerBwd {
  const bf16_t *wqkv_kn = nullptr, *wo_kn = nullptr, *w1_kn = nullptr, *w2_kn = nullptr;  // P48 packs, dX orientation
  const float *ln1g = nullptr, *ln2g = nullptr;
  const bf16_t* q = nullptr; long ldq = 0; const bf16_t* k = nullptr; long ldk = 0; const bf16_t* v = nullptr; long ldv = 0;  // saved
  const bf16_t *r1 = nullptr, *h1 = nullptr, *r2 = nullptr; const float* lse = nullptr;                                        // saved
  bf16_t *dr2 = nullptr, *dr2m = nullptr, *dh1 = nullptr, *dr1 = nullptr;  // written: operands of the weight-gradient GEMMs
  bf16_t* dq = nullptr; long lddq = 0; bf16_t* dk = nullptr; long lddk = 0; bf16_t* dv = nullptr; long lddv = 0;
  float *g_ln2g = nullptr, *g_ln2b = nullptr, *g_ln1g = nullptr, *g_ln1b = nullptr;  // += (the bias gradients b2 / b1 / bo are column sums
                                                                                   // of dr2m / dh1 / dr1: the weight-gradient GEMM takes them)
  DropCfg d_attn, d_postln, d_ff1, d_ff2;
};
struct GlobBwd {
  int B = 0, Cmax = 0;
  const float* x = nullptr;         // [B, Cmax, 384] the forward's input
  const long long* lens = nullptr;
  const float* n_gain = nullptr;
  const float* dpooled = nullptr;   // [B, 768]
  GlobLayerBwd self, ctx;
  float* dx = nullptr;              // [B, Cmax, 384] fp32 (written)
  float* dhidden = nullptr;         // [B, 384] fp32 (written)
  float *g_n_gain = nullptr, *g_n_bias = nullptr;  // +=
  int tiles = 0, warm_per_xcd = 0, xcd_first = 0, xcd_count = 8, tiles_per_xcd = 0;  // set by launch_glob_bwd (glob_xcd_set)
  unsigned long long* tstamps = nullptr;  // profiling aid: block 0 phase stamps at slots 16.. (tools/glob_stamps.py)
};
int launch_glob_bwd(const GlobBwd& p, hipStream_t st);
// XCDs the single-launch global passes of the calling thread place their workgroups on (first, count; default all 8).  A pass
// streams 3.5 MB of weights through each of its few workgroups; the video and the text network run at the same time, and with a
// tile of EACH on every XCD the two weight sets (7 MB) compete for every 4 MB L2.  The train step gives each side half of the
// XCDs (block b runs on XCD b % 8: an affinity used for speed only, any placement is correct).
void glob_xcd_set(int first, int count);

constexpr int FZ_BWD_NCS = 9 * FZ_D;  // floats per tile in PreAttnBwd::part
bool half_tiles(int T, int kernel);  // the chain kernel `kernel` (0 infc_qkv_fwd, 1 post_attn_fwd, 2 pre_attn_bwd, 3 qkv_bwd) runs on 64-row tiles for this many tokens
int launch_pre_attn_bwd(const PreAttnBwd& p, hipStream_t st);

}  // namespace coot
