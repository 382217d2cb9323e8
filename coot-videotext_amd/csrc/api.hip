// C-ABI of libcoot_hip.so and the native orchestration of the COOT retrieval hot path:
// which kernel runs when, on which buffers.  See include/coot_hip.h for the contract.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/coot_hip.h"
#include "attention.h"
#include "common.h"
#include "fused.h"
#include "gemm.h"
#include "loss.h"
#include "pool.h"
#include "ref_f32.h"
#include "rowops.h"

namespace coot {

static thread_local char g_err[512] = "";
// stream that takes the global network's batched weight-gradient launch (null: the pass's own stream), set around the global
// coot_net_bwd of a step with an early update (api_step.hip: side_backward); ordering event of the hand-over
static thread_local hipStream_t g_glob_flush_stream = nullptr;
static thread_local hipEvent_t g_glob_flush_ev = nullptr;
static int g_glob_flush_aux = 1;  // coot_set_option("glob_flush_aux", 0/1)
static inline int coot_option_glob_flush_aux() { return g_glob_flush_aux; }
extern "C" void coot_internal_set_glob_flush_stream(void* s) { g_glob_flush_stream = (hipStream_t)s; }
// Hand-over request of a step around its LOCAL network's forward (api_step.hip: side_forward): the pooling kernel writes the item
// embeddings into the global network's padded [B, Cmax, D] layout itself (pool.h: pk_*).  One-shot: a call that can serve the request
// takes it (`taken`); the caller launches the stand-alone pack kernel otherwise.
struct PoolHandover { const long long* counts = nullptr; int B = 0, Cmax = 0; float* out = nullptr; unsigned char* mask = nullptr; long long* lens = nullptr;
                      bool fwd = false, taken = false; };
static thread_local PoolHandover g_pool_handover;
static int g_pool_handover_on = 1;  // coot_set_option("pool_handover", 0/1)
extern "C" void coot_internal_set_pool_pack(const long long* counts, int B, int Cmax, float* out, unsigned char* mask, long long* lens) {
  g_pool_handover = PoolHandover{};
  if (counts && g_pool_handover_on) { g_pool_handover.counts = counts; g_pool_handover.B = B; g_pool_handover.Cmax = Cmax; g_pool_handover.out = out;
                                      g_pool_handover.mask = mask; g_pool_handover.lens = lens; g_pool_handover.fwd = true; }
}
extern "C" int coot_internal_pool_handover_taken(void) { const int t = g_pool_handover.taken ? 1 : 0; g_pool_handover = PoolHandover{}; return t; }
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(e));
  return -1;
}

// ---- bump allocator over caller-provided memory (or size planning when base == nullptr) -------
struct Arena {
  char* base; size_t cap; size_t off = 0; bool overflow = false;
  Arena(void* b, size_t c) : base((char*)b), cap(c) {}
  template <typename T> T* get(size_t n) {
    off = (off + 255) & ~(size_t)255;
    size_t bytes = n * sizeof(T);
    char* p = base ? base + off : nullptr;
    off += bytes;
    if (base && off > cap) overflow = true;
    return (T*)p;
  }
};

// ---- parameter layout ---------------------------------------------------------------------------
struct PEntry { std::string name; int64_t off; int64_t shape[4]; int ndim; };

struct LayerP {  // offsets (floats) into the flat arena
  int64_t wqkv, bq, bk, bv, wo, bo, ln1g, ln1b, w1, b1, w2, b2, ln2g, ln2b;
};
struct NetLayout {
  int64_t total = 0;
  int64_t n_gain = -1, n_bias = -1, in_w = -1, in_b = -1;
  std::vector<LayerP> layers, ctx;
  int64_t pw1 = -1, pb1 = -1, pw2 = -1, pb2 = -1;
  std::vector<PEntry> entries;
};

static int norm_cfg(const coot_net_config* c, coot_net_config* o) {
  *o = *c;
  if (o->ff_dim == 0) o->ff_dim = o->hidden_dim;
  if (o->pool_hidden == 0) o->pool_hidden = o->hidden_dim;
  if (o->ctx_num_layers <= 0) o->ctx_num_layers = 1;
  COOT_REQUIRE(o->hidden_dim % o->num_heads == 0, "hidden_dim %d not divisible by %d heads", o->hidden_dim, o->num_heads);
  const int dh = o->hidden_dim / o->num_heads;
  COOT_REQUIRE(dh % 16 == 0 && dh <= 64, "d_head = %d unsupported (16, 32, 48 or 64)", dh);
  COOT_REQUIRE(o->hidden_dim % 8 == 0 && o->ff_dim % 8 == 0 && o->input_dim % 8 == 0, "dims must be multiples of 8");
  COOT_REQUIRE(o->use_input_fc || o->input_dim == o->hidden_dim, "without input_fc, input_dim must equal hidden_dim");
  if (o->pooler == 0) {
    COOT_REQUIRE(o->pool_heads > 0 && o->pool_hidden % o->pool_heads == 0 && o->hidden_dim % o->pool_heads == 0, "bad pooler heads");
    COOT_REQUIRE((o->pool_hidden / o->pool_heads) % 8 == 0 && (o->hidden_dim / o->pool_heads) % 8 == 0, "pooler head dims must be multiples of 8");
  }
  COOT_REQUIRE(o->num_layers >= 1, "num_layers must be >= 1");
  // the 16-bit operand format is a property of the BUILD (common.h: libcoot_hip.so = bfloat16, libcoot_hip_f16.so = IEEE half): a
  // configuration that names the other one is refused, never silently computed in this build's format
  COOT_REQUIRE(o->dtype == COOT_DTYPE_NATIVE || o->dtype == COOT_DTYPE_F32,
               "dtype %d: this build of the library computes on %s operands (dtype %d) or in the fp32 reference mode (%d); the other 16-bit "
               "format is the other build (libcoot_hip.so / libcoot_hip_f16.so)", o->dtype, COOT_OPERAND_IS_F16 ? "IEEE half" : "bfloat16",
               COOT_DTYPE_NATIVE, COOT_DTYPE_F32);
  return 0;
}

static void build_layout(const coot_net_config& c, NetLayout& L) {
  int64_t off = 0;
  auto add = [&](const std::string& name, std::initializer_list<int64_t> shp) {
    PEntry e; e.name = name; e.off = off; e.ndim = (int)shp.size();
    int64_t n = 1; int i = 0;
    for (auto s : shp) { e.shape[i++] = s; n *= s; }
    for (; i < 4; ++i) e.shape[i] = 1;
    L.entries.push_back(e);
    int64_t o = off; off += n; return o;
  };
  const int64_t D = c.hidden_dim, F = c.ff_dim, Din = c.input_dim;
  L.n_gain = add("norm_input.gain", {Din});
  L.n_bias = add("norm_input.bias", {Din});
  if (c.use_input_fc) {
    L.in_w = add("input_fc.mlp.0.weight", {D, Din});
    L.in_b = add("input_fc.mlp.0.bias", {D});
  }
  auto layer = [&](const std::string& pre) {
    LayerP p;
    const std::string a = pre + "self_attention_layer.", f = pre + "pointwise_feedforward_layer.";
    // q,k,v weights are contiguous ([3D, D]) and so are their biases ([3D]) for the fused QKV GEMM
    p.wqkv = add(a + "sublayer.query_projection.weight", {D, D});
    add(a + "sublayer.key_projection.weight", {D, D});
    add(a + "sublayer.value_projection.weight", {D, D});
    p.bq = add(a + "sublayer.query_projection.bias", {D});
    p.bk = add(a + "sublayer.key_projection.bias", {D});
    p.bv = add(a + "sublayer.value_projection.bias", {D});
    p.wo = add(a + "sublayer.final_projection.weight", {D, D});
    p.bo = add(a + "sublayer.final_projection.bias", {D});
    p.ln1g = add(a + "layer_normalization.gain", {D});
    p.ln1b = add(a + "layer_normalization.bias", {D});
    p.w1 = add(f + "sublayer.feed_forward.0.weight", {F, D});
    p.b1 = add(f + "sublayer.feed_forward.0.bias", {F});
    p.w2 = add(f + "sublayer.feed_forward.3.weight", {D, F});
    p.b2 = add(f + "sublayer.feed_forward.3.bias", {D});
    p.ln2g = add(f + "layer_normalization.gain", {D});
    p.ln2b = add(f + "layer_normalization.bias", {D});
    return p;
  };
  for (int i = 0; i < c.num_layers; ++i) L.layers.push_back(layer("tf.encoder_layers." + std::to_string(i) + "."));
  if (c.use_context)
    for (int i = 0; i < c.ctx_num_layers; ++i) L.ctx.push_back(layer("tf_context.encoder_layers." + std::to_string(i) + "."));
  if (c.pooler == 0) {
    const int64_t H = c.pool_heads, dhp = c.pool_hidden / H, dop = D / H;
    L.pw1 = add("pooler.pools.0.genpool_w1_head", {H, D, dhp});
    L.pb1 = add("pooler.pools.0.genpool_b1_head", {H, dhp});
    L.pw2 = add("pooler.pools.0.genpool_w2_head", {H, dhp, dop});
    L.pb2 = add("pooler.pools.0.genpool_b2_head", {H, dop});
  }
  L.total = off;
}

// the fp32 reference mode's view of a network (ref_f32.h)
static RefNetDesc ref_desc(const coot_net_config& c, const NetLayout& L, int Nmax) {
  RefNetDesc d;
  d.Din = c.input_dim; d.D = c.hidden_dim; d.H = c.num_heads; d.F = c.ff_dim; d.num_layers = c.num_layers; d.use_input_fc = c.use_input_fc;
  d.use_context = c.use_context; d.ctx_num_layers = c.ctx_num_layers; d.pooler = c.pooler; d.pool_hidden = c.pool_hidden; d.pool_heads = c.pool_heads;
  d.Nmax = Nmax; d.n_gain = L.n_gain; d.n_bias = L.n_bias; d.in_w = L.in_w; d.in_b = L.in_b; d.pw1 = L.pw1; d.pb1 = L.pb1; d.pw2 = L.pw2; d.pb2 = L.pb2;
  auto cv = [](const LayerP& p) { return RefLayerP{p.wqkv, p.bq, p.bk, p.bv, p.wo, p.bo, p.ln1g, p.ln1b, p.w1, p.b1, p.w2, p.b2, p.ln2g, p.ln2b}; };
  for (const LayerP& p : L.layers) d.layers.push_back(cv(p));
  for (const LayerP& p : L.ctx) d.ctx.push_back(cv(p));
  return d;
}

// ---- bf16 weight pack layout ----------------------------------------------------------------------
struct LayerW {
  bf16_t *wqkv_nk, *wqkv_kn, *wo_nk, *wo_kn, *w1_nk, *w1_kn, *w2_nk, *w2_kn;
  bf16_t *f_wo = nullptr, *f_w1 = nullptr, *f_w2 = nullptr;  // P48 packs of the fused chains (fused.h), d_model = 384 only
  bf16_t *f_wo_kn = nullptr, *f_w1_kn = nullptr, *f_w2_kn = nullptr;  // the same for the dX orientation (backward chain)
  bf16_t *f_wqkv = nullptr, *f_wqkv_kn = nullptr;  // P48: [1152 x 384] forward, [384 x 1152] dX
};
struct WPack {
  bf16_t* in_w = nullptr;     // [D, Din]  = W * gain (LN affine folded)
  bf16_t* f_in_w = nullptr;   // the same in the P48 layout (fused input FC + QKV kernel)
  float* in_bias = nullptr;   // [D]       = b + W . norm_bias
  std::vector<LayerW> layers, ctx;
  bf16_t *pw1_nk = nullptr, *pw1_kn = nullptr, *pw2_nk = nullptr, *pw2_kn = nullptr;
  bf16_t *f_pw1 = nullptr, *f_pw2 = nullptr;  // P48: [768 x 384] (head-major rows), 2 x [192 x 384]
  bf16_t *f_pw1_kn = nullptr, *f_pw2_kn = nullptr;  // P48, dX orientation: 2 x [N = 384 (d), K = 384 (e)], 2 x [N = 384 (e), K = 192 (o)]
};
// the fused token-tile chains are specialised for the shipped model width
static bool fused_layer_ok(const coot_net_config& c) { return c.hidden_dim == FZ_D && c.ff_dim == FZ_D; }
static bool fused_pool_ok(const coot_net_config& c) {
  return fused_layer_ok(c) && c.pooler == 0 && c.pool_hidden == 2 * FZ_D && c.pool_heads == 2;
}
static void layout_wpack(const coot_net_config& c, Arena& A, WPack& W) {
  const size_t D = c.hidden_dim, F = c.ff_dim, Din = c.input_dim;
  if (c.use_input_fc) {
    W.in_w = A.get<bf16_t>(D * Din); W.in_bias = A.get<float>(D);
    if (fused_layer_ok(c) && Din % 64 == 0 && Din >= 128) W.f_in_w = A.get<bf16_t>(D * Din);  // infc_qkv_fwd's K loop: two 64-column slabs deep
  }
  auto lay = [&]() {
    LayerW w;
    w.wqkv_nk = A.get<bf16_t>(3 * D * D); w.wqkv_kn = A.get<bf16_t>(3 * D * D);
    w.wo_nk = A.get<bf16_t>(D * D); w.wo_kn = A.get<bf16_t>(D * D);
    w.w1_nk = A.get<bf16_t>(F * D); w.w1_kn = A.get<bf16_t>(F * D);
    w.w2_nk = A.get<bf16_t>(F * D); w.w2_kn = A.get<bf16_t>(F * D);
    if (fused_layer_ok(c)) {
      w.f_wo = A.get<bf16_t>(D * D); w.f_w1 = A.get<bf16_t>(F * D); w.f_w2 = A.get<bf16_t>(F * D);
      w.f_wo_kn = A.get<bf16_t>(D * D); w.f_w1_kn = A.get<bf16_t>(F * D); w.f_w2_kn = A.get<bf16_t>(F * D);
      w.f_wqkv = A.get<bf16_t>(3 * D * D); w.f_wqkv_kn = A.get<bf16_t>(3 * D * D);
    }
    return w;
  };
  for (int i = 0; i < c.num_layers; ++i) W.layers.push_back(lay());
  if (c.use_context) for (int i = 0; i < c.ctx_num_layers; ++i) W.ctx.push_back(lay());
  if (c.pooler == 0) {
    const size_t PH = c.pool_hidden;
    W.pw1_nk = A.get<bf16_t>(PH * D); W.pw1_kn = A.get<bf16_t>(PH * D);
    W.pw2_nk = A.get<bf16_t>(PH * (D / c.pool_heads)); W.pw2_kn = A.get<bf16_t>(PH * (D / c.pool_heads));
    if (fused_pool_ok(c)) {
      W.f_pw1 = A.get<bf16_t>(PH * D); W.f_pw2 = A.get<bf16_t>(PH * (D / c.pool_heads));
      W.f_pw1_kn = A.get<bf16_t>(PH * D); W.f_pw2_kn = A.get<bf16_t>(PH * (D / c.pool_heads));
    }
  }
}

// ---- saved-for-backward + scratch layouts ----------------------------------------------------------
struct LayerS {
  bf16_t *qkv, *ctx, *r1, *z1, *h1, *a1, *r2, *z2; float* lse;
};
struct CtxS {
  bf16_t *q, *kv, *cctx, *r1, *z1, *h1, *a1, *r2, *z2; float* lse;
};
struct Saved {
  bf16_t *xhat = nullptr, *h0 = nullptr, *z0 = nullptr;
  int* pos = nullptr;  // packed rows: position of every row within its sequence
  std::vector<LayerS> layers; std::vector<CtxS> ctx;
  bf16_t* cq_in = nullptr;
  bf16_t *hp = nullptr, *ap = nullptr, *s = nullptr; float *smax = nullptr, *ssum = nullptr, *pooled = nullptr;
};
static void layout_saved(const coot_net_config& c, int N_in, long Ttok, Arena& A, Saved& S) {
  // Every activation is allocated in whole 128-row tiles: the fused token-tile chains (fused.hip) read and write whole
  // tiles without row bounds checks; rows past T / N hold don't-care values that no kernel consumes.
  const size_t T = ((size_t)Ttok + 127) & ~(size_t)127, D = c.hidden_dim, F = c.ff_dim, H = c.num_heads;
  const size_t N = ((size_t)N_in + 127) & ~(size_t)127;
  if (c.use_input_fc) { S.xhat = A.get<bf16_t>(T * c.input_dim); S.h0 = A.get<bf16_t>(T * D); S.pos = A.get<int>(T); }
  S.z0 = A.get<bf16_t>(T * D);
  for (int i = 0; i < c.num_layers; ++i) {
    LayerS l;
    l.qkv = A.get<bf16_t>(T * 3 * D); l.lse = A.get<float>(T * H); l.ctx = A.get<bf16_t>(T * D);
    l.r1 = A.get<bf16_t>(T * D); l.z1 = A.get<bf16_t>(T * D); l.h1 = A.get<bf16_t>(T * F); l.a1 = A.get<bf16_t>(T * F);
    l.r2 = A.get<bf16_t>(T * D); l.z2 = A.get<bf16_t>(T * D);
    S.layers.push_back(l);
  }
  if (c.use_context) {
    S.cq_in = A.get<bf16_t>((size_t)N * D);
    for (int i = 0; i < c.ctx_num_layers; ++i) {
      CtxS l;
      l.q = A.get<bf16_t>((size_t)N * D); l.kv = A.get<bf16_t>(T * 2 * D); l.lse = A.get<float>((size_t)N * H);
      l.cctx = A.get<bf16_t>((size_t)N * D); l.r1 = A.get<bf16_t>((size_t)N * D); l.z1 = A.get<bf16_t>((size_t)N * D);
      l.h1 = A.get<bf16_t>((size_t)N * F); l.a1 = A.get<bf16_t>((size_t)N * F); l.r2 = A.get<bf16_t>((size_t)N * D);
      l.z2 = A.get<bf16_t>((size_t)N * D);
      S.ctx.push_back(l);
    }
  }
  if (c.pooler == 0) {
    const size_t PH = c.pool_hidden;
    S.hp = A.get<bf16_t>(T * PH); S.ap = A.get<bf16_t>(T * PH); S.s = A.get<bf16_t>(T * D);
    S.smax = A.get<float>((size_t)N * D); S.ssum = A.get<float>((size_t)N * D); S.pooled = A.get<float>((size_t)N * D);
  }
}

struct Scratch {  // backward temporaries
  bf16_t *dzA, *dzB, *dr2, *dr2m, *dh1, *dz1, *dr1, *dctx, *dqkv, *ds, *dhp, *dzp;
  float *delta, *Mbuf, *cvec, *tn_ws; size_t tn_ws_floats; float* part_ws; size_t part_floats, part_low;
  bf16_t *c_dq, *c_dkv, *c_d1, *c_d2, *c_dh1, *c_dz1, *c_dr1, *c_dctx, *c_dqin; float* c_delta;
};
static void layout_scratch(const coot_net_config& c, int N, long Ttok, Arena& A, Scratch& S) {
  // whole 128-row tiles, like layout_saved (the fused backward chain writes whole tiles)
  const size_t T = ((size_t)Ttok + 127) & ~(size_t)127, D = c.hidden_dim, F = c.ff_dim, H = c.num_heads;
  S.dzA = A.get<bf16_t>(T * D); S.dzB = A.get<bf16_t>(T * D); S.dr2 = A.get<bf16_t>(T * D); S.dr2m = A.get<bf16_t>(T * D);
  S.dh1 = A.get<bf16_t>(T * F); S.dz1 = A.get<bf16_t>(T * D); S.dr1 = A.get<bf16_t>(T * D); S.dctx = A.get<bf16_t>(T * D);
  S.dqkv = A.get<bf16_t>(T * 3 * D); S.delta = A.get<float>(T * H);
  S.ds = nullptr; S.dhp = nullptr; S.dzp = nullptr; S.Mbuf = nullptr; S.cvec = nullptr;
  if (c.pooler == 0) { S.ds = A.get<bf16_t>(T * D); S.dhp = A.get<bf16_t>(T * c.pool_hidden); S.dzp = A.get<bf16_t>(T * D); }
  if (c.use_input_fc) { S.Mbuf = A.get<float>(D * c.input_dim); S.cvec = A.get<float>(D); }
  {
    size_t w = gemm_tn_workspace_floats((int)T, 3 * (int)D, (int)D, 1);
    auto mx = [&](size_t v) { if (v > w) w = v; };
    mx(gemm_tn_workspace_floats((int)T, (int)D, (int)F, 1)); mx(gemm_tn_workspace_floats((int)T, (int)F, (int)D, 1));
    if (c.use_input_fc) mx(gemm_tn_workspace_floats((int)T, (int)D, c.input_dim, 1));
    if (c.pooler == 0) {
      const int Hh = c.pool_heads, dhp = c.pool_hidden / Hh, dop = (int)D / Hh;
      mx(gemm_tn_workspace_floats((int)T, dhp, dop, Hh)); mx(gemm_tn_workspace_floats((int)T, (int)D, dhp, Hh));
    }
    {  // batched weight-gradient launch (tn_batch_flush): every dW of the network in flight, <= 8 splits each
      const size_t smax = T / 512 > 8 ? 8 : (T / 512 > 0 ? T / 512 : 1);
      size_t sum = 0;
      const size_t per_layer = 4 * D * D + 2 * D * F;
      sum += per_layer * ((size_t)c.num_layers + (c.use_context ? (size_t)c.ctx_num_layers : 0));
      if (c.use_input_fc) sum += D * (size_t)c.input_dim;
      if (c.pooler == 0) sum += (size_t)c.pool_hidden * D + (size_t)c.pool_hidden * (D / c.pool_heads);
      mx(smax * sum);
    }
    S.tn_ws_floats = w; S.tn_ws = A.get<float>(w);
  }
  {
    size_t widest = 3 * D; if (F > widest) widest = F; if ((size_t)c.pool_hidden > widest) widest = c.pool_hidden;
    S.part_floats = (T / 64 + 1024) * widest;
    if (S.part_floats < (T / 64) * (size_t)FZ_BWD_NCS) S.part_floats = (T / 64) * (size_t)FZ_BWD_NCS;  // one partial row per (64-row) tile
    S.part_low = S.part_floats;  // immediate users; behind it: the regions of the deferred reductions of one layer (rowops.h):
    S.part_floats += (T / 64 + 1) * ((size_t)FZ_BWD_NCS + D) + (size_t)N * D + 1024;  // fused backward chain, QKV dX, pooling
    S.part_ws = A.get<float>(S.part_floats);
  }
  if (c.use_context) {
    const size_t n = N;
    S.c_dq = A.get<bf16_t>(n * D); S.c_dkv = A.get<bf16_t>(T * 2 * D); S.c_d1 = A.get<bf16_t>(n * D); S.c_d2 = A.get<bf16_t>(n * D);
    S.c_dh1 = A.get<bf16_t>(n * F); S.c_dz1 = A.get<bf16_t>(n * D); S.c_dr1 = A.get<bf16_t>(n * D); S.c_dctx = A.get<bf16_t>(n * D);
    S.c_dqin = A.get<bf16_t>(n * D); S.c_delta = A.get<float>(n * H);
  }
}

static thread_local const unsigned long long* g_seed_dev = nullptr;  // device base seed of the current call
static DropCfg mkdrop(int train, float p, uint64_t seed, unsigned site) {
  DropCfg d;
  if (train && p > 0.f) {
    double t = (double)p * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    if (d.thr == 0) d.thr = 1;
    if (d.thr < 65536u) d.thr = 65536u;  // 16-bit compare (common.h): p quantised to 1/65536, inv_keep from the quantised value
    d.inv_keep = (float)(1.0 / (1.0 - (double)(d.thr >> 16) / 65536.0));
    d.seed = seed; d.site = site; d.seed_ptr = g_seed_dev;
  }
  return d;
}
static void epi_drop(GemmEpi& e, const DropCfg& d, long ld) {
  e.drop_thr = d.thr; e.drop_inv_keep = d.inv_keep; e.drop_seed = d.seed; e.drop_seed_ptr = d.seed_ptr; e.drop_site = d.site; e.drop_ld = ld;
}
enum { SITE_ATTN = 1, SITE_POSTLN = 2, SITE_FF1 = 3, SITE_FF2 = 4, SITE_POOL1 = 5, SITE_POOL2 = 6, SITE_POOL3 = 7 };

#define RUN(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

// One TransformerEncoderLayer forward on `rows` query rows (nntrainer/models/transformer_legacy.py:420-438).
// Self-attention: xkv == xq, Lq == Lk.  Cross-attention (context block): xq = [N, D] (Lq = 1), xkv = tokens.
struct LayerBufs {
  bf16_t *q; long ldq; bf16_t *k; long ldk; bf16_t* v; long ldv;   // projected q/k/v (views of qkv or q + kv)
  bf16_t *ctx, *r1, *z1, *h1, *a1, *r2, *z2; float* lse;
};

// A network call processes up to two SEGMENTS of sequences through the same weights (e.g. the 64 whole
// videos and their 256 clips of one batch, coot/model_retrieval.py:104 and :120): all token-level kernels
// (GEMMs, LayerNorm, column sums) run once over the concatenated token matrix; only the kernels that see the
// sequence structure (attention, pooling, positional encoding) are segment aware.
struct Segs {
  int n = 1; int N[2] = {0, 0}; int L[2] = {0, 0}; const long long* lens[2] = {nullptr, nullptr};
  // packed (varlen) token rows: sequence i of the call (segment 0 first) is rows [cu[i], cu[i] + len) of every token matrix, no
  // padding rows exist; Tp = cu[Ntot] on the host.  Null: the padded layout, row = n L + l per segment.
  const int* cu = nullptr; int Tp = 0;
  int Ntot() const { return N[0] + (n > 1 ? N[1] : 0); }
  int Tpad() const { return N[0] * L[0] + (n > 1 ? N[1] * L[1] : 0); }
  int T() const { return cu ? Tp : Tpad(); }
};

// self-attention over every segment (rows of segment s start at sum_{s'<s} N*L), or one cross-attention
// (context block: one query row per sequence) when `cross`
static int attention_all(AttnArgs a, const Segs& sg, bool cross, bool bwd, hipStream_t st) {
  if (cross) {
    COOT_REQUIRE(sg.n == 1, "context networks take a single segment");
    a.lens = sg.lens[0]; a.Nseq = sg.N[0]; a.Lq = 1; a.Lk = sg.L[0];
    return bwd ? launch_attn_bwd(a, st) : launch_attn_fwd(a, st);
  }
  if (sg.cu) {  // packed rows: the short kernels, every sequence at its own row offset (both segments in one launch)
    COOT_REQUIRE(attn_short_path(sg.L[0]) && (sg.n == 1 || attn_short_path(sg.L[1])), "packed rows need the short attention kernels (L <= 128)");
    a.cu = sg.cu; a.lens = sg.lens[0]; a.Nseq = sg.N[0]; a.Lq = sg.L[0]; a.Lk = sg.L[0];
    if (sg.n == 2) { a.Nseq2 = sg.N[1]; a.L2 = sg.L[1]; a.lens2 = sg.lens[1]; a.seed2_delta = 0x9E3779B97F4A7C15ull; }
    return bwd ? launch_attn_bwd(a, st) : launch_attn_fwd(a, st);
  }
  if (sg.n == 2 && attn_short_path(sg.L[0]) && attn_short_path(sg.L[1])) {  // both segments in ONE launch of the short kernels
    a.lens = sg.lens[0]; a.Nseq = sg.N[0]; a.Lq = sg.L[0]; a.Lk = sg.L[0];
    a.Nseq2 = sg.N[1]; a.L2 = sg.L[1]; a.lens2 = sg.lens[1]; a.seed2_delta = 0x9E3779B97F4A7C15ull;
    return bwd ? launch_attn_bwd(a, st) : launch_attn_fwd(a, st);
  }
  long row = 0;
  const AttnArgs base = a;
  for (int s = 0; s < sg.n; ++s) {
    a = base;
    a.q = base.q + row * base.ldq; a.k = base.k + row * base.ldk; a.v = base.v + row * base.ldv; a.o = base.o + row * base.ldo;
    a.lse = base.lse + row * base.H;
    if (bwd) {
      a.dout = base.dout + row * base.lddo; a.delta = base.delta + row * base.H;
      a.dq = base.dq + row * base.lddq; a.dk = base.dk + row * base.lddk; a.dv = base.dv + row * base.lddv;
    }
    a.lens = sg.lens[s]; a.Nseq = sg.N[s]; a.Lq = sg.L[s]; a.Lk = sg.L[s];
    a.drop.seed = base.drop.seed + (unsigned long long)s * 0x9E3779B97F4A7C15ull;
    RUN(bwd ? launch_attn_bwd(a, st) : launch_attn_fwd(a, st));
    row += (long)sg.N[s] * sg.L[s];
  }
  return 0;
}

// GenPool score MLP fused behind the last encoder layer (fused path only)
struct PoolFuse { const bf16_t *pw1, *pw2; const float *pb1, *pb2; bf16_t *hp, *ap, *s; DropCfg d1, d2; };
struct PoolFuseBwd { const bf16_t *ds, *dzp, *hp, *pw2, *pw1; bf16_t* dhp; float* g_pb1; DropCfg d1; };
static int g_use_fused_infc = 1;  // coot_set_option("fused_infc", 0/1): input FC + QKV in one launch (+1.4 % on the step once its K loop was pipelined two slabs deep)
static int g_use_fused = 1;
static int g_grad_poison = 0;  // coot_set_option("grad_poison", 1) (tests): coot_nets_zero_grads fills the matrices it skips with NaN
// Workgroup cap of the prefetched input LayerNorm.  Measured on the ActivityNet shape (profiles/README.md, round 3): no cap 1.227, 1024: 1.235,
// 512: 1.217, 256: 1.213-1.228, 128: 1.259 ms per step (1.238 without the prefetch).  (The A/B switch "ln_stream_wgs" went in round 6.)
// The cap keeps the streaming launch from evicting the weights the single-launch global passes live on.  A feature stream of several
// hundred MB (BASELINE.json configs[4]: 64 clips per video, where the global networks run their per-op kernels for a millisecond) has the
// time and needs the bandwidth: 512 workgroups deliver 3.4 TB/s, 2 048 4.5, 4 096 5.1 at the same step time: the cap is lifted from 256 MB of input on (round 5, profiles/README.md).
static int ln_stream_cap(size_t input_bytes) { return input_bytes >= ((size_t)256 << 20) ? 4096 : 512; }
constexpr int g_ln_nt = 1;     // ... with streaming (non-temporal) loads / stores: next to the global networks it must not evict their weights from L2
static int g_pack_lazy = 1, g_pack_poison = 0;  // coot_set_option("pack_lazy" / "pack_poison"): lazily packed per-op layouts (below)
static int g_fz_debug = 0;
static unsigned long long* g_fz_tstamps = nullptr;
constexpr int g_fused_fwd_small = 1;  // forward chain on 32-token tiles below fused_min_rows
static int g_fused_min_rows = 1024;  // below this many tokens the per-op kernels win (one or two tiles cannot fill the chip)  // coot_set_option("fused", 0/1): A/B switch between the fused chains and the per-op kernels

static int layer_fwd(const coot_net_config& c, const float* P, const LayerP& lp, const LayerW& lw, const bf16_t* xq, int rows_q,
                     const bf16_t* xkv, int rows_kv, const Segs& sg, const LayerBufs& b,
                     float* z2_f32, long ldz2_f32, float pdrop, int train, uint64_t seed, unsigned site_base, hipStream_t st,
                     const PoolFuse* pool = nullptr, bool qkv_done = false) {
  const int D = c.hidden_dim, F = c.ff_dim, H = c.num_heads, dh = D / H;
  const bool self = (xq == xkv);
  if (self && qkv_done) {
    // q | k | v already written by the fused input-FC kernel
  } else if (self && lw.f_wqkv && g_use_fused && rows_q >= g_fused_min_rows) {
    QkvFwd f; f.T = rows_q; f.z = xq; f.wqkv = lw.f_wqkv; f.bias = P + lp.bq; f.qkv = b.q;
    RUN(launch_qkv_fwd(f, st));
  } else if (self) {
    GemmNT g; g.X = xq; g.ldx = D; g.W = lw.wqkv_nk; g.ldw = D; g.M = rows_q; g.N = 3 * D; g.K = D;
    g.epi.bias = P + lp.bq; g.epi.out = b.q; g.epi.ldc = 3 * D;
    RUN(launch_gemm_nt(g, st));
  } else {
    GemmNT g; g.X = xq; g.ldx = D; g.W = lw.wqkv_nk; g.ldw = D; g.M = rows_q; g.N = D; g.K = D;
    g.epi.bias = P + lp.bq; g.epi.out = b.q; g.epi.ldc = b.ldq;
    RUN(launch_gemm_nt(g, st));
    GemmNT g2; g2.X = xkv; g2.ldx = D; g2.W = lw.wqkv_nk + (size_t)D * D; g2.ldw = D; g2.M = rows_kv; g2.N = 2 * D; g2.K = D;
    g2.epi.bias = P + lp.bk; g2.epi.out = b.k; g2.epi.ldc = b.ldk;
    RUN(launch_gemm_nt(g2, st));
  }
  AttnArgs a; a.q = b.q; a.ldq = b.ldq; a.k = b.k; a.ldk = b.ldk; a.v = b.v; a.ldv = b.ldv; a.o = b.ctx; a.ldo = D; a.lse = b.lse;
  a.H = H; a.dh = dh; a.scale = 1.0f / sqrtf((float)dh);
  a.drop = mkdrop(train, pdrop, seed, site_base + SITE_ATTN);
  const bool post_fused = lw.f_wo && g_use_fused && (rows_q >= g_fused_min_rows || pool || g_fused_fwd_small);
  // the self-attention inside the token-tile chain (fused.h: FusedAttn) when every 16-row fragment of a tile lies in one
  // sequence — padded layout, sequence lengths multiples of 16 (ActivityNet: 80 frames, 64 / 16 words), one q | k | v matrix
  const bool attn_in_chain = post_fused && self && !sg.cu && H == 8 && dh == 48 && b.ldq == 3 * D && b.ldk == 3 * D && b.ldv == 3 * D &&
                             b.k == b.q + D && b.v == b.q + 2 * D && rows_q == sg.Tpad() &&
                             post_attn_can_fuse_attention(rows_q, pool != nullptr, sg.N[0], sg.L[0], sg.n > 1 ? sg.N[1] : 0, sg.n > 1 ? sg.L[1] : 0);
  if (!attn_in_chain) RUN(attention_all(a, sg, !self, false, st));
  if (post_fused) {  // out-proj ... LN2 (and the GenPool score MLP) as ONE launch over token tiles
    PostAttnFwd f; f.T = rows_q; f.ctx = b.ctx; f.xres = xq; f.wo = lw.f_wo; f.w1 = lw.f_w1; f.w2 = lw.f_w2;
    if (attn_in_chain) {
      f.attn.on = 1; f.attn.qkv = b.q; f.attn.lse = b.lse; f.ctx_w = b.ctx; f.attn.drop = a.drop; f.attn.scale = a.scale;
      f.attn.N0 = sg.N[0]; f.attn.L0 = sg.L[0]; f.attn.lens0 = sg.lens[0];
      if (sg.n > 1) { f.attn.N1 = sg.N[1]; f.attn.L1 = sg.L[1]; f.attn.lens1 = sg.lens[1]; f.attn.seed2_delta = 0x9E3779B97F4A7C15ull; }
    }
    f.bo = P + lp.bo; f.ln1g = P + lp.ln1g; f.ln1b = P + lp.ln1b; f.b1 = P + lp.b1; f.b2 = P + lp.b2; f.ln2g = P + lp.ln2g; f.ln2b = P + lp.ln2b;
    f.r1 = b.r1; f.z1 = b.z1; f.h1 = b.h1; f.a1 = b.a1; f.r2 = b.r2; f.z2 = b.z2; f.z2_f32 = z2_f32; f.ldz2_f32 = ldz2_f32;
    f.d_postln = mkdrop(train, pdrop, seed, site_base + SITE_POSTLN); f.d_ff1 = mkdrop(train, pdrop, seed, site_base + SITE_FF1);
    f.d_ff2 = mkdrop(train, pdrop, seed, site_base + SITE_FF2);
    if (pool) {
      f.do_pool = 1; f.pw1 = pool->pw1; f.pw2 = pool->pw2; f.pb1 = pool->pb1; f.pb2 = pool->pb2; f.hp = pool->hp; f.ap = pool->ap; f.s = pool->s;
      f.d_pool1 = pool->d1; f.d_pool2 = pool->d2;
    }
    f.tstamps = g_fz_tstamps;
    return launch_post_attn_fwd(f, st);
  }
  COOT_REQUIRE(!pool, "layer_fwd: fused pooling requested on the unfused path");
  {
    GemmNT g; g.X = b.ctx; g.ldx = D; g.W = lw.wo_nk; g.ldw = D; g.M = rows_q; g.N = D; g.K = D;
    g.epi.bias = P + lp.bo; g.epi.res = xq; g.epi.ldres = D; g.epi.out = b.r1; g.epi.ldc = D;
    RUN(launch_gemm_nt(g, st));
  }
  {
    LnFwd l; l.x = b.r1; l.x_f32 = 0; l.ldx = D; l.R = rows_q; l.D = D; l.gain = P + lp.ln1g; l.bias = P + lp.ln1b; l.y = b.z1; l.ldy = D;
    l.drop = mkdrop(train, pdrop, seed, site_base + SITE_POSTLN);
    RUN(launch_ln_fwd(l, st));
  }
  {
    GemmNT g; g.X = b.z1; g.ldx = D; g.W = lw.w1_nk; g.ldw = D; g.M = rows_q; g.N = F; g.K = D;
    g.epi.bias = P + lp.b1; g.epi.act = 1; g.epi.save_pre = b.h1; g.epi.ldpre = F; g.epi.out = b.a1; g.epi.ldc = F;
    epi_drop(g.epi, mkdrop(train, pdrop, seed, site_base + SITE_FF1), F);
    RUN(launch_gemm_nt(g, st));
  }
  {
    GemmNT g; g.X = b.a1; g.ldx = F; g.W = lw.w2_nk; g.ldw = F; g.M = rows_q; g.N = D; g.K = F;
    g.epi.bias = P + lp.b2; g.epi.res = b.z1; g.epi.ldres = D; g.epi.out = b.r2; g.epi.ldc = D;
    epi_drop(g.epi, mkdrop(train, pdrop, seed, site_base + SITE_FF2), D);
    RUN(launch_gemm_nt(g, st));
  }
  {
    LnFwd l; l.x = b.r2; l.x_f32 = 0; l.ldx = D; l.R = rows_q; l.D = D; l.gain = P + lp.ln2g; l.bias = P + lp.ln2b; l.y = b.z2; l.ldy = D;
    l.y32 = z2_f32; l.ldy32 = ldz2_f32;
    RUN(launch_ln_fwd(l, st));
  }
  return 0;
}

// Backward of layer_fwd.  dz2: grad wrt the layer output (bf16 [rows_q, D]) or fp32 (dz2_f32).
// Outputs: dxq (bf16 [rows_q, D], or fp32 dxq_f32), and for cross-attention dxkv_accum (bf16 [rows_kv, D], in-place +=).
struct LayerBwdBufs { bf16_t *dr2, *dr2m, *dh1, *dz1, *dr1, *dctx, *dq; long lddq; bf16_t* dk; long lddk; bf16_t* dv; long lddv; float* delta; };

static bool qkv_bwd_is_fused(const LayerW& lw, int rows_q, bool dxq_is_f32, const float* part_ws) {
  return lw.f_wqkv_kn && g_use_fused && rows_q >= g_fused_min_rows && !dxq_is_f32 && part_ws;
}

static int layer_bwd(const coot_net_config& c, const float* P, float* G, const LayerP& lp, const LayerW& lw, const bf16_t* xq,
                     int rows_q, const bf16_t* xkv, int rows_kv, const Segs& sg,
                     const LayerBufs& b, const LayerBwdBufs& w, const bf16_t* dz2, const float* dz2_f32, long lddz2_f32,
                     bf16_t* dxq, float* dxq_f32, bf16_t* dxkv_accum, const bf16_t* gelu_aux, float* gelu_colsum, float pdrop,
                     int train, uint64_t seed, unsigned site_base, hipStream_t st, const PoolFuseBwd* pool = nullptr, float* part_ws = nullptr) {
  const int D = c.hidden_dim, F = c.ff_dim, H = c.num_heads, dh = D / H;
  const bool self = (xq == xkv);
  const DropCfg d_ff2 = mkdrop(train, pdrop, seed, site_base + SITE_FF2);
  const bool fused = g_use_fused && lw.f_w2_kn && part_ws && rows_q >= g_fused_min_rows && (pool || dz2);
  COOT_REQUIRE(!pool || fused, "layer_bwd: fused pooling backward requested on the unfused path");
  if (fused) {  // pooling MLP dX + LN2 bwd + FF dX + LN1 bwd + out-proj dX as ONE launch over token tiles (+ one reduction)
    PreAttnBwd f; f.T = rows_q; f.h1 = b.h1; f.r2 = b.r2; f.r1 = b.r1; f.w2 = lw.f_w2_kn; f.w1 = lw.f_w1_kn; f.wo = lw.f_wo_kn;
    f.ln2g = P + lp.ln2g; f.ln1g = P + lp.ln1g; f.dr2 = w.dr2; f.dr2m = w.dr2m; f.dh1 = w.dh1; f.dr1 = w.dr1; f.dctx = w.dctx; f.part = part_ws;
    f.g_ln2g = G + lp.ln2g; f.g_ln2b = G + lp.ln2b; f.g_b2 = G + lp.b2; f.g_b1 = G + lp.b1; f.g_ln1g = G + lp.ln1g; f.g_ln1b = G + lp.ln1b;
    f.g_bo = G + lp.bo;
    f.d_ff2 = d_ff2; f.d_ff1 = mkdrop(train, pdrop, seed, site_base + SITE_FF1); f.d_postln = mkdrop(train, pdrop, seed, site_base + SITE_POSTLN);
    if (pool) {
      f.do_pool = 1; f.ds = pool->ds; f.dzp = pool->dzp; f.hp = pool->hp; f.pw2 = pool->pw2; f.pw1 = pool->pw1; f.dhp = pool->dhp;
      f.g_pb1 = pool->g_pb1; f.d_pool1 = pool->d1;
    } else {
      f.dz2 = dz2;
    }
    f.tstamps = g_fz_tstamps;
    RUN(launch_pre_attn_bwd(f, st));
  } else {
    LnBwd l; l.dy = dz2; l.lddy = D; l.dy32 = dz2_f32; l.lddy32 = lddz2_f32; l.x = b.r2; l.x_f32 = 0; l.ldx = D; l.gain = P + lp.ln2g;
    l.R = rows_q; l.D = D; l.dx = w.dr2; l.lddx = D; l.dgain = G + lp.ln2g; l.dbias = G + lp.ln2b; l.dxcolsum = G + lp.b2;
    if (d_ff2.thr) { l.dxm = w.dr2m; l.lddxm = D; l.dxm_drop = d_ff2; l.dxm_drop_ld = D; }
    RUN(launch_ln_bwd(l, st));
  }
  const bf16_t* df2 = d_ff2.thr ? w.dr2m : w.dr2;
  if (!fused) {  // dh1 = (df2 . W2) * gelu'(h1) * drop1 ; db1 = colsum(dh1)
    GemmNT g; g.X = df2; g.ldx = D; g.W = lw.w2_kn; g.ldw = D; g.M = rows_q; g.N = F; g.K = D;
    g.epi.act = 2; g.epi.aux = b.h1; g.epi.ldaux = F; g.epi.colsum = G + lp.b1; g.epi.out = w.dh1; g.epi.ldc = F;
    epi_drop(g.epi, mkdrop(train, pdrop, seed, site_base + SITE_FF1), F);
    RUN(launch_gemm_nt(g, st));
  }
  { GemmTN t; t.A = df2; t.lda = D; t.B = b.a1; t.ldb = F; t.T = rows_q; t.Mo = D; t.No = F; t.C = G + lp.w2; t.ldc = F; RUN(launch_gemm_tn(t, st)); }
  if (!fused) {  // dz1 = dh1 . W1 + dr2
    GemmNT g; g.X = w.dh1; g.ldx = F; g.W = lw.w1_kn; g.ldw = F; g.M = rows_q; g.N = D; g.K = F;
    g.epi.res = w.dr2; g.epi.ldres = D; g.epi.out = w.dz1; g.epi.ldc = D;
    RUN(launch_gemm_nt(g, st));
  }
  { GemmTN t; t.A = w.dh1; t.lda = F; t.B = b.z1; t.ldb = D; t.T = rows_q; t.Mo = F; t.No = D; t.C = G + lp.w1; t.ldc = D; RUN(launch_gemm_tn(t, st)); }
  if (!fused) {
    LnBwd l; l.dy = w.dz1; l.lddy = D; l.x = b.r1; l.x_f32 = 0; l.ldx = D; l.gain = P + lp.ln1g; l.R = rows_q; l.D = D;
    l.dx = w.dr1; l.lddx = D; l.dgain = G + lp.ln1g; l.dbias = G + lp.ln1b; l.dxcolsum = G + lp.bo;
    l.drop = mkdrop(train, pdrop, seed, site_base + SITE_POSTLN);
    RUN(launch_ln_bwd(l, st));
  }
  if (!fused) {
    GemmNT g; g.X = w.dr1; g.ldx = D; g.W = lw.wo_kn; g.ldw = D; g.M = rows_q; g.N = D; g.K = D; g.epi.out = w.dctx; g.epi.ldc = D;
    RUN(launch_gemm_nt(g, st));
  }
  { GemmTN t; t.A = w.dr1; t.lda = D; t.B = b.ctx; t.ldb = D; t.T = rows_q; t.Mo = D; t.No = D; t.C = G + lp.wo; t.ldc = D; RUN(launch_gemm_tn(t, st)); }
  AttnArgs a; a.q = b.q; a.ldq = b.ldq; a.k = b.k; a.ldk = b.ldk; a.v = b.v; a.ldv = b.ldv; a.o = b.ctx; a.ldo = D; a.lse = b.lse;
  a.H = H; a.dh = dh; a.scale = 1.0f / sqrtf((float)dh);
  a.drop = mkdrop(train, pdrop, seed, site_base + SITE_ATTN);
  a.dout = w.dctx; a.lddo = D; a.delta = w.delta; a.dq = w.dq; a.lddq = w.lddq; a.dk = w.dk; a.lddk = w.lddk; a.dv = w.dv; a.lddv = w.lddv;
  RUN(attention_all(a, sg, !self, true, st));
  if (self) {
    // bq | bk | bv gradients = column sums of dqkv: taken by the weight-gradient GEMM that streams dqkv anyway
    { GemmTN t; t.A = w.dq; t.lda = 3 * D; t.B = xq; t.ldb = D; t.T = rows_q; t.Mo = 3 * D; t.No = D; t.C = G + lp.wqkv; t.ldc = D;
      t.a_colsum = G + lp.bq; RUN(launch_gemm_tn(t, st)); }
    if (qkv_bwd_is_fused(lw, rows_q, dxq_f32 != nullptr, part_ws)) {
      QkvBwd f; f.T = rows_q; f.dqkv = w.dq; f.wqkv = lw.f_wqkv_kn; f.res = w.dr1; f.aux = gelu_aux; f.dz = dxq; f.colsum = gelu_colsum; f.part = part_ws;
      f.colsum_overwrite = 1;  // the column sums are written, not accumulated: the caller skips the zero fill on this path
      RUN(launch_qkv_bwd(f, st));
    } else {
      GemmNT g; g.X = w.dq; g.ldx = 3 * D; g.W = lw.wqkv_kn; g.ldw = 3 * D; g.M = rows_q; g.N = D; g.K = 3 * D;
      g.epi.res = w.dr1; g.epi.ldres = D;
      if (gelu_aux) { g.epi.act = 2; g.epi.aux = gelu_aux; g.epi.ldaux = D; g.epi.colsum = gelu_colsum; }
      if (dxq_f32) { g.epi.out = dxq_f32; g.epi.out_f32 = 1; } else g.epi.out = dxq;
      g.epi.ldc = D;
      RUN(launch_gemm_nt(g, st));
    }
  } else {
    { GemmTN t; t.A = w.dq; t.lda = w.lddq; t.B = xq; t.ldb = D; t.T = rows_q; t.Mo = D; t.No = D; t.C = G + lp.wqkv; t.ldc = D;
      t.a_colsum = G + lp.bq; RUN(launch_gemm_tn(t, st)); }
    { GemmTN t; t.A = w.dk; t.lda = w.lddk; t.B = xkv; t.ldb = D; t.T = rows_kv; t.Mo = 2 * D; t.No = D; t.C = G + lp.wqkv + (size_t)D * D; t.ldc = D;
      t.a_colsum = G + lp.bk; RUN(launch_gemm_tn(t, st)); }
    {  // dxq = dq . Wq + dr1
      GemmNT g; g.X = w.dq; g.ldx = w.lddq; g.W = lw.wqkv_kn; g.ldw = 3 * D; g.M = rows_q; g.N = D; g.K = D;
      g.epi.res = w.dr1; g.epi.ldres = D;
      if (dxq_f32) { g.epi.out = dxq_f32; g.epi.out_f32 = 1; } else g.epi.out = dxq;
      g.epi.ldc = D;
      RUN(launch_gemm_nt(g, st));
    }
    {  // dxkv += dkv . Wkv
      GemmNT g; g.X = w.dk; g.ldx = w.lddk; g.W = lw.wqkv_kn + D; g.ldw = 3 * D; g.M = rows_kv; g.N = D; g.K = 2 * D;
      g.epi.res = dxkv_accum; g.epi.ldres = D; g.epi.out = dxkv_accum; g.epi.ldc = D;
      RUN(launch_gemm_nt(g, st));
    }
  }
  return 0;
}

static int g_use_glob_fused_bwd = 1;  // coot_set_option("glob_fused_bwd", 0/1): their backward likewise (glob_bwd_kernel + one batched weight-gradient GEMM)
static int g_use_glob_fused = 1;  // coot_set_option("glob_fused", 0/1): context networks as one launch per pass (fused.hip: glob_fwd_kernel)
// the shape every shipped global network has: d_model 384, 8 heads, no input FC, one encoder layer + one context layer, avg_special
static bool glob_fused_ok(const coot_net_config& c, const Segs& sg) {
  return g_use_fused && g_use_glob_fused && fused_layer_ok(c) && c.num_heads == 8 && c.use_context && !c.use_input_fc && c.pooler != 0 &&
         c.num_layers == 1 && c.ctx_num_layers == 1 && sg.n == 1 && glob_fwd_supported(sg.L[0]);
}

static int g_use_packed = 1;  // coot_set_option("packed", 0/1): honour cu_seqlens (0: process the padded layout even when the caller passes them)
// Packed (varlen) token rows are taken by the fused local-network path: input LayerNorm (gathers the rows), the token-tile chains
// (row independent), the short attention and the pooling kernels (sequence offsets from cu_seqlens).  Forward and backward take the
// same decision from the same arguments.
static bool packed_ok(const coot_net_config& c, const WPack& W, const Segs& sg, const coot_packed_seqs* pk) {
  if (!g_use_packed || !pk || !pk->cu_seqlens || pk->total_tokens <= 0) return false;
  return g_use_fused && g_use_fused_infc && c.use_input_fc && !c.use_context && fused_pool_ok(c) && W.f_in_w && W.f_pw1 &&
         W.f_pw1_kn && c.input_dim % 64 == 0 && pk->total_tokens >= g_fused_min_rows && pk->total_tokens <= sg.Tpad() &&
         attn_short_path(sg.L[0]) && (sg.n == 1 || attn_short_path(sg.L[1]));
}

static LayerBufs self_bufs(const LayerS& s, int D) {
  LayerBufs b; b.q = s.qkv; b.ldq = 3 * D; b.k = s.qkv + D; b.ldk = 3 * D; b.v = s.qkv + 2 * D; b.ldv = 3 * D;
  b.ctx = s.ctx; b.r1 = s.r1; b.z1 = s.z1; b.h1 = s.h1; b.a1 = s.a1; b.r2 = s.r2; b.z2 = s.z2; b.lse = s.lse;
  return b;
}
static LayerBufs ctx_bufs(const CtxS& s, int D) {
  LayerBufs b; b.q = s.q; b.ldq = D; b.k = s.kv; b.ldk = 2 * D; b.v = s.kv + D; b.ldv = 2 * D;
  b.ctx = s.cctx; b.r1 = s.r1; b.z1 = s.z1; b.h1 = s.h1; b.a1 = s.a1; b.r2 = s.r2; b.z2 = s.z2; b.lse = s.lse;
  return b;
}

}  // namespace coot

using namespace coot;

extern "C" {

const char* coot_last_error(void) { return coot::g_err; }
int coot_version(void) { return COOT_ABI_VERSION; }
// host-only (tests): the keep-scales (0 or 1 / keep) the kernels draw for n consecutive elements idx0 .. of an element-wise
// dropout site, resp. for the attention probabilities (row32, k = 0 .. n - 1) — evaluated by the SAME functions (common.h) and
// the same quantisation of p (mkdrop) as on the device
int coot_debug_dropout_scales(uint64_t seed, unsigned site, uint64_t idx0, int64_t n, float p, float* out_host) {
  COOT_REQUIRE(out_host && n >= 0 && p > 0.f, "debug_dropout_scales: bad arguments");
  const DropCfg d = mkdrop(1, p, seed, site);
  const unsigned key = drop_key(seed, site);
  for (int64_t i = 0; i < n; ++i) out_host[i] = drop_scale_key(key, idx0 + (uint64_t)i, d.thr, d.inv_keep);
  return 0;
}
int coot_debug_attn_dropout_scales(uint64_t seed, unsigned site, unsigned row32, int Lk, float p, float* out_host) {
  COOT_REQUIRE(out_host && Lk >= 0 && p > 0.f, "debug_attn_dropout_scales: bad arguments");
  const DropCfg d = mkdrop(1, p, seed, site);
  const unsigned key = drop_key(seed, site);
  for (int k = 0; k < Lk; ++k) out_host[k] = attn_drop_scale(key, row32, k, (unsigned)(Lk + 1) >> 1, d.thr, d.inv_keep);
  return 0;
}
int coot_debug_timestamps(void* dev_u64) { g_fz_tstamps = (unsigned long long*)dev_u64; return 0; }
extern "C" void coot_step_stamps_enable(int on);  // api_step.hip
extern "C" void coot_step_grad_write(int on);
extern "C" int coot_internal_stage_hits(void);
extern "C" int coot_internal_stream_counter(int which);  // api_step.hip: StreamPicker
extern "C++" { namespace coot { int det_bypass_count(); } }  // det.hip
int coot_get_option(const char* name, int* value) {
  if (!value) { set_error("get_option: null result"); return -1; }
  if (!strcmp(name, "tn_dma")) { *value = get_tn_dma(); return 0; }
  if (!strcmp(name, "xcd_order")) { *value = get_xcd_order(); return 0; }
  if (!strcmp(name, "tn_mode")) { *value = get_tn_mode(); return 0; }
  if (!strcmp(name, "stage_hits")) { *value = coot_internal_stage_hits(); return 0; }  // steps of this thread that used a prepared input stage
  if (!strcmp(name, "fused_attn_launches")) { *value = fused_attn_launches(); return 0; }
  // concurrent-stream selection of this thread (coot_stream_create_concurrent, the library's own stream): spin-kernel tests run,
  // candidates that shared a hardware queue with a stream they have to run beside, selections that found no concurrent stream at all
  if (!strcmp(name, "stream_overlap_tests")) { *value = coot_internal_stream_counter(0); return 0; }
  if (!strcmp(name, "stream_candidates_rejected")) { *value = coot_internal_stream_counter(1); return 0; }
  if (!strcmp(name, "stream_unresolved")) { *value = coot_internal_stream_counter(2); return 0; }
  if (!strcmp(name, "operand_f16")) { *value = COOT_OPERAND_IS_F16; return 0; }  // which build this is (common.h)
  if (!strcmp(name, "det_bypasses")) { *value = det_bypass_count(); return 0; }  // deterministic mode: addends that took the float atomic (synchronises; -1: mode off)
  set_error("get_option: unknown or write-only option %s", name);
  return -2;
}
int coot_set_option(const char* name, int value) {
  if (!strcmp(name, "tn_mode")) { set_tn_mode(value); return 0; }
  if (!strcmp(name, "step_stamps")) { coot_step_stamps_enable(value); return 0; }
  if (!strcmp(name, "fused")) { g_use_fused = value; return 0; }
  if (!strcmp(name, "glob_fused")) { g_use_glob_fused = value; return 0; }
  if (!strcmp(name, "glob_fused_bwd")) { g_use_glob_fused_bwd = value; return 0; }
  if (!strcmp(name, "packed")) { g_use_packed = value; return 0; }
  if (!strcmp(name, "fused_infc")) { g_use_fused_infc = value; return 0; }
  if (!strcmp(name, "fused_attn")) { set_fused_attn(value); return 0; }  // 0: the self-attention of the local networks stays a launch of its own
  if (!strcmp(name, "tn_dma")) { set_tn_dma(value); return 0; }
  if (!strcmp(name, "pack_lazy")) { g_pack_lazy = value; return 0; }
  if (!strcmp(name, "pack_poison")) { g_pack_poison = value; return 0; }
  if (!strcmp(name, "grad_poison")) { g_grad_poison = value; return 0; }
  if (!strcmp(name, "tn_target_wgs")) { set_tn_target_wgs(value); return 0; }
  if (!strcmp(name, "xcd_order")) { set_xcd_order(value); return 0; }
  if (!strcmp(name, "grad_write")) { coot_step_grad_write(value); return 0; }
  if (!strcmp(name, "glob_flush_aux")) { g_glob_flush_aux = value; return 0; }
  if (!strcmp(name, "pool_handover")) { g_pool_handover_on = value; return 0; }  // 0: the pack between the local and the global forward stays a launch of its own  // 0: the global network's weight-gradient launch stays on its side's stream
  if (!strcmp(name, "cl_col_split")) { set_cl_col_split(value); return 0; }
  if (!strcmp(name, "cl_small")) { set_cl_small(value); return 0; }
  if (!strcmp(name, "fz_debug")) { g_fz_debug = value; return 0; }
  if (!strcmp(name, "fused_min_rows")) { g_fused_min_rows = value; return 0; }
  set_error("unknown option %s", name);
  return -2;
}

int64_t coot_net_param_numel(const coot_net_config* cfg) {
  coot_net_config c; if (norm_cfg(cfg, &c)) return -1;
  NetLayout L; build_layout(c, L); return L.total;
}
int coot_net_param_count(const coot_net_config* cfg) {
  coot_net_config c; if (norm_cfg(cfg, &c)) return -1;
  NetLayout L; build_layout(c, L); return (int)L.entries.size();
}
int coot_net_param_info(const coot_net_config* cfg, int index, char* name, int name_len, int64_t* offset, int64_t shape[4], int* ndim) {
  coot_net_config c; RUN(norm_cfg(cfg, &c));
  NetLayout L; build_layout(c, L);
  COOT_REQUIRE(index >= 0 && index < (int)L.entries.size(), "param index %d out of range", index);
  const PEntry& e = L.entries[index];
  snprintf(name, name_len, "%s", e.name.c_str());
  *offset = e.off; *ndim = e.ndim;
  for (int i = 0; i < 4; ++i) shape[i] = e.shape[i];
  return 0;
}
int coot_net_out_dim(const coot_net_config* cfg) { return cfg->hidden_dim * (cfg->use_context ? 2 : 1); }

size_t coot_net_wpack_bytes(const coot_net_config* cfg) {
  coot_net_config c; if (norm_cfg(cfg, &c)) return 0;
  Arena A(nullptr, 0); WPack W; layout_wpack(c, A, W); return A.off + 256;
}

// Records the pack jobs of one network into `jobs`, offsets relative to (P0, wpack0) — a base at or below every network of the
// launch, so one launch can carry several networks (coot_nets_pack_weights); a full job table is flushed on the way.
// Lazy per-op layouts.  Every weight exists in up to four bf16 layouts: [n][k] and [k][n] for the per-op GEMMs, and the two
// fragment-major P48 images of the fused kernels.  A network that runs on the fused kernels (every launch of the benchmark
// shapes) never reads the first two, yet rebuilding them after every optimizer step was half of the pack launch (52 us at the
// end of the video side's critical path, ~170 MB of the step's HBM traffic).  coot_set_option("pack_lazy", 1) (default): the
// repack writes the fused images only — and the per-op ones of weights that have no fused image; the per-op layouts of the
// pack are marked stale in a host-side table, and the first per-op GEMM that reads one (launch_gemm_nt's weight hook, armed by
// coot_net_fwd / coot_net_bwd) packs them on its stream first.  A pack that needed them once keeps packing them eagerly.
// coot_set_option("pack_poison", 1) (tests) fills stale per-op layouts with NaN patterns so that a reader the hook misses
// cannot go unnoticed.
enum { PACK_ALL = 0, PACK_FUSED = 1, PACK_PEROP_REST = 2 };
struct PackState { bool perop_fresh = false, perop_wanted = false; };  // a pack this table has never seen: assume nothing (first per-op use completes it)
static std::mutex g_pack_mu;
static std::unordered_map<const void*, PackState> g_pack_state;
static PackState pack_state_get(const void* wpack) { std::lock_guard<std::mutex> lk(g_pack_mu); auto it = g_pack_state.find(wpack); return it == g_pack_state.end() ? PackState() : it->second; }
static void pack_state_set(const void* wpack, PackState s) { std::lock_guard<std::mutex> lk(g_pack_mu); g_pack_state[wpack] = s; }

static int pack_net_jobs(const coot_net_config* cfg, const float* P, void* wpack, const float* P0, void* wpack0, PackJobs& jobs, hipStream_t st,
                         int mode = PACK_ALL, bool* has_lazy = nullptr) {
  coot_net_config c; RUN(norm_cfg(cfg, &c));
  const int64_t pd = P - P0;  // >= 0
  NetLayout L; build_layout(c, L);
  Arena A(wpack, (size_t)-1); WPack W; layout_wpack(c, A, W);
  const int D = c.hidden_dim, F = c.ff_dim, Din = c.input_dim;
  auto flush = [&]() -> int { int rc = launch_pack_jobs(P0, wpack0, jobs, st); jobs.n = 0; jobs.mv = PackMatvec(); return rc; };
  auto add = [&](int64_t src_off, long lds, int R, int Cc, bf16_t* dst, long ldd, int transpose, int64_t colscale_off) -> int {
    if (jobs.n == 56) RUN(flush());
    PackJob& j = jobs.j[jobs.n++];
    j.src_off = src_off + pd; j.dst_byte_off = (char*)dst - (char*)wpack0; j.R = R; j.C = Cc; j.lds = lds; j.ldd = ldd;
    j.transpose = transpose; j.p48 = 0; j.colscale_off = colscale_off >= 0 ? colscale_off + pd : -1;
    return 0;
  };
  // logical [N, K] = src (transpose == 0) or src^T (transpose == 1), written in the P48 layout of the fused kernels
  auto add48 = [&](int64_t src_off, long lds, int R, int Cc, bf16_t* dst, int transpose) -> int {
    RUN(add(src_off, lds, R, Cc, dst, 0, transpose, -1));
    jobs.j[jobs.n - 1].p48 = 1;
    return 0;
  };
  // per-op layouts of a weight group: always with PACK_ALL; with PACK_FUSED only when the group has no fused image; with
  // PACK_PEROP_REST exactly the ones PACK_FUSED left out.  Fused images: PACK_ALL and PACK_FUSED.
  bool lazy = false;
  auto perop = [&](bool has_fused) { if (has_fused && mode == PACK_FUSED) lazy = true; return mode == PACK_ALL || (mode == PACK_FUSED ? !has_fused : has_fused); };
  const bool fused_on = mode != PACK_PEROP_REST;
  auto poison = [&](bf16_t* dst, size_t elems) -> int {
    if (!g_pack_poison || mode != PACK_FUSED) return 0;
    return check_hip(hipMemsetAsync(dst, 0xFF, elems * sizeof(bf16_t), st), "poison");
  };
  if (c.use_input_fc) {
    if (perop(W.f_in_w != nullptr)) RUN(add(L.in_w, Din, D, Din, W.in_w, Din, 0, L.n_gain));  // W * gain: LN affine folded into the FC
    else RUN(poison(W.in_w, (size_t)D * Din));
    if (W.f_in_w && fused_on) { RUN(add(L.in_w, Din, D, Din, W.f_in_w, 0, 0, L.n_gain)); jobs.j[jobs.n - 1].p48 = 1; }
    // folded bias b' = b + W . norm_bias: rides on the (last) pack launch of this call
    if (fused_on) { jobs.mv.W = P + L.in_w; jobs.mv.ldw = Din; jobs.mv.N = D; jobs.mv.K = Din; jobs.mv.v = P + L.n_bias; jobs.mv.b = P + L.in_b; jobs.mv.out = W.in_bias; }
  }
  auto pack_layer = [&](const LayerP& lp, const LayerW& lw) -> int {
    if (perop(lw.f_wo != nullptr)) {
      RUN(add(lp.wqkv, D, 3 * D, D, lw.wqkv_nk, D, 0, -1));
      RUN(add(lp.wqkv, D, 3 * D, D, lw.wqkv_kn, 3 * D, 1, -1));
      RUN(add(lp.wo, D, D, D, lw.wo_nk, D, 0, -1));
      RUN(add(lp.wo, D, D, D, lw.wo_kn, D, 1, -1));
      RUN(add(lp.w1, D, F, D, lw.w1_nk, D, 0, -1));
      RUN(add(lp.w1, D, F, D, lw.w1_kn, F, 1, -1));
      RUN(add(lp.w2, F, D, F, lw.w2_nk, F, 0, -1));
      RUN(add(lp.w2, F, D, F, lw.w2_kn, D, 1, -1));
    } else {
      RUN(poison(lw.wqkv_nk, (size_t)(6 * D * D + 4 * F * D)));  // wqkv_nk .. w2_kn are contiguous (layout_wpack)
    }
    if (lw.f_wo && fused_on) {
      RUN(add48(lp.wo, D, D, D, lw.f_wo, 0));
      RUN(add48(lp.w1, D, F, D, lw.f_w1, 0));
      RUN(add48(lp.w2, F, D, F, lw.f_w2, 0));
      // dX orientation: logical [n = input feature][k = output feature] = W^T
      RUN(add48(lp.wo, D, D, D, lw.f_wo_kn, 1));
      RUN(add48(lp.w1, D, F, D, lw.f_w1_kn, 1));
      RUN(add48(lp.w2, F, D, F, lw.f_w2_kn, 1));
      RUN(add48(lp.wqkv, D, 3 * D, D, lw.f_wqkv, 0));
      RUN(add48(lp.wqkv, D, 3 * D, D, lw.f_wqkv_kn, 1));
    }
    return 0;
  };
  for (int i = 0; i < c.num_layers; ++i) RUN(pack_layer(L.layers[i], W.layers[i]));
  for (size_t i = 0; i < L.ctx.size(); ++i) RUN(pack_layer(L.ctx[i], W.ctx[i]));
  if (c.pooler == 0) {
    const int H = c.pool_heads, PH = c.pool_hidden, dhp = PH / H, dop = D / H;
    const bool pool_perop = perop(W.f_pw1 != nullptr);
    if (!pool_perop) RUN(poison(W.pw1_nk, (size_t)(2 * PH * D + 2 * PH * dop)));  // pw1_nk .. pw2_kn are contiguous
    for (int h = 0; h < H; ++h) {
      if (pool_perop) {
        // W1[h]: [D, dhp].  nk: rows (h*dhp + e) = W1[h][:, e]  -> [PH, D].  kn: [D, PH] with row d = concat_h W1[h][d][:]
        RUN(add(L.pw1 + (int64_t)h * D * dhp, dhp, D, dhp, W.pw1_nk + (size_t)h * dhp * D, D, 1, -1));
        RUN(add(L.pw1 + (int64_t)h * D * dhp, dhp, D, dhp, W.pw1_kn + (size_t)h * dhp, PH, 0, -1));
        // W2[h]: [dhp, dop].  nk (fwd): [dop, dhp] per head.  kn (dX): natural [dhp, dop] per head
        RUN(add(L.pw2 + (int64_t)h * dhp * dop, dop, dhp, dop, W.pw2_nk + (size_t)h * dop * dhp, dhp, 1, -1));
        RUN(add(L.pw2 + (int64_t)h * dhp * dop, dop, dhp, dop, W.pw2_kn + (size_t)h * dhp * dop, dop, 0, -1));
      }
      if (W.f_pw1 && fused_on) {  // logical [n = pool feature e][k = d] = W1[h]^T, [n = o][k = e] = W2[h]^T
        RUN(add48(L.pw1 + (int64_t)h * D * dhp, dhp, D, dhp, W.f_pw1 + (size_t)h * dhp * D, 1));
        RUN(add48(L.pw2 + (int64_t)h * dhp * dop, dop, dhp, dop, W.f_pw2 + (size_t)h * dop * dhp, 1));
        // dX orientation: dz[d] += sum_e dhp[e] W1[h][d][e];  dhp[e] = sum_o ds[o] W2[h][e][o]  (the sources are already [n][k])
        RUN(add48(L.pw1 + (int64_t)h * D * dhp, dhp, D, dhp, W.f_pw1_kn + (size_t)h * D * dhp, 0));
        RUN(add48(L.pw2 + (int64_t)h * dhp * dop, dop, dhp, dop, W.f_pw2_kn + (size_t)h * dhp * dop, 0));
      }
    }
  }
  if (has_lazy) *has_lazy = lazy;
  return 0;
}

// ---- the weight hook of the lazily packed per-op layouts ----------------------------------------------------------------------
struct PerOpGuard { bool armed = false; coot_net_config cfg; const float* P = nullptr; void* wpack = nullptr; const char* lo[8]; const char* hi[8]; int n = 0; };
static thread_local PerOpGuard g_guard;
static int perop_weight_hook(const void* Wp, hipStream_t st) {
  PerOpGuard& G = g_guard;
  if (!G.armed) return 0;
  bool hit = false;
  for (int i = 0; i < G.n && !hit; ++i) hit = (const char*)Wp >= G.lo[i] && (const char*)Wp < G.hi[i];
  if (!hit) return 0;
  G.armed = false;
  PackJobs jobs;
  RUN(pack_net_jobs(&G.cfg, G.P, G.wpack, G.P, G.wpack, jobs, st, PACK_PEROP_REST));
  RUN(launch_pack_jobs(G.P, G.wpack, jobs, st));
  PackState ps; ps.perop_fresh = true; ps.perop_wanted = true;
  pack_state_set(G.wpack, ps);
  return 0;
}
// arms the hook for the scope of one coot_net_fwd / coot_net_bwd call when the network's per-op layouts are stale
struct PerOpGuardScope {
  PerOpGuardScope(const coot_net_config& c, const WPack& W, const float* P, const void* wpack) {
    PerOpGuard& G = g_guard;
    G.armed = false;
    if (pack_state_get(wpack).perop_fresh) return;
    G.cfg = c; G.P = P; G.wpack = (void*)wpack; G.n = 0;
    const size_t D = c.hidden_dim, F = c.ff_dim, Din = c.input_dim;
    auto range = [&](const bf16_t* p, size_t elems) { if (p && G.n < 8) { G.lo[G.n] = (const char*)p; G.hi[G.n] = (const char*)(p + elems); ++G.n; } };
    if (c.use_input_fc && W.f_in_w) range(W.in_w, D * Din);
    for (const LayerW& lw : W.layers) if (lw.f_wo) range(lw.wqkv_nk, 6 * D * D + 4 * F * D);
    for (const LayerW& lw : W.ctx) if (lw.f_wo) range(lw.wqkv_nk, 6 * D * D + 4 * F * D);
    if (c.pooler == 0 && W.f_pw1) range(W.pw1_nk, 2 * (size_t)c.pool_hidden * D + 2 * (size_t)c.pool_hidden * (D / c.pool_heads));
    G.armed = G.n > 0;
    set_gemm_weight_hook(perop_weight_hook);
  }
  ~PerOpGuardScope() { g_guard.armed = false; }
};

int coot_nets_pack_weights(int n, const coot_net_config* const* cfgs, const float* const* Ps, void* const* wpacks, coot_stream_t stream) {
  COOT_REQUIRE(n >= 1 && cfgs && Ps && wpacks, "nets_pack_weights: bad arguments");
  const float* P0 = Ps[0]; char* w0 = (char*)wpacks[0];
  for (int i = 1; i < n; ++i) { if (Ps[i] < P0) P0 = Ps[i]; if ((char*)wpacks[i] < w0) w0 = (char*)wpacks[i]; }
  for (int i = 0; i < n; ++i)
    COOT_REQUIRE(((const char*)Ps[i] - (const char*)P0) % 4 == 0 && ((char*)wpacks[i] - w0) % 16 == 0, "nets_pack_weights: unaligned arenas");
  PackJobs jobs;
  for (int i = 0; i < n; ++i) {
    if (jobs.mv.W && cfgs[i]->use_input_fc) { RUN(launch_pack_jobs(P0, w0, jobs, (hipStream_t)stream)); jobs.n = 0; jobs.mv = PackMatvec(); }  // one rider per launch
    PackState ps = pack_state_get(wpacks[i]);
    const int mode = (g_pack_lazy && !ps.perop_wanted) ? PACK_FUSED : PACK_ALL;
    bool lazy = false;
    RUN(pack_net_jobs(cfgs[i], Ps[i], wpacks[i], P0, w0, jobs, (hipStream_t)stream, mode, &lazy));
    ps.perop_fresh = !lazy;
    pack_state_set(wpacks[i], ps);
  }
  return launch_pack_jobs(P0, w0, jobs, (hipStream_t)stream);
}
int coot_net_pack_weights(const coot_net_config* cfg, const float* P, void* wpack, coot_stream_t stream) {
  const coot_net_config* cfgs[1] = {cfg}; const float* Ps[1] = {P}; void* ws[1] = {wpack};
  return coot_nets_pack_weights(1, cfgs, Ps, ws, stream);
}

size_t coot_net_saved_bytes(const coot_net_config* cfg, int N, int Lseq, int N2, int L2) {
  coot_net_config c; if (norm_cfg(cfg, &c)) return 0;
  if (c.dtype == COOT_DTYPE_F32) {  // the reference mode's fp32 workspace
    NetLayout L; build_layout(c, L);
    return ref_f32_workspace_bytes(ref_desc(c, L, N + N2), (long)N * Lseq + (long)N2 * L2);
  }
  Arena A(nullptr, 0); Saved S; layout_saved(c, N + N2, (long)N * Lseq + (long)N2 * L2, A, S); return A.off + 256;
}
size_t coot_net_scratch_bytes(const coot_net_config* cfg, int N, int Lseq, int N2, int L2) {
  coot_net_config c; if (norm_cfg(cfg, &c)) return 0;
  if (c.dtype == COOT_DTYPE_F32) {  // the reference mode's backward temporaries
    NetLayout L; build_layout(c, L);
    return ref_f32_scratch_bytes(ref_desc(c, L, N + N2), (long)N * Lseq + (long)N2 * L2);
  }
  Arena A(nullptr, 0); Scratch S; layout_scratch(c, N + N2, (long)N * Lseq + (long)N2 * L2, A, S); return A.off + 256;
}

// ---- gradients written instead of accumulated ------------------------------------------------------------------------------------
// coot_net_bwd ACCUMULATES into the gradient arena (the contract of the autograd route and of gradient accumulation), so a step
// used to zero all 30 MB of it and every weight-gradient launch read its destination back.  Every weight MATRIX gradient is the
// result of exactly one weight-gradient problem per pass: coot_net_grads_overwrite(1) (thread-local; coot_train_step sets it
// around its backward) makes those problems write, and coot_nets_zero_grads(..., skip_matrices = 1) zeroes only what is still
// accumulated (biases, LayerNorm parameters: < 1 % of the arena) — one launch instead of four fills.
static thread_local int g_grad_overwrite = 0;
int coot_net_grads_overwrite(int on) { g_grad_overwrite = on ? 1 : 0; return 0; }

struct ZeroRanges { float* p[64]; int n[64]; int poison[64]; int count; };
__global__ __launch_bounds__(256) void zero_ranges_kernel(ZeroRanges r) {
  float* p = r.p[blockIdx.x];
  const int n = r.n[blockIdx.x];
  const float v = r.poison[blockIdx.x] ? __uint_as_float(0x7FC00000u) : 0.f;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) p[i] = v;
}
// the gradient matrices a backward pass WRITES in overwrite mode, as (offset, elements) in ascending order
static void written_matrices(const coot_net_config& c, const NetLayout& L, std::vector<std::pair<int64_t, int64_t>>& mats) {
  const int64_t D = c.hidden_dim, F = c.ff_dim, Din = c.input_dim;
  if (c.use_input_fc) mats.push_back({L.in_w, D * Din});
  auto layer = [&](const LayerP& lp) { mats.push_back({lp.wqkv, 3 * D * D}); mats.push_back({lp.wo, D * D}); mats.push_back({lp.w1, F * D}); mats.push_back({lp.w2, D * F}); };
  for (const LayerP& lp : L.layers) layer(lp);
  for (const LayerP& lp : L.ctx) layer(lp);
  if (c.pooler == 0) { mats.push_back({L.pw1, D * (int64_t)c.pool_hidden}); mats.push_back({L.pw2, (int64_t)c.pool_hidden * (D / c.pool_heads)}); }
  std::sort(mats.begin(), mats.end());
}
// host-only (tests): the ranges coot_nets_zero_grads(skip_matrices = 1) leaves alone; returns their number (or -1)
int coot_debug_written_matrices(const coot_net_config* cfg, int64_t* offsets, int64_t* sizes, int max_ranges) {
  coot_net_config c; if (norm_cfg(cfg, &c)) return -1;
  NetLayout L; build_layout(c, L);
  std::vector<std::pair<int64_t, int64_t>> mats; written_matrices(c, L, mats);
  if ((int)mats.size() > max_ranges) return -1;
  for (size_t i = 0; i < mats.size(); ++i) { offsets[i] = mats[i].first; sizes[i] = mats[i].second; }
  return (int)mats.size();
}
int coot_nets_zero_grads(int nnets, const coot_net_config* const* cfgs, float* const* grads, int skip_matrices, coot_stream_t stream) {
  return coot_nets_zero_grads_ex(nnets, cfgs, grads, skip_matrices, nullptr, nullptr, 0, stream);
}
// ... plus n_extra more fp32 ranges in the same launch (the step's embedding-gradient block and loss words)
int coot_nets_zero_grads_ex(int nnets, const coot_net_config* const* cfgs, float* const* grads, int skip_matrices, float* const* extra,
                            const int64_t* extra_n, int n_extra, coot_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  COOT_REQUIRE(nnets >= 1 && nnets <= 8 && cfgs && grads && (n_extra == 0 || (extra && extra_n)), "nets_zero_grads: bad arguments");
  ZeroRanges zr; zr.count = 0;
  auto flush = [&]() -> int {
    if (zr.count == 0) return 0;
    int big = 0;
    for (int i = 0; i < zr.count; ++i) if (zr.n[i] > big) big = zr.n[i];
    const int gy = big > (1 << 16) ? 64 : 2;  // vectors: two workgroups per range; a large block (embedding gradients, poison): 64
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(zr.count, gy), dim3(256), 0, st, zr);
    COOT_CHECK_LAUNCH("zero_ranges");
    zr.count = 0;
    return 0;
  };
  auto add = [&](float* p, int64_t n, int poison) -> int {
    if (n <= 0) return 0;
    if (zr.count == 64) RUN(flush());
    zr.p[zr.count] = p; zr.n[zr.count] = (int)n; zr.poison[zr.count] = poison; ++zr.count;
    return 0;
  };
  for (int k = 0; k < nnets; ++k) {
    coot_net_config c; RUN(norm_cfg(cfgs[k], &c));
    NetLayout L; build_layout(c, L);
    if (!skip_matrices) { RUN(check_hip(hipMemsetAsync(grads[k], 0, (size_t)L.total * sizeof(float), st), "memset grads")); continue; }
    // the matrices a backward pass writes (offset, elements), ascending offsets; everything between them is zeroed
    std::vector<std::pair<int64_t, int64_t>> mats; written_matrices(c, L, mats);
    int64_t pos = 0;
    for (const auto& m : mats) {
      COOT_REQUIRE(m.first >= pos && m.first + m.second <= L.total, "nets_zero_grads: parameter layout");
      RUN(add(grads[k] + pos, m.first - pos, 0));
      if (g_grad_poison) RUN(add(grads[k] + m.first, m.second, 1));
      pos = m.first + m.second;
    }
    RUN(add(grads[k] + pos, L.total - pos, 0));
  }
  for (int e = 0; e < n_extra; ++e)
    for (int64_t o = 0; o < extra_n[e]; o += (1 << 30)) RUN(add(extra[e] + o, extra_n[e] - o < (1 << 30) ? extra_n[e] - o : (1 << 30), 0));
  return flush();
}

// Input stage of a local network (api_step.hip: software-pipelined input LayerNorm).  With a stage set, the parameter-free x^ of the
// input LayerNorm (its gain / bias live in the packed input-FC weights) and the packed rows' position table live in the caller's
// stage instead of the saved arena.  mode 1: coot_net_fwd ONLY normalises (writes the stage) and returns; mode 2: the stage already
// holds this batch's x^ — the LayerNorm launch is skipped; mode 0: normalise into the stage and go on.  coot_net_bwd reads x^ there.
struct InputStage { bf16_t* xhat = nullptr; int* pos = nullptr; int mode = 0; };
thread_local InputStage g_input_stage;
void coot_internal_set_input_stage(void* xhat, void* pos, int mode) {
  g_input_stage.xhat = (bf16_t*)xhat; g_input_stage.pos = (int*)pos; g_input_stage.mode = mode;
}

int coot_net_fwd(const coot_net_config* cfg, const float* P, const void* wpack, const float* pe, const float* feats,
                 const int64_t* lengths, int N, int Lseq, const float* feats2, const int64_t* lengths2, int N2, int L2,
                 const float* hidden, float* pooled, float* per_token, void* saved,
                 size_t saved_bytes, void* scratch, size_t scratch_bytes, int train, uint64_t seed, const uint64_t* seed_dev,
                 coot_stream_t stream, const coot_packed_seqs* packed) {
  hipStream_t st = (hipStream_t)stream;
  struct SeedScope { SeedScope(const uint64_t* p) { g_seed_dev = (const unsigned long long*)p; } ~SeedScope() { g_seed_dev = nullptr; } } seedscope(seed_dev);
  coot_net_config c; RUN(norm_cfg(cfg, &c));
  COOT_REQUIRE(P && wpack && pe && feats && lengths && pooled && saved, "net_fwd: null pointer");
  if (packed && packed->source != COOT_SOURCE_PADDED && N2 > 0 && !feats2) feats2 = feats;  // (one packed matrix carries both segments)
  COOT_REQUIRE(!c.use_context || hidden, "net_fwd: context network needs hidden state (transformer_legacy.py:252)");
  COOT_REQUIRE(Lseq <= 1000 && L2 <= 1000, "net_fwd: sequence length %d exceeds positional table (max_len 1000)", Lseq);
  if (N <= 0) return 0;
  Segs sg; sg.N[0] = N; sg.L[0] = Lseq; sg.lens[0] = (const long long*)lengths;
  if (N2 > 0) {
    COOT_REQUIRE(feats2 && lengths2 && !c.use_context && !per_token, "net_fwd: second segment needs feats2/lengths2 (local networks, no per-token output)");
    sg.n = 2; sg.N[1] = N2; sg.L[1] = L2; sg.lens[1] = (const long long*)lengths2;
  }
  const int Ntot = sg.Ntot();
  NetLayout L; build_layout(c, L);
  if (c.dtype == COOT_DTYPE_F32) {  // fp32 reference mode (ref_f32.hip): the reference's op sequence in fp32
    COOT_REQUIRE(!train, "net_fwd: the f32 reference mode is an eval-mode checker (train must be 0)");
    COOT_REQUIRE(!(c.use_input_fc && g_input_stage.xhat != nullptr), "net_fwd: the f32 reference mode normalises its own input (no input stage may be announced)");
    COOT_REQUIRE(!packed || packed->source == COOT_SOURCE_PADDED, "net_fwd: the f32 reference mode reads the reference's padded batch");
    RefSegs rs; rs.n = sg.n;
    for (int s_ = 0; s_ < sg.n; ++s_) { rs.N[s_] = sg.N[s_]; rs.L[s_] = sg.L[s_]; rs.lens[s_] = sg.lens[s_]; }
    return ref_f32_forward(ref_desc(c, L, Ntot), P, pe, feats, feats2, rs, hidden, pooled, per_token, saved, saved_bytes, st);
  }
  Arena AW((void*)wpack, (size_t)-1); WPack W; layout_wpack(c, AW, W);
  PerOpGuardScope perop_guard(c, W, P, wpack);
  Arena AS(saved, saved_bytes); Saved S; layout_saved(c, Ntot, sg.Tpad(), AS, S);  // sized for the padded layout (>= the packed rows)
  COOT_REQUIRE(!AS.overflow, "net_fwd: saved buffer too small (%zu < %zu)", saved_bytes, AS.off);
  const bool staged = c.use_input_fc && g_input_stage.xhat != nullptr;
  const int stage_mode = staged ? g_input_stage.mode : 0;
  if (staged) { S.xhat = g_input_stage.xhat; S.pos = g_input_stage.pos; }
  const int source = packed ? packed->source : COOT_SOURCE_PADDED;
  COOT_REQUIRE(source >= COOT_SOURCE_PADDED && source <= COOT_SOURCE_PACKED_BF16, "net_fwd: packed source %d", source);
  if (packed_ok(c, W, sg, packed)) {
    COOT_REQUIRE(!per_token, "net_fwd: per-token output is not available with packed rows");
    sg.cu = packed->cu_seqlens; sg.Tp = packed->total_tokens;
  }
  COOT_REQUIRE(source == COOT_SOURCE_PADDED || sg.cu, "net_fwd: a packed feature source needs the packed-row path (local network on the fused "
               "kernels, >= %d tokens, sequences of <= 128 rows): there is no padded tensor to fall back to", g_fused_min_rows);
  const int D = c.hidden_dim, T = sg.T(), Din = c.input_dim, T0 = N * Lseq;
  const int out_dim = D * (c.use_context ? 2 : 1);

  bool qkv_done = false;  // the fused input-FC kernel also produces layer 0's q | k | v
  if (c.use_input_fc) {
    LnFwd l; l.x = feats; l.x_f32 = 1; l.ldx = Din; l.R = T0; l.D = Din; l.y = S.xhat; l.ldy = Din;
    if (sg.n > 1) { l.x2 = feats2; l.R0 = T0; l.R = T; }  // both segments in one launch
    if (sg.cu) { l.R = T; l.cu = sg.cu; l.nseq = Ntot; l.N0 = N; l.L0 = Lseq; l.L1 = L2; l.pos_out = S.pos; }  // gathers the valid rows
    if (source != COOT_SOURCE_PADDED) { l.src_packed = 1; l.x2 = nullptr; l.x_f32 = source == COOT_SOURCE_PACKED_F32; }  // ... or reads them in place
    l.nt = stage_mode == 1 && g_ln_nt;
    l.max_wgs = stage_mode == 1 ? ln_stream_cap((size_t)l.R * Din * (l.x_f32 ? 4 : 2)) : 0;
    if (stage_mode != 2) RUN(launch_ln_fwd(l, st));
    if (stage_mode == 1) return 0;
    if (W.f_in_w && W.layers[0].f_wqkv && g_use_fused && g_use_fused_infc && T >= g_fused_min_rows) {
      InfcQkvFwd f; f.T = T; f.Din = Din; f.xhat = S.xhat; f.win = W.f_in_w; f.bin = W.in_bias; f.pe = pe; f.T0 = T0; f.L1 = Lseq;
      f.pos = sg.cu ? S.pos : nullptr;
      f.L2 = sg.n > 1 ? L2 : Lseq; f.wqkv = W.layers[0].f_wqkv; f.bqkv = P + L.layers[0].bq; f.h0 = S.h0; f.z0 = S.z0; f.qkv = S.layers[0].qkv;
      f.tstamps = g_fz_tstamps;
      RUN(launch_infc_qkv_fwd(f, st));
      qkv_done = true;
    } else {
      GemmNT g; g.X = S.xhat; g.ldx = Din; g.W = W.in_w; g.ldw = Din; g.M = T; g.N = D; g.K = Din;
      g.epi.bias = W.in_bias; g.epi.act = 1; g.epi.save_pre = S.h0; g.epi.ldpre = D; g.epi.pe = pe; g.epi.pe_L = Lseq;
      g.epi.pe_T0 = T0; g.epi.pe_L2 = sg.n > 1 ? L2 : Lseq;
      g.epi.out = S.z0; g.epi.ldc = D;
      RUN(launch_gemm_nt(g, st));
    }
  } else if (!(glob_fused_ok(c, sg) && W.layers[0].f_wqkv && W.ctx[0].f_wqkv)) {  // (the single-launch context network normalises its input itself)
    LnFwd l; l.x = feats; l.x_f32 = 1; l.ldx = Din; l.R = T0; l.D = Din; l.gain = P + L.n_gain; l.bias = P + L.n_bias;
    l.pe = pe; l.pe_L = Lseq; l.y = S.z0; l.ldy = D;
    RUN(launch_ln_fwd(l, st));
    if (sg.n > 1) { l.x = feats2; l.R = T - T0; l.pe_L = L2; l.y = S.z0 + (size_t)T0 * D; RUN(launch_ln_fwd(l, st)); }
  }
  if (glob_fused_ok(c, sg) && W.layers[0].f_wqkv && W.ctx[0].f_wqkv) {
    // the whole context network in ONE launch (fused.hip: glob_fwd_kernel); saves the same tensors as the per-op path below
    GlobFwd f; f.B = N; f.Cmax = Lseq; f.x = feats; f.lens = (const long long*)lengths; f.hidden = hidden; f.pe = pe;
    f.n_gain = P + L.n_gain; f.n_bias = P + L.n_bias; f.z0 = S.z0; f.cq_in = S.cq_in; f.pooled = pooled; f.per_token = per_token; f.train = train; f.tstamps = g_fz_tstamps;
    auto fill = [&](GlobLayerFwd& g, const LayerP& lp, const LayerW& lw, const LayerBufs& b, float pdrop, unsigned site_base) {
      g.wqkv = lw.f_wqkv; g.wo = lw.f_wo; g.w1 = lw.f_w1; g.w2 = lw.f_w2;
      g.bqkv = P + lp.bq; g.bo = P + lp.bo; g.ln1g = P + lp.ln1g; g.ln1b = P + lp.ln1b; g.b1 = P + lp.b1; g.b2 = P + lp.b2;
      g.ln2g = P + lp.ln2g; g.ln2b = P + lp.ln2b;
      g.q = b.q; g.ldq = b.ldq; g.k = b.k; g.ldk = b.ldk; g.v = b.v; g.ldv = b.ldv;
      g.ctx = b.ctx; g.r1 = b.r1; g.z1 = b.z1; g.h1 = b.h1; g.a1 = b.a1; g.r2 = b.r2; g.z2 = b.z2; g.lse = b.lse;
      g.d_attn = mkdrop(train, pdrop, seed, site_base + SITE_ATTN); g.d_postln = mkdrop(train, pdrop, seed, site_base + SITE_POSTLN);
      g.d_ff1 = mkdrop(train, pdrop, seed, site_base + SITE_FF1); g.d_ff2 = mkdrop(train, pdrop, seed, site_base + SITE_FF2);
    };
    fill(f.self, L.layers[0], W.layers[0], self_bufs(S.layers[0], D), c.dropout, 0u);
    fill(f.ctx, L.ctx[0], W.ctx[0], ctx_bufs(S.ctx[0], D), c.ctx_dropout, 16u * 8);
    COOT_REQUIRE((c.dropout > 0.f) == (c.ctx_dropout > 0.f) || !train, "net_fwd: the fused context network takes dropout on both layers or on none");
    return launch_glob_fwd(f, st);
  }
  const bf16_t* z = S.z0;
  const bool pool_fused = g_use_fused && fused_pool_ok(c) && W.f_pw1 && !c.use_context && T >= g_fused_min_rows;
  for (int i = 0; i < c.num_layers; ++i) {
    LayerBufs b = self_bufs(S.layers[i], D);
    const bool last = (i == c.num_layers - 1);
    PoolFuse pf;
    if (last && pool_fused) {
      pf.pw1 = W.f_pw1; pf.pw2 = W.f_pw2; pf.pb1 = P + L.pb1; pf.pb2 = P + L.pb2; pf.hp = S.hp; pf.ap = S.ap; pf.s = S.s;
      pf.d1 = mkdrop(train, c.pool_dropout, seed, 16u * 15 + SITE_POOL1); pf.d2 = mkdrop(train, c.pool_dropout, seed, 16u * 15 + SITE_POOL2);
    }
    RUN(layer_fwd(c, P, L.layers[i], W.layers[i], z, T, z, T, sg, b, last ? per_token : nullptr, D, c.dropout,
                  train, seed, 16u * i, st, (last && pool_fused) ? &pf : nullptr, i == 0 && qkv_done));
    z = S.layers[i].z2;
  }
  if (c.use_context) {
    RUN(launch_cast_f32_bf16(hidden, D, N, D, S.cq_in, D, st));
    const bf16_t* cq = S.cq_in;
    for (int i = 0; i < c.ctx_num_layers; ++i) {
      LayerBufs b = ctx_bufs(S.ctx[i], D);
      const bool last = (i == c.ctx_num_layers - 1);
      RUN(layer_fwd(c, P, L.ctx[i], W.ctx[i], cq, N, z, T, sg, b, last ? pooled + D : nullptr, out_dim, c.ctx_dropout,
                    train, seed, 16u * (8 + i), st));
      cq = S.ctx[i].z2;
    }
  }
  if (c.pooler == 0) {
    const int H = c.pool_heads, PH = c.pool_hidden, dhp = PH / H, dop = D / H;
    if (!pool_fused) {
      GemmNT g; g.X = z; g.ldx = D; g.W = W.pw1_nk; g.ldw = D; g.M = T; g.N = PH; g.K = D;
      g.epi.bias = P + L.pb1; g.epi.act = 1; g.epi.save_pre = S.hp; g.epi.ldpre = PH; g.epi.out = S.ap; g.epi.ldc = PH;
      epi_drop(g.epi, mkdrop(train, c.pool_dropout, seed, 16u * 15 + SITE_POOL1), PH);
      RUN(launch_gemm_nt(g, st));
    }
    if (!pool_fused) {
      GemmNT g; g.X = S.ap; g.ldx = PH; g.W = W.pw2_nk; g.ldw = dhp; g.M = T; g.N = dop; g.K = dhp;
      g.groups = H; g.zX = dhp; g.zW = (long)dop * dhp; g.zOut = dop;
      g.epi.bias = P + L.pb2; g.epi.out = S.s; g.epi.ldc = D;
      epi_drop(g.epi, mkdrop(train, c.pool_dropout, seed, 16u * 15 + SITE_POOL2), D);
      RUN(launch_gemm_nt(g, st));
    }
    long row = 0; int n0 = 0;
    PoolArgs ps[2];
    for (int sidx = 0; sidx < sg.n; ++sidx) {
      PoolArgs& p = ps[sidx]; p.s = S.s + row * D; p.lds = D; p.z = z + row * D; p.ldz = D; p.lens = sg.lens[sidx]; p.N = sg.N[sidx]; p.L = sg.L[sidx];
      if (sg.cu) { p.s = S.s; p.z = z; p.cu = sg.cu + n0; }  // packed: global row offsets per sequence
      p.D = D; p.pooled = pooled + (size_t)n0 * out_dim; p.ldp = out_dim;
      p.pooled_copy = S.pooled + (size_t)n0 * D; p.smax = S.smax + (size_t)n0 * D; p.ssum = S.ssum + (size_t)n0 * D;
      p.drop_w = mkdrop(train, c.pool_dropout, seed + 977u * sidx, 16u * 15 + SITE_POOL3);
      row += (long)sg.N[sidx] * sg.L[sidx]; n0 += sg.N[sidx];
    }
    if (g_pool_handover.fwd && sg.n == 2 && !c.use_context) {  // the item segment's rows also go into the global network's padded layout
      PoolArgs& p = ps[1]; const PoolHandover& h = g_pool_handover;
      p.pk_counts = h.counts; p.pk_B = h.B; p.pk_Cmax = h.Cmax; p.pk_out = h.out; p.pk_mask = h.mask; p.pk_lens = h.lens;
      g_pool_handover.taken = true;
    }
    RUN(launch_pool_fwd2(ps, sg.n, st));  // both segments (e.g. videos and clips) in one launch
  } else {
    COOT_REQUIRE(sg.n == 1, "avg_special networks take a single segment");
    RUN(launch_avgpool_fwd(z, D, sg.lens[0], N, Lseq, D, pooled, out_dim, st));
  }
  (void)scratch; (void)scratch_bytes;
  return 0;
}

int coot_net_bwd(const coot_net_config* cfg, const float* P, const void* wpack, const float* pe, const float* feats,
                 const int64_t* lengths, int N, int Lseq, const float* feats2, const int64_t* lengths2, int N2, int L2,
                 const float* hidden, const float* dpooled, float* G, float* dhidden,
                 float* dfeats, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, int train, uint64_t seed,
                 const uint64_t* seed_dev, coot_stream_t stream, const coot_packed_seqs* packed) {
  hipStream_t st = (hipStream_t)stream;
  struct SeedScope { SeedScope(const uint64_t* p) { g_seed_dev = (const unsigned long long*)p; } ~SeedScope() { g_seed_dev = nullptr; } } seedscope(seed_dev);
  coot_net_config c; RUN(norm_cfg(cfg, &c));
  COOT_REQUIRE(P && wpack && feats && lengths && dpooled && G && saved && scratch, "net_bwd: null pointer");
  if (packed && packed->source != COOT_SOURCE_PADDED && N2 > 0 && !feats2) feats2 = feats;  // (one packed matrix carries both segments)
  COOT_REQUIRE(!(dfeats && c.use_input_fc), "net_bwd: dfeats is only available for networks without input_fc");
  if (N <= 0) return 0;
  Segs sg; sg.N[0] = N; sg.L[0] = Lseq; sg.lens[0] = (const long long*)lengths;
  if (N2 > 0) {
    COOT_REQUIRE(feats2 && lengths2 && !c.use_context && !dfeats, "net_bwd: second segment only for local networks");
    sg.n = 2; sg.N[1] = N2; sg.L[1] = L2; sg.lens[1] = (const long long*)lengths2;
  }
  const int Ntot = sg.Ntot();
  NetLayout L; build_layout(c, L);
  if (c.dtype == COOT_DTYPE_F32) {  // fp32 reference mode (ref_f32.hip): the derivative of the reference's op sequence, all fp32
    COOT_REQUIRE(!train, "net_bwd: the f32 reference mode is an eval-mode checker (train must be 0)");
    // ref_f32_backward ACCUMULATES every parameter gradient into G: under coot_net_grads_overwrite(1) (the step API's mode: only the
    // bias / LayerNorm vectors are zeroed, the weight-matrix gradients are expected to be WRITTEN) it would add each step's matrices
    // onto the previous step's — refuse instead of doing that silently
    COOT_REQUIRE(!g_grad_overwrite, "net_bwd: the f32 reference mode accumulates into zeroed gradient arenas (coot_net_grads_overwrite must be 0)");
    COOT_REQUIRE(!packed || packed->source == COOT_SOURCE_PADDED, "net_bwd: the f32 reference mode reads the reference's padded batch");
    COOT_REQUIRE(!c.use_context || hidden, "net_bwd: context network needs hidden state (transformer_legacy.py:252)");
    RefSegs rs; rs.n = sg.n;
    for (int s_ = 0; s_ < sg.n; ++s_) { rs.N[s_] = sg.N[s_]; rs.L[s_] = sg.L[s_]; rs.lens[s_] = sg.lens[s_]; }
    return ref_f32_backward(ref_desc(c, L, Ntot), P, G, feats, feats2, rs, hidden, dpooled, dhidden, dfeats, saved, saved_bytes, scratch, scratch_bytes, st);
  }
  COOT_REQUIRE(!COOT_OPERAND_IS_F16, "net_bwd: the f16 operand build is forward-only — the reference trains its fp16 path under a GradScaler "
               "(coot/trainer_retrieval.py:277-285, nntrainer/trainer_base.py:106-109); unscaled half gradients underflow and are not offered");
  Arena AW((void*)wpack, (size_t)-1); WPack W; layout_wpack(c, AW, W);
  PerOpGuardScope perop_guard(c, W, P, wpack);
  Arena AS(saved, saved_bytes); Saved S; layout_saved(c, Ntot, sg.Tpad(), AS, S);
  COOT_REQUIRE(!AS.overflow, "net_bwd: saved buffer too small");
  if (c.use_input_fc && g_input_stage.xhat) { S.xhat = g_input_stage.xhat; S.pos = g_input_stage.pos; }  // (the forward's input stage)
  Arena AX(scratch, scratch_bytes); Scratch X; layout_scratch(c, Ntot, sg.Tpad(), AX, X);
  if (packed_ok(c, W, sg, packed)) { sg.cu = packed->cu_seqlens; sg.Tp = packed->total_tokens; }  // the forward's decision
  COOT_REQUIRE(!AX.overflow, "net_bwd: scratch buffer too small (%zu < %zu)", scratch_bytes, AX.off);
  struct TnWs { TnWs(float* p, size_t n) { set_tn_default_workspace(p, n); } ~TnWs() { set_tn_default_workspace(nullptr, 0); } } tnws(X.tn_ws, X.tn_ws_floats);
  struct PartWs { PartWs(float* p, size_t n, size_t lo) { set_partials_workspace(p, n, lo); } ~PartWs() { set_partials_workspace(nullptr, 0); } } partws(X.part_ws, X.part_floats, X.part_low);
  // every weight gradient GEMM of the pass is recorded and launched as ONE batched kernel (gemm.h: tn_batch_*).  With a
  // single encoder (and context) layer no operand buffer is re-used before the end of the pass; deeper networks flush
  // after every layer, whose scratch buffers the next layer overwrites.
  struct TnBatchScope { TnBatchScope() { tn_batch_begin(); } ~TnBatchScope() { tn_batch_end(); } } tnbatch;
  // coot_net_grads_overwrite(1): every weight-matrix gradient is WRITTEN by this pass (each comes from exactly one problem)
  struct OverwriteScope { OverwriteScope(bool on) { set_tn_force_overwrite(on); } ~OverwriteScope() { set_tn_force_overwrite(false); } } ovw(g_grad_overwrite != 0);
  // the column-sum reductions of the fused kernels (bias / LayerNorm gradients) are recorded and run as one launch at the end
  struct DeferScope { DeferScope() { colsum_defer_begin(); } ~DeferScope() { colsum_defer_end(); } } deferscope;
  const bool flush_per_layer = c.num_layers > 1 || (c.use_context && c.ctx_num_layers > 1);
  const int D = c.hidden_dim, T = sg.T(), Din = c.input_dim;
  const long long* lens = sg.lens[0];
  const int out_dim = D * (c.use_context ? 2 : 1);
  const bf16_t* zL = S.layers[c.num_layers - 1].z2;
  bf16_t* dz = X.dzA;     // grad wrt the last layer's output tokens
  bf16_t* dz_other = X.dzB;

  if (glob_fused_ok(c, sg) && g_use_glob_fused_bwd && W.layers[0].f_wqkv_kn && W.ctx[0].f_wqkv_kn) {
    // the whole context network backward in ONE launch (fused.hip: glob_bwd_kernel) + the batched weight-gradient GEMM
    COOT_REQUIRE(dhidden && hidden, "net_bwd: dhidden required for context networks");
    GlobBwd f; f.B = N; f.Cmax = Lseq; f.x = feats; f.lens = lens; f.n_gain = P + L.n_gain; f.dpooled = dpooled; f.dx = dfeats; f.dhidden = dhidden;
    f.g_n_gain = G + L.n_gain; f.g_n_bias = G + L.n_bias; f.tstamps = g_fz_tstamps ? g_fz_tstamps + 16 : nullptr;
    struct Dy { bf16_t *dr2, *dr2m, *dh1, *dr1, *dq; long lddq; bf16_t* dk; long lddk; bf16_t* dv; long lddv; };
    auto fill = [&](GlobLayerBwd& g, const LayerP& lp, const LayerW& lw, const LayerBufs& b, const Dy& d, float pdrop, unsigned site_base) {
      g.wqkv_kn = lw.f_wqkv_kn; g.wo_kn = lw.f_wo_kn; g.w1_kn = lw.f_w1_kn; g.w2_kn = lw.f_w2_kn; g.ln1g = P + lp.ln1g; g.ln2g = P + lp.ln2g;
      g.q = b.q; g.ldq = b.ldq; g.k = b.k; g.ldk = b.ldk; g.v = b.v; g.ldv = b.ldv; g.r1 = b.r1; g.h1 = b.h1; g.r2 = b.r2; g.lse = b.lse;
      g.dr2 = d.dr2; g.dr2m = d.dr2m; g.dh1 = d.dh1; g.dr1 = d.dr1; g.dq = d.dq; g.lddq = d.lddq; g.dk = d.dk; g.lddk = d.lddk; g.dv = d.dv; g.lddv = d.lddv;
      g.g_ln2g = G + lp.ln2g; g.g_ln2b = G + lp.ln2b; g.g_ln1g = G + lp.ln1g; g.g_ln1b = G + lp.ln1b;
      g.d_attn = mkdrop(train, pdrop, seed, site_base + SITE_ATTN); g.d_postln = mkdrop(train, pdrop, seed, site_base + SITE_POSTLN);
      g.d_ff1 = mkdrop(train, pdrop, seed, site_base + SITE_FF1); g.d_ff2 = mkdrop(train, pdrop, seed, site_base + SITE_FF2);
    };
    const LayerBufs bs = self_bufs(S.layers[0], D), bc = ctx_bufs(S.ctx[0], D);
    const Dy ds{X.dr2, X.dr2m, X.dh1, X.dr1, X.dqkv, 3L * D, X.dqkv + D, 3L * D, X.dqkv + 2 * D, 3L * D};
    const Dy dc{X.c_d1, X.c_d2, X.c_dh1, X.c_dr1, X.c_dq, (long)D, X.c_dkv, 2L * D, X.c_dkv + D, 2L * D};
    fill(f.self, L.layers[0], W.layers[0], bs, ds, c.dropout, 0u);
    fill(f.ctx, L.ctx[0], W.ctx[0], bc, dc, c.ctx_dropout, 16u * 8);
    RUN(launch_glob_bwd(f, st));
    // the weight gradients layer_bwd records, from the dY tensors the kernel wrote (one batched launch at the flush below)
    auto tn = [&](const bf16_t* A, long lda, const bf16_t* Bm, long ldb, int Tr, int Mo, int No, float* Cw, float* a_colsum) {
      GemmTN t; t.A = A; t.lda = lda; t.B = Bm; t.ldb = ldb; t.T = Tr; t.Mo = Mo; t.No = No; t.C = Cw; t.ldc = No; t.a_colsum = a_colsum;
      return launch_gemm_tn(t, st);
    };
    const int F = c.ff_dim;
    {  // context layer: rows = sequences for the chain and the query projection, tokens for the key / value projections
      const LayerP& lp = L.ctx[0];
      // (b2, b1, bo gradients = column sums of the A operands: taken by the GEMM that streams them)
      RUN(tn(f.ctx.d_ff2.thr ? dc.dr2m : dc.dr2, D, bc.a1, F, N, D, F, G + lp.w2, G + lp.b2));
      RUN(tn(dc.dh1, F, bc.z1, D, N, F, D, G + lp.w1, G + lp.b1));
      RUN(tn(dc.dr1, D, bc.ctx, D, N, D, D, G + lp.wo, G + lp.bo));
      RUN(tn(dc.dq, D, S.cq_in, D, N, D, D, G + lp.wqkv, G + lp.bq));
      RUN(tn(dc.dk, 2 * D, zL, D, T, 2 * D, D, G + lp.wqkv + (size_t)D * D, G + lp.bk));
    }
    {
      const LayerP& lp = L.layers[0];
      RUN(tn(f.self.d_ff2.thr ? ds.dr2m : ds.dr2, D, bs.a1, F, T, D, F, G + lp.w2, G + lp.b2));
      RUN(tn(ds.dh1, F, bs.z1, D, T, F, D, G + lp.w1, G + lp.b1));
      RUN(tn(ds.dr1, D, bs.ctx, D, T, D, D, G + lp.wo, G + lp.bo));
      RUN(tn(ds.dq, 3 * D, S.z0, D, T, 3 * D, D, G + lp.wqkv, G + lp.bq));
    }
    // The weight gradients (and the deferred bias / LayerNorm column sums) of the global network are read by its UPDATE only, not by the
    // local backward that follows on this stream: a step that updates the global network early on another stream (api_step.hip:
    // early_update_fire) takes the batched launch over there (coot_internal_set_glob_flush_stream) — it left 18-33 us of kernel plus a
    // launch gap between the global backward and the first kernel of the local backward on each side's critical stream.
    hipStream_t fs = st;
    if (g_glob_flush_stream && g_glob_flush_stream != st && coot_option_glob_flush_aux()) {
      fs = g_glob_flush_stream;
      if (!g_glob_flush_ev) RUN(check_hip(hipEventCreateWithFlags(&g_glob_flush_ev, hipEventDisableTiming | hipEventDisableSystemFence), "hipEventCreate"));
      RUN(check_hip(hipEventRecord(g_glob_flush_ev, st), "eventRecord"));
      RUN(check_hip(hipStreamWaitEvent(fs, g_glob_flush_ev, 0), "streamWait"));
    }
    RUN(tn_batch_flush(fs));
    RUN(colsum_defer_flush(fs));
    return 0;
  }
  // pooling MLP dX + the last encoder layer's LN / FF / out-proj dX in one fused launch (fused.hip: pre_attn_bwd_kernel)
  const bool pool_bwd_fused = g_use_fused && fused_pool_ok(c) && W.f_pw1_kn && !c.use_context && T >= g_fused_min_rows;
  if (c.pooler == 0) {
    const int H = c.pool_heads, PH = c.pool_hidden, dhp = PH / H, dop = D / H;
    long row = 0; int n0 = 0;
    PoolArgs ps[2];
    for (int sidx = 0; sidx < sg.n; ++sidx) {
      PoolArgs& p = ps[sidx]; p.s = S.s + row * D; p.lds = D; p.z = zL + row * D; p.ldz = D; p.lens = sg.lens[sidx]; p.N = sg.N[sidx]; p.L = sg.L[sidx];
      p.D = D; p.pooled = S.pooled + (size_t)n0 * D; p.ldp = D; p.smax = S.smax + (size_t)n0 * D; p.ssum = S.ssum + (size_t)n0 * D;
      p.drop_w = mkdrop(train, c.pool_dropout, seed + 977u * sidx, 16u * 15 + SITE_POOL3);
      p.drop_s = mkdrop(train, c.pool_dropout, seed, 16u * 15 + SITE_POOL2); p.drop_s_ld = D; p.drop_s_row0 = row;
      p.dpooled = dpooled + (size_t)n0 * out_dim; p.lddp = out_dim; p.ds = X.ds + row * D; p.ldds = D; p.dz = X.dzp + row * D; p.lddz = D;
      if (sg.cu) { p.s = S.s; p.z = zL; p.ds = X.ds; p.dz = X.dzp; p.drop_s_row0 = 0; p.cu = sg.cu + n0; }  // packed: global row offsets
      p.ds_colsum = G + L.pb2;
      row += (long)sg.N[sidx] * sg.L[sidx]; n0 += sg.N[sidx];
    }
    RUN(launch_pool_bwd2(ps, sg.n, st));  // one launch + one bias-gradient reduction for both segments
    { GemmTN t; t.A = S.ap; t.lda = PH; t.B = X.ds; t.ldb = D; t.T = T; t.Mo = dhp; t.No = dop; t.C = G + L.pw2; t.ldc = dop;
      t.groups = H; t.zA = dhp; t.zB = dop; t.zC = (long)dhp * dop; RUN(launch_gemm_tn(t, st)); }
    if (!pool_bwd_fused) {  // dhp = (ds_h . W2[h]^T) * gelu'(hp) * drop1 ; db1p = colsum
      GemmNT g; g.X = X.ds; g.ldx = D; g.W = W.pw2_kn; g.ldw = dop; g.M = T; g.N = dhp; g.K = dop;
      g.groups = H; g.zX = dop; g.zW = (long)dhp * dop; g.zOut = dhp;
      g.epi.act = 2; g.epi.aux = S.hp; g.epi.ldaux = PH; g.epi.colsum = G + L.pb1; g.epi.out = X.dhp; g.epi.ldc = PH;
      epi_drop(g.epi, mkdrop(train, c.pool_dropout, seed, 16u * 15 + SITE_POOL1), PH);
      RUN(launch_gemm_nt(g, st));
    }
    { GemmTN t; t.A = zL; t.lda = D; t.B = X.dhp; t.ldb = PH; t.T = T; t.Mo = D; t.No = dhp; t.C = G + L.pw1; t.ldc = dhp;
      t.groups = H; t.zA = 0; t.zB = dhp; t.zC = (long)D * dhp; RUN(launch_gemm_tn(t, st)); }
    if (!pool_bwd_fused) {  // dz = dhp . W1p + dz(pool direct)
      GemmNT g; g.X = X.dhp; g.ldx = PH; g.W = W.pw1_kn; g.ldw = PH; g.M = T; g.N = D; g.K = PH;
      g.epi.res = X.dzp; g.epi.ldres = D; g.epi.out = dz; g.epi.ldc = D;
      RUN(launch_gemm_nt(g, st));
    }
  } else {
    COOT_REQUIRE(sg.n == 1, "avg_special networks take a single segment");
    RUN(launch_avgpool_bwd(dpooled, out_dim, lens, N, Lseq, D, dz, D, st));
  }

  if (c.use_context) {
    COOT_REQUIRE(dhidden, "net_bwd: dhidden required for context networks");
    for (int i = c.ctx_num_layers - 1; i >= 0; --i) {
      LayerBufs b = ctx_bufs(S.ctx[i], D);
      LayerBwdBufs w; w.dr2 = X.c_d1; w.dr2m = X.c_d2; w.dh1 = X.c_dh1; w.dz1 = X.c_dz1; w.dr1 = X.c_dr1; w.dctx = X.c_dctx;
      w.dq = X.c_dq; w.lddq = D; w.dk = X.c_dkv; w.lddk = 2 * D; w.dv = X.c_dkv + D; w.lddv = 2 * D; w.delta = X.c_delta;
      const bf16_t* qin = i == 0 ? S.cq_in : S.ctx[i - 1].z2;
      const bool last = (i == c.ctx_num_layers - 1);
      const bool first = (i == 0);
      RUN(layer_bwd(c, P, G, L.ctx[i], W.ctx[i], qin, N, zL, T, sg, b, w, last ? nullptr : X.c_dqin,
                    last ? dpooled + D : nullptr, out_dim, first ? nullptr : X.c_dqin, first ? dhidden : nullptr, dz, nullptr, nullptr,
                    c.ctx_dropout, train, seed, 16u * (8 + i), st));
      if (flush_per_layer) { RUN(tn_batch_flush(st)); RUN(colsum_defer_flush(st)); }
    }
  }

  for (int i = c.num_layers - 1; i >= 0; --i) {
    LayerBufs b = self_bufs(S.layers[i], D);
    LayerBwdBufs w; w.dr2 = X.dr2; w.dr2m = X.dr2m; w.dh1 = X.dh1; w.dz1 = X.dz1; w.dr1 = X.dr1; w.dctx = X.dctx;
    w.dq = X.dqkv; w.lddq = 3 * D; w.dk = X.dqkv + D; w.lddk = 3 * D; w.dv = X.dqkv + 2 * D; w.lddv = 3 * D; w.delta = X.delta;
    const bf16_t* zin = i == 0 ? S.z0 : S.layers[i - 1].z2;
    const bool fc0 = (i == 0 && c.use_input_fc);
    if (fc0 && !qkv_bwd_is_fused(W.layers[i], T, false, X.part_ws)) RUN(launch_fill_f32(X.cvec, D, 0.f, st));  // the fused QKV dX writes it
    PoolFuseBwd pfb;
    const bool lastl = (i == c.num_layers - 1);
    if (lastl && pool_bwd_fused) {
      pfb.ds = X.ds; pfb.dzp = X.dzp; pfb.hp = S.hp; pfb.pw2 = W.f_pw2_kn; pfb.pw1 = W.f_pw1_kn; pfb.dhp = X.dhp; pfb.g_pb1 = G + L.pb1;
      pfb.d1 = mkdrop(train, c.pool_dropout, seed, 16u * 15 + SITE_POOL1);
    }
    RUN(layer_bwd(c, P, G, L.layers[i], W.layers[i], zin, T, zin, T, sg, b, w, dz, nullptr, 0, dz_other, nullptr, nullptr,
                  fc0 ? S.h0 : nullptr, fc0 ? X.cvec : nullptr, c.dropout, train, seed, 16u * i, st,
                  (lastl && pool_bwd_fused) ? &pfb : nullptr, X.part_ws));
    if (flush_per_layer) { RUN(tn_batch_flush(st)); RUN(colsum_defer_flush(st)); }
    bf16_t* t = dz; dz = dz_other; dz_other = t;
  }
  // dz now holds: dh0 (input-FC nets: already multiplied by gelu'(h0)) or dz0 (grad wrt LN(x)+pe)
  if (c.use_input_fc) {
    { GemmTN t; t.A = dz; t.lda = D; t.B = S.xhat; t.ldb = Din; t.T = T; t.Mo = D; t.No = Din; t.C = X.Mbuf; t.ldc = Din;
      t.overwrite = 1;  // M = dh0^T . xhat: written, not accumulated (was a zero-fill launch in front of the batch)
      RUN(launch_gemm_tn(t, st)); }
    RUN(tn_batch_flush(st, true));  // all weight gradients of the pass (the parameter-gradient kernel below reads Mbuf) and, in the
                                    // same reduce launch, every deferred column sum (it reads cvec)
    RUN(launch_infc_param_grads(X.Mbuf, P + L.in_w, P + L.n_gain, P + L.n_bias, X.cvec, D, Din, G + L.in_w, G + L.n_gain, G + L.n_bias,
                                G + L.in_b /* db_in += colsum(dh0) */, st, g_grad_overwrite));
  } else {
    LnBwd l; l.dy = dz; l.lddy = D; l.x = feats; l.x_f32 = 1; l.ldx = Din; l.gain = P + L.n_gain; l.R = T; l.D = D;
    l.dx32 = dfeats; l.lddx32 = Din; l.dgain = G + L.n_gain; l.dbias = G + L.n_bias;
    if (!dfeats) { l.dx = dz_other; l.lddx = D; }
    RUN(launch_ln_bwd(l, st));
    RUN(tn_batch_flush(st, true));  // (+ the deferred column sums, in its reduce launch)
  }
  (void)pe; (void)hidden;
  return 0;
}

}  // extern "C"
