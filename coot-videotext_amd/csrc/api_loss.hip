// C-ABI, part 2: losses, packing, kernel-level entry points and hardware probes.
#include <string.h>

#include "../../include/coot_hip.h"
#include "attention.h"
#include "common.h"
#include "gemm.h"
#include "loss.h"
#include "pool.h"
#include "rowops.h"

using namespace coot;

#define RUN(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

namespace {
struct Bump {
  char* base; size_t cap; size_t off = 0; bool overflow = false;
  Bump(void* b, size_t c) : base((char*)b), cap(c) {}
  template <typename T> T* get(size_t n) {
    off = (off + 255) & ~(size_t)255;
    char* p = base ? base + off : nullptr;
    off += n * sizeof(T);
    if (base && off > cap) overflow = true;
    return (T*)p;
  }
};
}  // namespace

extern "C" {

int coot_pack_fwd(const float* emb, const int64_t* counts, int B, int Cmax, int D, float* out, uint8_t* mask, int64_t* lens,
                  coot_stream_t stream) {
  return launch_pack_fwd(emb, (const long long*)counts, B, Cmax, D, out, mask, (long long*)lens, (hipStream_t)stream);
}
int coot_pack_bwd(const float* dout, const int64_t* counts, int B, int Cmax, int D, float* demb, coot_stream_t stream) {
  return launch_pack_bwd(dout, (const long long*)counts, B, Cmax, D, demb, (hipStream_t)stream);
}

size_t coot_contrastive_scratch_bytes(int n_high, int n_low, int d_high, int d_low) {
  return contrastive_fused_scratch_bytes(n_high, n_low, d_high, d_low);
}

int coot_contrastive_fwd_bwd(const coot_contrastive_config* cfg, int n_high, int n_low, int d_high, int d_low, const float* vid_emb,
                             const float* par_emb, const float* clip_emb, const float* sent_emb, const float* vid_ctx,
                             const float* par_ctx, float* loss, float* d_vid_emb, float* d_par_emb, float* d_clip_emb,
                             float* d_sent_emb, float* d_vid_ctx, float* d_par_ctx, void* scratch, size_t scratch_bytes,
                             coot_stream_t stream) {
  return coot_contrastive_fwd_bwd_part(cfg, n_high, n_low, d_high, d_low, vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx, loss, d_vid_emb,
                                       d_par_emb, d_clip_emb, d_sent_emb, d_vid_ctx, d_par_ctx, scratch, scratch_bytes,
                                       COOT_CONTRASTIVE_GLOBAL | COOT_CONTRASTIVE_LOCAL, stream);
}

int coot_contrastive_fwd_bwd_part(const coot_contrastive_config* cfg, int n_high, int n_low, int d_high, int d_low, const float* vid_emb,
                                  const float* par_emb, const float* clip_emb, const float* sent_emb, const float* vid_ctx,
                                  const float* par_ctx, float* loss, float* d_vid_emb, float* d_par_emb, float* d_clip_emb,
                                  float* d_sent_emb, float* d_vid_ctx, float* d_par_ctx, void* scratch, size_t scratch_bytes, int part,
                                  coot_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  COOT_REQUIRE((part & ~3) == 0 && part != 0, "contrastive: part = COOT_CONTRASTIVE_GLOBAL | COOT_CONTRASTIVE_LOCAL");
  COOT_REQUIRE(cfg && vid_emb && par_emb && clip_emb && sent_emb && vid_ctx && par_ctx && loss && scratch, "contrastive: null pointer");
  const bool bwd = d_vid_emb != nullptr;
  COOT_REQUIRE(!bwd || (d_par_emb && d_clip_emb && d_sent_emb && d_vid_ctx && d_par_ctx), "contrastive: gradient pointers must be all set or all null");
  const float* vs[6] = {vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx};
  float* dvs[6] = {d_vid_emb, d_par_emb, d_clip_emb, d_sent_emb, d_vid_ctx, d_par_ctx};
  // coot/trainer_retrieval.py:168-182 (note :181 weights the context cluster term with weight_low_internal)
  const float w_pair[3] = {cfg->weight_high, cfg->weight_low, cfg->weight_context};
  const float w_self[3] = {0.5f * cfg->weight_high_internal, 0.5f * cfg->weight_low_internal,
                           cfg->weight_context_internal != 0.f ? 0.5f * cfg->weight_low_internal : 0.f};
  const int pair_mask = ((part & COOT_CONTRASTIVE_GLOBAL) ? 1 : 0) | ((part & COOT_CONTRASTIVE_LOCAL) ? 6 : 0);
  return launch_contrastive_fused(vs, dvs, n_high, n_low, d_high, d_low, w_pair, w_self, cfg->margin, loss, scratch, scratch_bytes, st, nullptr,
                                  nullptr, pair_mask);
}

size_t coot_contrastive_f32_scratch_bytes(int n_high, int n_low, int d_high, int d_low) {
  return contrastive_f32_scratch_bytes(n_high, n_low, d_high, d_low);
}

int coot_contrastive_fwd_bwd_f32(const coot_contrastive_config* cfg, int n_high, int n_low, int d_high, int d_low, const float* vid_emb,
                                 const float* par_emb, const float* clip_emb, const float* sent_emb, const float* vid_ctx,
                                 const float* par_ctx, float* loss, float* d_vid_emb, float* d_par_emb, float* d_clip_emb,
                                 float* d_sent_emb, float* d_vid_ctx, float* d_par_ctx, void* scratch, size_t scratch_bytes,
                                 coot_stream_t stream) {
  COOT_REQUIRE(cfg && vid_emb && par_emb && clip_emb && sent_emb && vid_ctx && par_ctx && loss && scratch, "contrastive_f32: null pointer");
  const bool bwd = d_vid_emb != nullptr;
  COOT_REQUIRE(!bwd || (d_par_emb && d_clip_emb && d_sent_emb && d_vid_ctx && d_par_ctx), "contrastive_f32: gradient pointers must be all set or all null");
  const float* vs[6] = {vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx};
  float* dvs[6] = {d_vid_emb, d_par_emb, d_clip_emb, d_sent_emb, d_vid_ctx, d_par_ctx};
  const float w_pair[3] = {cfg->weight_high, cfg->weight_low, cfg->weight_context};
  const float w_self[3] = {0.5f * cfg->weight_high_internal, 0.5f * cfg->weight_low_internal,
                           cfg->weight_context_internal != 0.f ? 0.5f * cfg->weight_low_internal : 0.f};  // (:181, as the fast path)
  return launch_contrastive_f32(vs, dvs, n_high, n_low, d_high, d_low, w_pair, w_self, cfg->margin, loss, scratch, scratch_bytes, (hipStream_t)stream);
}

int coot_contrastive_fwd_bwd_dp(const coot_contrastive_config* cfg, int n_high, int n_low, int d_high, int d_low, const float* const sets[6],
                                const int64_t ld[6], float* loss, float* const d_own[6], int own_high0, int own_high, int own_low0, int own_low,
                                void* scratch, size_t scratch_bytes, coot_stream_t stream) {
  COOT_REQUIRE(cfg && sets && ld && loss && d_own && scratch, "contrastive_dp: null pointer");
  COOT_REQUIRE(own_high0 >= 0 && own_high0 + own_high <= n_high && own_low0 >= 0 && own_low0 + own_low <= n_low, "contrastive_dp: window outside the batch");
  const float w_pair[3] = {cfg->weight_high, cfg->weight_low, cfg->weight_context};
  const float w_self[3] = {0.5f * cfg->weight_high_internal, 0.5f * cfg->weight_low_internal,
                           cfg->weight_context_internal != 0.f ? 0.5f * cfg->weight_low_internal : 0.f};
  long ldv[6];
  for (int i = 0; i < 6; ++i) { COOT_REQUIRE(sets[i] && d_own[i] && ld[i] % 4 == 0, "contrastive_dp: set %d", i); ldv[i] = (long)ld[i]; }
  const int window[4] = {own_high0, own_high, own_low0, own_low};
  return launch_contrastive_fused(sets, d_own, n_high, n_low, d_high, d_low, w_pair, w_self, cfg->margin, loss, scratch, scratch_bytes,
                                  (hipStream_t)stream, ldv, window);
}

int coot_contrastive_fwd_bwd_dp_blocks(const coot_contrastive_config* cfg, int world, int rank, const int64_t* counts_high,
                                       const int64_t* counts_low, int d_high, int d_low, const float* blocks, const int64_t* set_base,
                                       const int64_t ld[6], float* loss, float* const d_own[6], void* scratch, size_t scratch_bytes,
                                       coot_stream_t stream) {
  COOT_REQUIRE(cfg && counts_high && counts_low && blocks && set_base && ld && loss && d_own && scratch, "contrastive_dp_blocks: null pointer");
  COOT_REQUIRE(world >= 1 && world <= CL_MAX_RANKS && rank >= 0 && rank < world, "contrastive_dp_blocks: %d ranks (1 .. %d), rank %d", world,
               CL_MAX_RANKS, rank);
  const float w_pair[3] = {cfg->weight_high, cfg->weight_low, cfg->weight_context};
  const float w_self[3] = {0.5f * cfg->weight_high_internal, 0.5f * cfg->weight_low_internal,
                           cfg->weight_context_internal != 0.f ? 0.5f * cfg->weight_low_internal : 0.f};
  ClBlocks B; B.blocks = blocks; B.world = world;
  long n[2] = {0, 0};
  for (int r = 0; r < world; ++r) {
    COOT_REQUIRE(counts_high[r] >= 0 && counts_low[r] >= 0, "contrastive_dp_blocks: negative row count of rank %d", r);
    B.row0[0][r] = (int)n[0]; B.row0[1][r] = (int)n[1];
    n[0] += counts_high[r]; n[1] += counts_low[r];
    for (int s = 0; s < 6; ++s) { COOT_REQUIRE(set_base[s * world + r] % 4 == 0, "contrastive_dp_blocks: set %d of rank %d is not 16-byte aligned", s, r); B.base[s][r] = (long)set_base[s * world + r]; }
  }
  COOT_REQUIRE(n[0] < (1l << 30) && n[1] < (1l << 30), "contrastive_dp_blocks: batch too large");
  B.row0[0][world] = (int)n[0]; B.row0[1][world] = (int)n[1];
  long ldv[6];
  const float* vs[6];
  const int window[4] = {B.row0[0][rank], (int)counts_high[rank], B.row0[1][rank], (int)counts_low[rank]};
  for (int s = 0; s < 6; ++s) {
    COOT_REQUIRE(d_own[s] && ld[s] % 4 == 0, "contrastive_dp_blocks: set %d", s);
    ldv[s] = (long)ld[s];
    // the window's rows (the only ones read through v[s]: F.normalize backward of this rank's rows) live in this rank's block
    vs[s] = blocks + B.base[s][rank] - (long)window[(s == 2 || s == 3) ? 2 : 0] * ldv[s];
  }
  return launch_contrastive_fused(vs, d_own, (int)n[0], (int)n[1], d_high, d_low, w_pair, w_self, cfg->margin, loss, scratch, scratch_bytes,
                                  (hipStream_t)stream, ldv, window, 7, &B);
}

int coot_cyclecons_fwd_bwd(const float* clip, const float* sent, const int64_t* clip_lens, const int64_t* sent_lens,
                           const int64_t* idx_clip, const int64_t* idx_sent, int B, int Cc, int Cs, int D, float weight,
                           float inv_batch, float* loss, float* rows_clip, float* rows_sent, float* dclip, float* dsent,
                           coot_stream_t stream) {
  CycleArgs a; a.clip = clip; a.sent = sent; a.clip_lens = (const long long*)clip_lens; a.sent_lens = (const long long*)sent_lens;
  a.idx_clip = (const long long*)idx_clip; a.idx_sent = (const long long*)idx_sent; a.B = B; a.Cc = Cc; a.Cs = Cs; a.D = D;
  a.weight = weight; a.inv_batch = inv_batch; a.loss = loss; a.rows_clip = rows_clip; a.rows_sent = rows_sent; a.dclip = dclip; a.dsent = dsent;
  return launch_cyclecons(a, (hipStream_t)stream);
}

// ---- kernel-level entry points -----------------------------------------------------------------------
int coot_gemm_nt(const void* X, int64_t ldx, const void* W, int64_t ldw, int M, int N, int K, const float* bias, int act,
                 const void* residual_bf16, int64_t ldres, void* out, int64_t ldc, int out_f32, coot_stream_t stream) {
  GemmNT g; g.X = (const bf16_t*)X; g.ldx = ldx; g.W = (const bf16_t*)W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
  g.epi.bias = bias; g.epi.act = act; g.epi.res = (const bf16_t*)residual_bf16; g.epi.ldres = ldres; g.epi.out = out; g.epi.ldc = ldc;
  g.epi.out_f32 = out_f32;
  return launch_gemm_nt(g, (hipStream_t)stream);
}
size_t coot_gemm_tn_workspace_bytes(int T, int Mo, int No) { return gemm_tn_workspace_floats(T, Mo, No, 1) * sizeof(float); }
int coot_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int T, int Mo, int No, float* C, int64_t ldc,
                 void* workspace, size_t workspace_bytes, coot_stream_t stream) {
  GemmTN t; t.A = (const bf16_t*)A; t.lda = lda; t.B = (const bf16_t*)B; t.ldb = ldb; t.T = T; t.Mo = Mo; t.No = No; t.C = C; t.ldc = ldc;
  t.ws = (float*)workspace; t.ws_floats = workspace_bytes / sizeof(float);
  return launch_gemm_tn(t, (hipStream_t)stream);
}
int coot_gemm_tn_batch(const coot_tn_problem* p, int n, void* workspace, size_t workspace_bytes, uint64_t* stamps, coot_stream_t stream) {
  if (n < 0 || (n > 0 && !p)) { set_error("gemm_tn_batch: bad arguments"); return -1; }
  float* old_ws; size_t old_n;
  get_tn_default_workspace(&old_ws, &old_n);
  set_tn_default_workspace((float*)workspace, workspace_bytes / sizeof(float));
  tn_batch_begin();
  int rc = 0;
  for (int i = 0; i < n && !rc; ++i) {
    GemmTN t; t.A = (const bf16_t*)p[i].A; t.lda = p[i].lda; t.B = (const bf16_t*)p[i].B; t.ldb = p[i].ldb; t.T = p[i].T; t.Mo = p[i].Mo; t.No = p[i].No;
    t.C = p[i].C; t.ldc = p[i].ldc; t.a_colsum = p[i].a_colsum; t.overwrite = p[i].overwrite;
    t.groups = p[i].groups > 0 ? p[i].groups : 1; t.zA = p[i].zA; t.zB = p[i].zB; t.zC = p[i].zC;
    t.stamps = (unsigned long long*)stamps;  // phase stamps: tile (0, 0, 0) of EVERY problem writes them (same slots: use one problem, or read them as "some problem")
    rc = launch_gemm_tn(t, (hipStream_t)stream);
  }
  if (!rc) rc = tn_batch_flush((hipStream_t)stream);
  tn_batch_end();
  set_tn_default_workspace(old_ws, old_n);
  return rc;
}
int coot_debug_tn_xcd_map(int n, const int* gx, const int* gy, const int* groups, const int* splits, int* out_item, int* out_local, int max_blocks) {
  return tn_debug_xcd_map(n, gx, gy, groups, splits, out_item, out_local, max_blocks);
}
int coot_ln_fwd(const float* x, int R, int D, const float* gain, const float* bias, void* y_bf16, float* y_f32, coot_stream_t stream) {
  LnFwd l; l.x = x; l.x_f32 = 1; l.ldx = D; l.R = R; l.D = D; l.gain = gain; l.bias = bias; l.y = (bf16_t*)y_bf16; l.ldy = D; l.y32 = y_f32; l.ldy32 = D;
  return launch_ln_fwd(l, (hipStream_t)stream);
}
int coot_attn_fwd(const void* qkv, int Nseq, int L, int H, int dh, const int64_t* lens, void* out, float* lse, coot_stream_t stream) {
  const int D = H * dh;
  AttnArgs a; a.q = (const bf16_t*)qkv; a.k = a.q + D; a.v = a.q + 2 * D; a.ldq = a.ldk = a.ldv = 3 * D; a.o = (bf16_t*)out; a.ldo = D;
  a.lse = lse; a.lens = (const long long*)lens; a.Nseq = Nseq; a.Lq = L; a.Lk = L; a.H = H; a.dh = dh; a.scale = 1.0f / sqrtf((float)dh);
  return launch_attn_fwd(a, (hipStream_t)stream);
}

}  // extern "C"

// Clock monitor: ONE wave samples (100 MHz real-time counter, shader clock counter) every `interval_ticks` ticks of the former,
// n samples: the shader clock actually delivered while other streams run (power management lowers it under matrix load).
__global__ void clock_monitor_kernel(unsigned long long* out, int n, int interval_ticks) {
  if (threadIdx.x != 0) return;
  unsigned long long next = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    unsigned long long rt;
    do { __builtin_amdgcn_s_sleep(8); rt = __builtin_amdgcn_s_memrealtime(); } while (rt < next);
    out[2 * i] = rt;
    out[2 * i + 1] = __builtin_amdgcn_s_memtime();
    next = rt + interval_ticks;
  }
}
extern "C" int coot_debug_clock_monitor(uint64_t* out, int n, int interval_ticks, coot_stream_t stream) {
  hipLaunchKernelGGL(clock_monitor_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, n, interval_ticks);
  COOT_CHECK_LAUNCH("clock_monitor");
  return 0;
}

extern "C" int coot_timing_enable(int on) { gemm_timing_enable(on); return 0; }
extern "C" int coot_timing_collect(int only_big_k, double* ms, double* flops, int* launches) {
  int rc = gemm_timing_collect(only_big_k, ms, flops, launches);
  if (rc) set_error("timing_collect: event query failed");
  return rc;
}

// probe: LDS holds tile[r][c] = r*64 + c (16 rows x 64 cols bf16 bit patterns as raw u16);
// lane p of each 16-lane group g supplies the address of row (4*g + (p>>2)), cols (p&3)*4..+3.
__global__ void probe_tr16_kernel(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) tile[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, p = lane & 15;
  const uint16_t* addr = &tile[(4 * g + (p >> 2)) * 64 + (p & 3) * 4];
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(addr));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
extern "C" int coot_probe_tr16(uint16_t* out, coot_stream_t stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  COOT_CHECK_LAUNCH("probe_tr16");
  return 0;
}
