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
inline int pad8(int n) { return (n + 7) & ~7; }

struct EmbSet {  // one normalised embedding set
  const float* v; float* dv; int N, d;
  bf16_t *a, *aT; float *inv, *da;
};
struct LossScratch {
  EmbSet e[6];
  float* S; bf16_t *G, *GT; float* gd;
};
// sets: 0 vid_emb 1 par_emb (n_high, d_high) | 2 clip_emb 3 sent_emb (n_low, d_low) | 4 vid_ctx 5 par_ctx (n_high, d_low)
void layout_loss(int n_high, int n_low, int d_high, int d_low, Bump& A, LossScratch& L) {
  const int Ns[6] = {n_high, n_high, n_low, n_low, n_high, n_high};
  const int ds[6] = {d_high, d_high, d_low, d_low, d_low, d_low};
  for (int i = 0; i < 6; ++i) {
    EmbSet& e = L.e[i]; e.N = Ns[i]; e.d = ds[i];
    e.a = A.get<bf16_t>((size_t)pad8(e.N) * e.d); e.aT = A.get<bf16_t>((size_t)e.d * pad8(e.N));
    e.inv = A.get<float>(e.N); e.da = A.get<float>((size_t)e.N * e.d);
  }
  const int nmax = n_high > n_low ? n_high : n_low, np = pad8(nmax);
  L.S = A.get<float>((size_t)nmax * np); L.G = A.get<bf16_t>((size_t)nmax * np); L.GT = A.get<bf16_t>((size_t)nmax * np);
  L.gd = A.get<float>(nmax);
}

// one ContrastiveLoss term w * L(A, B): loss, dA, dB
int contrastive_term(LossScratch& L, int ia, int ib, float w, float margin, float* loss, bool bwd, hipStream_t st) {
  EmbSet& A = L.e[ia]; EmbSet& B = L.e[ib];
  const int N = A.N, d = A.d, np = pad8(N);
  {
    // S [N, np]: the normalised sets are stored with pad8(N) zero rows, so columns >= N come out 0
    GemmNT g; g.X = A.a; g.ldx = d; g.W = B.a; g.ldw = d; g.M = N; g.N = np; g.K = d;
    g.epi.out = L.S; g.epi.ldc = np; g.epi.out_f32 = 1;
    RUN(launch_gemm_nt(g, st));
  }
  RUN(check_hip(hipMemsetAsync(L.G, 0, (size_t)N * np * sizeof(bf16_t), st), "memset G"));
  RUN(check_hip(hipMemsetAsync(L.GT, 0, (size_t)N * np * sizeof(bf16_t), st), "memset GT"));
  RUN(check_hip(hipMemsetAsync(L.gd, 0, (size_t)N * sizeof(float), st), "memset gd"));
  RUN(launch_hinge(L.S, np, N, margin, w, loss, L.G, L.GT, np, L.gd, st));
  if (!bwd) return 0;
  const float alpha = w / ((float)N * (float)N);
  {  // dA += alpha * G . b + gd * b
    GemmNT g; g.X = L.G; g.ldx = np; g.W = B.aT; g.ldw = np; g.M = N; g.N = d; g.K = np;
    g.epi.alpha = alpha; g.epi.rowscale = L.gd; g.epi.diag_src = B.a; g.epi.lddiag = d;
    g.epi.out = A.da; g.epi.ldc = d; g.epi.out_f32 = 1; g.epi.accumulate = 1;
    RUN(launch_gemm_nt(g, st));
  }
  {  // dB += alpha * G^T . a + gd * a
    GemmNT g; g.X = L.GT; g.ldx = np; g.W = A.aT; g.ldw = np; g.M = N; g.N = d; g.K = np;
    g.epi.alpha = alpha; g.epi.rowscale = L.gd; g.epi.diag_src = A.a; g.epi.lddiag = d;
    g.epi.out = B.da; g.epi.ldc = d; g.epi.out_f32 = 1; g.epi.accumulate = 1;
    RUN(launch_gemm_nt(g, st));
  }
  return 0;
}
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
  Bump A(nullptr, 0); LossScratch L; layout_loss(n_high, n_low, d_high, d_low, A, L); return A.off + 256;
}

int coot_contrastive_fwd_bwd(const coot_contrastive_config* cfg, int n_high, int n_low, int d_high, int d_low, const float* vid_emb,
                             const float* par_emb, const float* clip_emb, const float* sent_emb, const float* vid_ctx,
                             const float* par_ctx, float* loss, float* d_vid_emb, float* d_par_emb, float* d_clip_emb,
                             float* d_sent_emb, float* d_vid_ctx, float* d_par_ctx, void* scratch, size_t scratch_bytes,
                             coot_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  COOT_REQUIRE(cfg && vid_emb && par_emb && clip_emb && sent_emb && vid_ctx && par_ctx && loss && scratch, "contrastive: null pointer");
  COOT_REQUIRE(d_high % 8 == 0 && d_low % 8 == 0, "contrastive: embedding dims must be multiples of 8");
  const bool bwd = d_vid_emb != nullptr;
  COOT_REQUIRE(!bwd || (d_par_emb && d_clip_emb && d_sent_emb && d_vid_ctx && d_par_ctx), "contrastive: gradient pointers must be all set or all null");
  Bump A(scratch, scratch_bytes); LossScratch L; layout_loss(n_high, n_low, d_high, d_low, A, L);
  COOT_REQUIRE(!A.overflow, "contrastive: scratch too small (%zu < %zu)", scratch_bytes, A.off);
  const float* vs[6] = {vid_emb, par_emb, clip_emb, sent_emb, vid_ctx, par_ctx};
  float* dvs[6] = {d_vid_emb, d_par_emb, d_clip_emb, d_sent_emb, d_vid_ctx, d_par_ctx};
  for (int i = 0; i < 6; ++i) {
    EmbSet& e = L.e[i]; e.v = vs[i]; e.dv = dvs[i];
    RUN(check_hip(hipMemsetAsync(e.aT, 0, (size_t)e.d * pad8(e.N) * sizeof(bf16_t), st), "memset aT"));
    RUN(check_hip(hipMemsetAsync(e.a, 0, (size_t)pad8(e.N) * e.d * sizeof(bf16_t), st), "memset a"));
    RUN(launch_l2norm_fwd(e.v, e.d, e.N, e.d, e.a, e.d, e.aT, pad8(e.N), e.inv, st));
    if (bwd) RUN(check_hip(hipMemsetAsync(e.da, 0, (size_t)e.N * e.d * sizeof(float), st), "memset da"));
  }
  // coot/trainer_retrieval.py:168-182 (note :181 weights the context cluster term with weight_low_internal)
  if (cfg->weight_high != 0.f) RUN(contrastive_term(L, 0, 1, cfg->weight_high, cfg->margin, loss, bwd, st));
  if (cfg->weight_low != 0.f) RUN(contrastive_term(L, 2, 3, cfg->weight_low, cfg->margin, loss, bwd, st));
  if (cfg->weight_context != 0.f) RUN(contrastive_term(L, 4, 5, cfg->weight_context, cfg->margin, loss, bwd, st));
  if (cfg->weight_high_internal != 0.f) {
    RUN(contrastive_term(L, 0, 0, 0.5f * cfg->weight_high_internal, cfg->margin, loss, bwd, st));
    RUN(contrastive_term(L, 1, 1, 0.5f * cfg->weight_high_internal, cfg->margin, loss, bwd, st));
  }
  if (cfg->weight_low_internal != 0.f) {
    RUN(contrastive_term(L, 2, 2, 0.5f * cfg->weight_low_internal, cfg->margin, loss, bwd, st));
    RUN(contrastive_term(L, 3, 3, 0.5f * cfg->weight_low_internal, cfg->margin, loss, bwd, st));
  }
  if (cfg->weight_context_internal != 0.f) {
    RUN(contrastive_term(L, 4, 4, 0.5f * cfg->weight_low_internal, cfg->margin, loss, bwd, st));
    RUN(contrastive_term(L, 5, 5, 0.5f * cfg->weight_low_internal, cfg->margin, loss, bwd, st));
  }
  if (bwd)
    for (int i = 0; i < 6; ++i) {
      EmbSet& e = L.e[i];
      RUN(launch_l2norm_bwd(e.da, e.d, e.v, e.d, e.inv, e.N, e.d, e.dv, e.d, 1, st));
    }
  return 0;
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
