// C-ABI, part 3: the whole retrieval training step as native code (coot/trainer_retrieval.py:253-291):
// forward of both sides on two HIP streams, losses, backward, fused Adam — one C call per step (or four phase
// calls when torch.distributed collectives sit between the phases).  No Python, no autograd bookkeeping and no
// allocator traffic between the ~350 kernel launches of a step: the step was host-bound (9.5 us per launch from
// Python, hipGraph replay no better on ROCm 7: ~8 us per node) — from C a launch costs ~3.5 us.
#include <string.h>

#include "../../include/coot_hip.h"
#include "common.h"
#include "det.h"
#include "gemm.h"
#include "fused.h"
#include "pool.h"
#include "rowops.h"

#include <vector>
using namespace coot;

#define RUN(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

namespace {

struct Bump {
  char* base; size_t cap; size_t off = 0; bool overflow = false;
  Bump(void* b, size_t c) : base((char*)b), cap(c) {}
  void* get(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    char* p = base ? base + off : nullptr;
    off += bytes;
    if (base && off > cap) overflow = true;
    return p;
  }
  template <typename T> T* arr(size_t n) { return (T*)get(n * sizeof(T)); }
};

struct StepWs {
  void *saved_lv, *saved_gv, *saved_lt, *saved_gt; size_t sz_lv, sz_gv, sz_lt, sz_gt;
  void *scratch_v, *scratch_t; size_t sz_sv, sz_st;
  void* loss_scratch; size_t sz_loss;
  // embeddings (this rank) and their gradients — used by the single-call step
  float *local_v, *local_t, *glob_v, *glob_t, *resh_v, *resh_t;
  float *d_local_v, *d_local_t, *d_glob_v, *d_glob_t, *d_resh_v, *d_resh_t;
  float *dhid_v, *dhid_t, *dfeat_v, *dfeat_t;
  unsigned char *mask_v, *mask_t; long long *lens_v, *lens_t, *idx;
  float* zero_begin; size_t zero_bytes;  // gradient buffers that must start at 0, contiguous
};

inline size_t max2(size_t a, size_t b) { return a > b ? a : b; }
inline size_t align256(size_t a) { return (a + 255) & ~(size_t)255; }

void layout_step(const coot_step_config& c, const coot_step_dims& d, Bump& A, StepWs& W) {
  const size_t D = c.net[0].hidden_dim;
  W.sz_lv = coot_net_saved_bytes(&c.net[0], d.B, d.Lv, d.Nc, d.Lc);
  W.sz_gv = coot_net_saved_bytes(&c.net[1], d.B, d.Cmax_clip, 0, 0);
  W.sz_lt = coot_net_saved_bytes(&c.net[2], d.B, d.Lp, d.Nc, d.Ls);
  W.sz_gt = coot_net_saved_bytes(&c.net[3], d.B, d.Cmax_sent, 0, 0);
  W.saved_lv = A.get(W.sz_lv); W.saved_gv = A.get(W.sz_gv); W.saved_lt = A.get(W.sz_lt); W.saved_gt = A.get(W.sz_gt);
  // backward: the global network's scratch sits BEHIND the local network's (side_backward) — its weight-gradient GEMMs run on an
  // aux stream next to the local backward and still read their operands there
  W.sz_sv = align256(coot_net_scratch_bytes(&c.net[0], d.B, d.Lv, d.Nc, d.Lc)) + coot_net_scratch_bytes(&c.net[1], d.B, d.Cmax_clip, 0, 0);
  W.sz_st = align256(coot_net_scratch_bytes(&c.net[2], d.B, d.Lp, d.Nc, d.Ls)) + coot_net_scratch_bytes(&c.net[3], d.B, d.Cmax_sent, 0, 0);
  W.scratch_v = A.get(W.sz_sv); W.scratch_t = A.get(W.sz_st);
  W.sz_loss = coot_contrastive_scratch_bytes(d.B, d.Nc, 2 * (int)D, (int)D);
  W.loss_scratch = A.get(W.sz_loss);
  const size_t nl = (size_t)(d.B + d.Nc) * D, ng = (size_t)d.B * 2 * D;
  const size_t nrv = (size_t)d.B * d.Cmax_clip * D, nrt = (size_t)d.B * d.Cmax_sent * D;
  W.local_v = A.arr<float>(nl); W.local_t = A.arr<float>(nl); W.glob_v = A.arr<float>(ng); W.glob_t = A.arr<float>(ng);
  W.resh_v = A.arr<float>(nrv); W.resh_t = A.arr<float>(nrt);
  W.dhid_v = A.arr<float>((size_t)d.B * D); W.dhid_t = A.arr<float>((size_t)d.B * D);
  W.dfeat_v = A.arr<float>(nrv); W.dfeat_t = A.arr<float>(nrt);
  W.mask_v = A.arr<unsigned char>((size_t)d.B * d.Cmax_clip); W.mask_t = A.arr<unsigned char>((size_t)d.B * d.Cmax_sent);
  W.lens_v = A.arr<long long>(d.B); W.lens_t = A.arr<long long>(d.B); W.idx = A.arr<long long>(2 * (size_t)d.B);
  // one contiguous block of gradient buffers zeroed with a single memset per step
  A.get(0);
  const size_t z0 = (A.off + 255) & ~(size_t)255;
  W.d_local_v = A.arr<float>(nl); W.d_local_t = A.arr<float>(nl); W.d_glob_v = A.arr<float>(ng); W.d_glob_t = A.arr<float>(ng);
  W.d_resh_v = A.arr<float>(nrv); W.d_resh_t = A.arr<float>(nrt);
  W.zero_begin = (float*)(A.base ? A.base + z0 : nullptr);
  W.zero_bytes = A.off - z0;
}

// packed rows of a side (coot_step_batch.cu_vis / cu_txt + coot_step_dims.tok_vis / tok_txt), or "padded" (cu_seqlens = NULL)
struct SidePacked { coot_packed_seqs v, t; };
SidePacked side_packed(const coot_step_batch& x, const coot_step_dims& d) {
  SidePacked p;
  p.v.cu_seqlens = d.tok_vis > 0 ? x.cu_vis : nullptr; p.v.total_tokens = d.tok_vis; p.v.source = d.source;
  p.t.cu_seqlens = d.tok_txt > 0 ? x.cu_txt : nullptr; p.t.total_tokens = d.tok_txt; p.t.source = d.source;
  return p;
}

int check_cfg(const coot_step_config& c) {
  const int D = c.net[0].hidden_dim;
  for (int i = 0; i < 4; ++i)  // the fp32 reference mode is a per-network checker (coot_net_fwd / coot_net_bwd, eval): not a mode of the step
    COOT_REQUIRE(c.net[i].dtype == COOT_DTYPE_NATIVE, "step: network %d has dtype %d — the step API runs the bf16 path only (this build's 16-bit operand "
                 "format, dtype %d; COOT_DTYPE_F32 is the checker mode of coot_net_fwd / coot_net_bwd)", i, c.net[i].dtype, COOT_DTYPE_NATIVE);
  COOT_REQUIRE(c.net[1].hidden_dim == D && c.net[2].hidden_dim == D && c.net[3].hidden_dim == D, "step: the four networks must share hidden_dim");
  COOT_REQUIRE(c.net[0].use_input_fc && c.net[2].use_input_fc && !c.net[0].use_context && !c.net[2].use_context &&
               c.net[0].pooler == 0 && c.net[2].pooler == 0, "step: local networks = input_fc + atn pooler, no context");
  COOT_REQUIRE(!c.net[1].use_input_fc && !c.net[3].use_input_fc && c.net[1].use_context && c.net[3].use_context &&
               c.net[1].input_dim == D && c.net[3].input_dim == D, "step: global networks = context networks on the local output dim");
  return 0;
}

// Cross-stream ordering.  A hop (event record + stream wait) whose event is already complete costs nothing on the waiting stream
// (tools/micro/hopgap.hip: 1.0 us between its kernels, as without the wait; round 1 read ~30 us per hop off the step stamps, which was
// the cost of the stamps' own event records).  What a hop does cost is the wait itself, so the step is laid out to keep the heavier
// video side on ONE stream from its first launch to its Adam update: the caller may pass side_v == main (hops between equal streams
// vanish), the losses run on the video stream, and waits sit where the data is first read.
struct Hops {
  static constexpr int N = 16;
  hipEvent_t ev[N]; bool made = false;
  int init() {
    if (made) return 0;
    // hipEventDisableSystemFence: by default every hipEventRecord performs a SYSTEM-scope release — an L2 writeback / invalidation
    // that makes device memory visible to the host.  These events only order streams of one device (kernel boundaries already release
    // at device scope), and the fences evicted what the latency-bound global passes keep in L2: step 1.207 -> 1.201 ms
    // (profiles/r04_ab_event_fence.txt)
    for (int i = 0; i < N; ++i) RUN(check_hip(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventDisableSystemFence), "hipEventCreate"));
    made = true; return 0;
  }
  // everything enqueued on `to` after this call runs after everything enqueued on `from` before it
  int hop(int slot, hipStream_t from, hipStream_t to) {
    if (from == to) return 0;
    RUN(record(slot, from));
    return wait(slot, to);
  }
  // the two halves, for a wait that is enqueued later than the record
  int record(int slot, hipStream_t from) {
    RUN(init());
    return check_hip(hipEventRecord(ev[slot], from), "eventRecord");
  }
  int wait(int slot, hipStream_t to) { return check_hip(hipStreamWaitEvent(to, ev[slot], 0), "streamWait"); }
};
thread_local Hops g_hops;
thread_local const uint64_t* g_step_seed_dev = nullptr;  // device base seed of a replayable step (StepState::seed below), else null
thread_local int g_resh_wait_slot = -1;  // hop slot the video side waits on before it reads d_resh (coot_train_step), -1: none
thread_local void* g_glob_done[2] = {nullptr, nullptr};  // optional caller events: video / text global backward done

// Early update of the global networks (coot_train_step): their gradients are final behind the global backward, a whole local backward
// before the step's end — Adam + weight pack of that network run on the internal stream under the local backward instead of at the
// tail of the side's stream (the video side's tail is the end of the step's critical path).  side_backward fires it; slots 11 / 12
// (global backward done), 13 / 14 (update done) per side.  Per side, and only when its local network has kEarlyMinTokens rows or
// more: the two hops and two launches on a third stream take ~60 us, which a long local backward hides (ActivityNet shape 1.192 ->
// 1.184 ms, yc2_2d3d 1.344 -> 1.333) and a short one does not (yc2_100m, 3 840 / 3 072 rows: 0.734 -> 0.751 ms with it);
// profiles/r04_ab_early_update.txt.
constexpr long kEarlyMinTokens = 8192;
struct EarlyUpdate { bool on[2] = {false, false}; const coot_step_config* cfg = nullptr; const coot_step_buffers* b = nullptr; int64_t step = 0; bool repack = false; };
thread_local EarlyUpdate g_early;
int early_update_fire(int side, int gi, hipStream_t st);  // below (needs adam_nets)
int g_grad_write = 1;  // coot_set_option("grad_write", 0/1): coot_train_step writes the weight-matrix gradients and zeroes only the rest
// bf16 weight packs of `count` networks in one launch
int pack_nets(const coot_step_config& c, const coot_step_buffers& b, const int* nets, int count, void* stream) {
  const coot_net_config* cfgs[4]; const float* Ps[4]; void* ws[4];
  for (int k = 0; k < count; ++k) { cfgs[k] = &c.net[nets[k]]; Ps[k] = b.params[nets[k]]; ws[k] = b.wpack[nets[k]]; }
  return coot_nets_pack_weights(count, cfgs, Ps, ws, stream);
}

// Coarse timeline of one step measured with HIP events (coot_set_option("step_stamps", 1); coot_debug_step_stamps()):
// a profiler's launch interception makes this path host-bound, so whether the two sides really overlap can only be
// seen with events recorded by the step itself.
struct StepStamps {
  static constexpr int MAXN = 40;
  hipEvent_t ev[MAXN]; const char* label[MAXN]; int n = 0, made = 0; bool on = false;
  void begin() { n = 0; }
  void mark(const char* what, hipStream_t st) {
    if (!on || n >= MAXN) return;
    if (made <= n) { if (hipEventCreate(&ev[n]) != hipSuccess) return; made = n + 1; }
    if (hipEventRecord(ev[n], st) != hipSuccess) return;
    label[n++] = what;
  }
};
StepStamps g_stamps;

// ---- software-pipelined input LayerNorm (coot_step_set_input_stages / coot_step_set_next_batch, COOT_STEP_INPUT_STAGES) ----------
// The input LayerNorm of the local networks has no parameters of its own (gain / bias are folded into the packed input-FC weights),
// so x^ of batch t + 1 depends on nothing step t computes.  It is the one HBM-bound kernel at the head of the step's critical path
// (2 launches, 390 MB, ~60 us on the video stream) while a quarter of the step — the global networks and the loss between them — runs
// on 8-16 CUs with the memory system idle.  With two caller-owned stages, step t normalises batch t + 1 into the stage it does not
// use, on its own stream behind both sides' local forward passes; step t + 1 finds its x^ ready and starts with the input FC.
extern "C" void coot_internal_set_input_stage(void* xhat, void* pos, int mode);  // api.hip
extern "C" void coot_internal_set_glob_flush_stream(void* stream);            // api.hip
extern "C" void coot_internal_set_pool_pack(const long long* counts, int B, int Cmax, float* out, unsigned char* mask, long long* lens);
extern "C" int coot_internal_pool_handover_taken(void);

struct StageLayout { char *xv, *xt, *pv, *pt; size_t bytes; };
StageLayout stage_layout(const coot_step_config& c, const coot_step_dims& d, void* base) {
  auto pad = [](size_t t) { return (t + 127) & ~(size_t)127; };  // whole 128-row tiles, as the saved arena (api.hip: layout_saved)
  const size_t Tv = pad((size_t)d.B * d.Lv + (size_t)d.Nc * d.Lc), Tt = pad((size_t)d.B * d.Lp + (size_t)d.Nc * d.Ls);
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off = (off + n + 255) & ~(size_t)255; return o; };
  const size_t oxv = take(Tv * c.net[0].input_dim * 2), oxt = take(Tt * c.net[2].input_dim * 2), opv = take(Tv * 4), opt = take(Tt * 4);
  StageLayout L; char* b = (char*)base;
  L.xv = b ? b + oxv : nullptr; L.xt = b ? b + oxt : nullptr; L.pv = b ? b + opv : nullptr; L.pt = b ? b + opt : nullptr; L.bytes = off;
  return L;
}
// ---- streams that really run concurrently (include/coot_hip.h: coot_stream_create_concurrent) ----------------------------------------
// HIP multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in the order they are created; two
// streams on one queue run their kernels one after the other whatever the events between them say.  The step's two sides on one queue:
// 1.72 instead of 1.22 ms (profiles/r06_stream_queues.txt: one more stream created anywhere in the process before the trainer's shifts
// the mapping).  So no stream of the step is taken on trust: a candidate is accepted when a 100-us spin kernel on it and one on every
// stream it has to run beside finish in the time of one.
__global__ void spin_kernel(int ticks /* of the 100 MHz real-time counter */) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(16);
}
struct StreamPicker {
  static constexpr int kSpinTicks = 10000, kNice = 3, kTries = 32, kSettle = 16;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  std::vector<hipStream_t> live;  // streams handed out by coot_stream_create_concurrent and not destroyed yet
  int tests = 0, rejected = 0, unresolved = 0;  // coot_get_option("stream_overlap_tests" / "stream_candidates_rejected" / "stream_unresolved")
  int init() {
    if (e0) return 0;
    RUN(check_hip(hipEventCreate(&e0), "hipEventCreate"));
    RUN(check_hip(hipEventCreate(&e2), "hipEventCreate"));
    RUN(check_hip(hipEventCreateWithFlags(&e1, hipEventDisableTiming), "hipEventCreate"));
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, nullptr, 1);  // (code object loaded before anything is timed)
    return check_hip(hipDeviceSynchronize(), "deviceSynchronize");
  }
  // 1: kernels on a and b overlap; 0: they run one after the other; < 0: error.  Synchronises the device.
  int overlap(hipStream_t a, hipStream_t b) {
    if (a == b) return 0;
    if (init()) return -1;
    ++tests;
    if (check_hip(hipDeviceSynchronize(), "deviceSynchronize")) return -1;
    bool ok = hipEventRecord(e0, a) == hipSuccess;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, kSpinTicks);
    ok = ok && hipStreamWaitEvent(b, e0, 0) == hipSuccess;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, kSpinTicks);
    ok = ok && hipEventRecord(e1, b) == hipSuccess && hipStreamWaitEvent(a, e1, 0) == hipSuccess && hipEventRecord(e2, a) == hipSuccess;
    ok = ok && hipEventSynchronize(e2) == hipSuccess;
    float ms = 0.f;
    ok = ok && hipEventElapsedTime(&ms, e0, e2) == hipSuccess;
    if (!ok) { set_error("stream overlap test: %s", hipGetErrorString(hipGetLastError())); return -1; }
    return ms < 1.6f * (kSpinTicks * 1e-5f) ? 1 : 0;  // one spin: 0.1 ms; two in a row: 0.2 ms
  }
  // a new stream whose kernels overlap those of every stream in must[] — and, if there is such a candidate, of every stream in nice[].
  // Candidates stay alive until the choice is made (a destroyed stream's queue slot may be handed to the next one created).
  int create(const hipStream_t* must, int n_must, const hipStream_t* nice, int n_nice, int priority, hipStream_t* out, int* concurrent) {
    RUN(init());
    int lo = 0, hi = 0;
    RUN(check_hip(hipDeviceGetStreamPriorityRange(&lo, &hi), "streamPriorityRange"));  // lo: numerically largest = lowest priority
    const int prio = priority > 0 ? lo : (priority < 0 ? hi : 0);
    hipStream_t cand[kTries]; int n = 0, pick = -1, pick_must = -1;
    for (; n < kTries && pick < 0; ++n) {
      RUN(check_hip(hipStreamCreateWithPriority(&cand[n], hipStreamNonBlocking, prio), "hipStreamCreate"));
      bool ok = true;
      for (int i = 0; i < n_must && ok; ++i) { const int r = overlap(cand[n], must[i]); if (r < 0) return 1; ok = r == 1; }
      if (!ok) { ++rejected; continue; }
      if (pick_must < 0) pick_must = n;
      for (int i = 0; i < n_nice && ok; ++i) { const int r = overlap(cand[n], nice[i]); if (r < 0) return 1; ok = r == 1; }
      if (ok) pick = n; else ++rejected;
      // HIP hands a new stream the least-loaded queue: with many streams alive the candidates can sit on ONE queue until its load catches
      // up with the others', so the search is long; without a full match after kSettle candidates: settle for must[]
      if (n + 1 >= kSettle && pick < 0 && pick_must >= 0) { ++n; break; }
    }
    const bool full = pick >= 0;
    if (pick < 0) pick = pick_must >= 0 ? pick_must : n - 1;
    if (concurrent) *concurrent = (full || pick_must >= 0) ? 1 : 0;
    if (!full && pick_must < 0) {
      ++unresolved;
      fprintf(stderr, "coot: no stream found that runs concurrently with the step's streams (%d candidates; GPU_MAX_HW_QUEUES too small, "
                      "or a tool that serialises the queues): the step's sides will run one after the other\n", n);
    }
    for (int i = 0; i < n; ++i) if (i != pick) (void)hipStreamDestroy(cand[i]);
    *out = cand[pick];
    return 0;
  }
};
thread_local StreamPicker g_streams;

struct InputPipe {
  void* stage[2] = {nullptr, nullptr}; size_t bytes = 0;
  bool next_valid = false; coot_step_batch next_x; coot_step_dims next_d;  // the batch of the NEXT step (one-shot: consumed by a step)
  bool ready = false; int idx = 0; coot_step_batch have_x; coot_step_dims have_d;  // stage[idx] holds x^ of (have_x, have_d)
  hipEvent_t done = nullptr; hipStream_t stream = nullptr;
  int hits = 0;  // steps of this thread that found their x^ prepared (coot_get_option("stage_hits"): tests assert the timed mode ran)
  hipStream_t beside[2] = {nullptr, nullptr}; bool checked = false;  // the side streams `stream` was verified against
  int init() {
    if (stream) return 0;
    RUN(check_hip(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate"));
    return check_hip(hipEventCreateWithFlags(&done, hipEventDisableTiming | hipEventDisableSystemFence), "hipEventCreate");
  }
  // At the head of a step, before anything is enqueued: the library's own stream (next batch's input LayerNorm, early update of the
  // global networks) must run BESIDE the step's two sides, not in one hardware queue with one of them.  Verified once per pair of side
  // streams (a new trainer brings new ones); replaced if it collides.  Synchronises the device when it has to test; never while `sm`
  // is being captured (a captured step uses neither the input stages nor the early update).
  int ensure_beside(hipStream_t sm, hipStream_t sv, hipStream_t st) {
    if (checked && beside[0] == sv && beside[1] == st) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(sm, &cs) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (cs != hipStreamCaptureStatusNone) return 0;
    bool keep = stream != nullptr;
    if (keep) {
      const int a = g_streams.overlap(stream, sv), b = g_streams.overlap(stream, st);
      if (a < 0 || b < 0) return 1;
      keep = a == 1 && b == 1;
    }
    if (!keep) {
      const hipStream_t must[2] = {sv, st};
      hipStream_t fresh = nullptr; int conc = 0;
      // ... and beside the most recent streams handed to the caller (a data-parallel step's communication stream), if the queues allow
      const int nl = (int)g_streams.live.size(), nn = nl < StreamPicker::kNice ? nl : StreamPicker::kNice;
      RUN(g_streams.create(must, 2, g_streams.live.data() + (nl - nn), nn, 0, &fresh, &conc));
      if (stream) { RUN(check_hip(hipStreamSynchronize(stream), "streamSynchronize")); (void)hipStreamDestroy(stream); }
      stream = fresh;
      if (!done) RUN(check_hip(hipEventCreateWithFlags(&done, hipEventDisableTiming | hipEventDisableSystemFence), "hipEventCreate"));
    }
    beside[0] = sv; beside[1] = st; checked = true;
    return 0;
  }
};
thread_local InputPipe g_pipe;
struct StageScope { ~StageScope() { coot_internal_set_input_stage(nullptr, nullptr, 0); } };  // no exit path leaves a stage set
// what a step (coot_train_step, or coot_step_forward ... coot_step_backward) knows about its stages
struct PipeStep { bool piped = false, hit = false, prefetch = false; int cur = 0; StageLayout SL{}; };
thread_local PipeStep g_phase_pipe;  // of the phase calls: set by coot_step_forward, read and cleared by coot_step_backward
// this batch's x^ in stage[cur] (already there if the previous step normalised it: `hit`), the next batch's into the other
// `announced`: the caller vouches that (x, d) is the batch it passed to coot_step_set_next_batch, unchanged (COOT_STEP_STAGE_ANNOUNCED).
// Pointer and shape equality alone is not an identity: a loader's fixed-shape arena slot is refilled in place.
int pipe_begin(const coot_step_config& cfg, const coot_step_batch& x, const coot_step_dims& d, hipStream_t sv, hipStream_t st, PipeStep& ps,
               bool announced) {
  ps = PipeStep{}; ps.piped = true;
  COOT_REQUIRE(g_pipe.stage[0] && g_pipe.stage[1], "step: input stages requested without coot_step_set_input_stages");
  ps.hit = announced && g_pipe.ready && !memcmp(&g_pipe.have_x, &x, sizeof(x)) && !memcmp(&g_pipe.have_d, &d, sizeof(d));
  ps.cur = ps.hit ? g_pipe.idx : (g_pipe.ready ? g_pipe.idx ^ 1 : 0);
  ps.SL = stage_layout(cfg, d, g_pipe.stage[ps.cur]);
  COOT_REQUIRE(ps.SL.bytes <= g_pipe.bytes, "step: input stages too small (%zu < %zu)", g_pipe.bytes, ps.SL.bytes);
  if (ps.hit) ++g_pipe.hits;
  if (g_pipe.ready) {
    // hit: x^ must have landed.  Miss: the prepared stage is dropped, but the internal stream may still be READING the batch it was
    // given — order this step (and with it whatever the caller enqueues behind it, e.g. a refill of that arena slot) after it
    RUN(check_hip(hipStreamWaitEvent(sv, g_pipe.done, 0), "streamWait"));
    RUN(check_hip(hipStreamWaitEvent(st, g_pipe.done, 0), "streamWait"));
  }
  g_pipe.ready = false;
  ps.prefetch = g_pipe.next_valid;
  return 0;
}
// x^ of the next batch, behind both local forward passes (hop slots 9 / 10: recorded by side_forward)
int pipe_prefetch(const coot_step_config& cfg, const coot_step_buffers& b, const PipeStep& ps, float* dummy_v, float* dummy_t, void* saved_lv,
                  void* saved_lt, int train, uint64_t seed) {
  RUN(g_pipe.init());
  const coot_step_batch& nx = g_pipe.next_x; const coot_step_dims& nd = g_pipe.next_d;
  const StageLayout NL = stage_layout(cfg, nd, g_pipe.stage[ps.cur ^ 1]);
  COOT_REQUIRE(NL.bytes <= g_pipe.bytes, "step: input stages too small for the next batch (%zu < %zu)", g_pipe.bytes, NL.bytes);
  const SidePacked npk = side_packed(nx, nd);
  RUN(g_hops.wait(9, g_pipe.stream));
  RUN(g_hops.wait(10, g_pipe.stream));
  g_pipe.next_valid = false;
  StageScope scope;
  coot_internal_set_input_stage(NL.xv, NL.pv, 1);  // mode 1: coot_net_fwd normalises into the stage and returns
  RUN(coot_net_fwd(&cfg.net[0], b.params[0], b.wpack[0], b.pe[0], nx.vid_feat, nx.vid_len, nd.B, nd.Lv, nx.clip_feat, nx.clip_len, nd.Nc, nd.Lc,
                   nullptr, dummy_v, nullptr, saved_lv, (size_t)-1, nullptr, 0, train, seed, nullptr, g_pipe.stream, &npk.v));
  coot_internal_set_input_stage(NL.xt, NL.pt, 1);
  RUN(coot_net_fwd(&cfg.net[2], b.params[2], b.wpack[2], b.pe[2], nx.par_feat, nx.par_len, nd.B, nd.Lp, nx.sent_feat, nx.sent_len, nd.Nc, nd.Ls,
                   nullptr, dummy_t, nullptr, saved_lt, (size_t)-1, nullptr, 0, train, seed, nullptr, g_pipe.stream, &npk.t));
  coot_internal_set_input_stage(nullptr, nullptr, 0);
  RUN(check_hip(hipEventRecord(g_pipe.done, g_pipe.stream), "eventRecord"));
  g_pipe.ready = true; g_pipe.idx = ps.cur ^ 1; g_pipe.have_x = nx; g_pipe.have_d = nd;
  return 0;
}

// one side (video or text): local(ctx segment + item segment) -> pack -> global
int side_forward(const coot_step_config& c, const coot_step_buffers& b, int li, int gi, const float* ctx_feat, const int64_t* ctx_len,
                 int Lctx, const float* item_feat, const int64_t* item_len, int Litem, const int64_t* item_num, int Cmax,
                 const coot_step_dims& d, float* local_out, float* glob_out, float* resh, unsigned char* mask, long long* lens,
                 void* saved_l, size_t sz_l, void* saved_g, size_t sz_g, int train, uint64_t seed, hipStream_t st, bool pack = true,
                 const coot_packed_seqs* pk = nullptr, int local_done_slot = -1) {
  const int D = c.net[0].hidden_dim;
  if (pack) {
    { const int two[2] = {li, gi}; RUN(pack_nets(c, b, two, 2, st)); }
  }
  g_stamps.mark(li == 0 ? "video: weights packed" : "text: weights packed", st);
  // the local network's pooling kernel writes the item embeddings into the global network's padded layout itself if it can (pool.h: pk_*)
  if (d.Nc > 0) coot_internal_set_pool_pack((const long long*)item_num, d.B, Cmax, resh, mask, lens);
  const int rc_lf = coot_net_fwd(&c.net[li], b.params[li], b.wpack[li], b.pe[li], ctx_feat, ctx_len, d.B, Lctx, item_feat, item_len, d.Nc, Litem,
                                 nullptr, local_out, nullptr, saved_l, sz_l, nullptr, 0, train, seed + 11 * li, g_step_seed_dev, st, pk);
  const bool packed_by_pool = coot_internal_pool_handover_taken() != 0;
  RUN(rc_lf);
  g_stamps.mark(li == 0 ? "video: local forward done" : "text: local forward done", st);
  if (local_done_slot >= 0) RUN(g_hops.record(local_done_slot, st));  // the local embeddings exist: the other side may start on their loss terms
  if (!packed_by_pool) RUN(launch_pack_fwd(local_out + (size_t)d.B * D, (const long long*)item_num, d.B, Cmax, D, resh, mask, lens, st));
  // the two sides' global passes run at the same time: each on its own half of the XCDs (its 3.5 MB of weights then own those L2s;
  // measured harmless rather than useful: the passes are latency chains, profiles/README.md round 3)
  glob_xcd_set(li == 0 ? 0 : 4, 4);
  const int rc_gf = coot_net_fwd(&c.net[gi], b.params[gi], b.wpack[gi], b.pe[gi], resh, item_num, d.B, Cmax, nullptr, nullptr, 0, 0,
                                 local_out /* context = first B rows */, glob_out, nullptr, saved_g, sz_g, nullptr, 0, train, seed + 11 * gi,
                                 g_step_seed_dev, st, nullptr);
  glob_xcd_set(0, 8);
  RUN(rc_gf);
  g_stamps.mark(li == 0 ? "video: global forward done" : "text: global forward done", st);
  return 0;
}

// backward of one side.  d_local [B+Nc, D]: in = loss gradients wrt (context | item embeddings), d_glob [B, 2D],
// d_resh [B, Cmax, D] = cycle-consistency gradient wrt the packed tensor.
int side_backward(const coot_step_config& c, const coot_step_buffers& b, int li, int gi, const float* ctx_feat, const int64_t* ctx_len,
                  int Lctx, const float* item_feat, const int64_t* item_len, int Litem, const int64_t* item_num, int Cmax,
                  const coot_step_dims& d, const float* local_out, const float* resh, float* d_local, const float* d_glob,
                  const float* d_resh, float* dhid, float* dfeat, void* saved_l, size_t sz_l, void* saved_g, size_t sz_g, void* scratch,
                  size_t sz_scratch, int train, uint64_t seed, hipStream_t st, const coot_packed_seqs* pk = nullptr) {
  const int D = c.net[0].hidden_dim;
  g_stamps.mark(li == 0 ? "video: backward starts" : "text: backward starts", st);
  const int side = li == 0 ? 0 : 1;
  // scratch: [local network | global network]
  const size_t sz_loc = align256(coot_net_scratch_bytes(&c.net[li], d.B, Lctx, d.Nc, Litem));
  const size_t sz_glob = coot_net_scratch_bytes(&c.net[gi], d.B, Cmax, 0, 0);
  COOT_REQUIRE(sz_loc + sz_glob <= sz_scratch, "step backward: scratch too small (%zu < %zu)", sz_scratch, sz_loc + sz_glob);
  glob_xcd_set(side == 0 ? 0 : 4, 4);
  // with an early update the global network's weight gradients are consumed on the library's own stream: their launch goes there too
  if (g_early.on[side] && g_pipe.init() == 0) coot_internal_set_glob_flush_stream(g_pipe.stream);
  const int rc_g = coot_net_bwd(&c.net[gi], b.params[gi], b.wpack[gi], b.pe[gi], resh, item_num, d.B, Cmax, nullptr, nullptr, 0, 0, local_out, d_glob,
                                b.grads[gi], dhid, dfeat, saved_g, sz_g, (char*)scratch + sz_loc, sz_glob, train, seed + 11 * gi, g_step_seed_dev, st,
                                nullptr);
  glob_xcd_set(0, 8);
  coot_internal_set_glob_flush_stream(nullptr);
  RUN(rc_g);
  g_stamps.mark(li == 0 ? "video: global backward done" : "text: global backward done", st);
  // data parallel: the global network's gradients are final here — the caller's communication stream may start reducing them under
  // the local backward (coot_step_set_global_done_events)
  if (g_glob_done[side]) {
    if (det_on()) RUN(det_flush_range(b.grads[gi], (size_t)coot_net_param_numel(&c.net[gi]) * sizeof(float), st));  // (their fixed-point sums first)
    RUN(check_hip(hipEventRecord((hipEvent_t)g_glob_done[side], st), "eventRecord"));
  }
  if (g_early.on[side]) RUN(early_update_fire(side, gi, st));
  // the cycle-consistency gradients come from the other stream; they are first needed HERE, a whole global backward after the
  // contrastive loss — waiting only now keeps the cross-stream hop off the critical path
  if (g_resh_wait_slot >= 0 && li == 0) RUN(g_hops.wait(g_resh_wait_slot, st));
  // context grad += dhidden, item grads += unpack(global input grad) + unpack(cycle-consistency grad): one launch
  RUN(launch_pack_bwd_join(dfeat, d_resh, dhid, (const long long*)item_num, d.B, Cmax, D, d_local + (size_t)d.B * D, d_local, st));
  const int rc = coot_net_bwd(&c.net[li], b.params[li], b.wpack[li], b.pe[li], ctx_feat, ctx_len, d.B, Lctx, item_feat, item_len, d.Nc, Litem,
                              nullptr, d_local, b.grads[li], nullptr, nullptr, saved_l, sz_l, scratch, sz_loc, train, seed + 11 * li, g_step_seed_dev,
                              st, pk);
  RUN(rc);
  // deterministic mode (det.h): this side's fixed-point sums -> its two gradient arenas, behind everything that added to them
  if (det_on()) {
    RUN(det_flush_range(b.grads[li], (size_t)coot_net_param_numel(&c.net[li]) * sizeof(float), st));
    RUN(det_flush_range(b.grads[gi], (size_t)coot_net_param_numel(&c.net[gi]) * sizeof(float), st));
  }
  g_stamps.mark(li == 0 ? "video: local backward done" : "text: local backward done", st);
  return 0;
}

// idx[b] uniform in [0, len[b])  — th.multinomial(mask, 1) of coot/loss_fn.py:306-314 on the device
__global__ void sample_idx_kernel(const long long* lens_a, const long long* lens_b, int B, unsigned long long seed, long long* idx,
                                  const unsigned long long* seed_dev = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * B) return;
  seed = eff_seed(seed, seed_dev);
  const long long len = i < B ? lens_a[i] : lens_b[i - B];
  const unsigned r = rng_u32(seed, 0xCCu, (unsigned long long)i);
  long long v = (long long)(((unsigned long long)r * (unsigned long long)len) >> 32);
  idx[i] = v < len ? v : len - 1;
}

// The optimizers of nntrainer/optimization.py:45-181, one elementwise rule each, scalars precomputed on the host:
//   mode 0  torch.optim.Adam (coupled L2 weight decay):  g' = g + wd d p;  m, v <- g';  p -= lr a m / (sqrt(v) b + eps)
//           a = 1 / (1 - beta1^t), b = 1 / sqrt(1 - beta2^t)
//   RAdam (the in-file class, :79-181; decoupled decay p -= wd d lr p, moments from the raw gradient):
//   mode 1  rectified (N_sma >= 5):  p -= a lr m / (sqrt(v) + eps),  a = step_size
//   mode 2  degenerated to SGD:      p -= a lr m
//   mode 3  N_sma < 5 without degenerated_to_sgd: only the moments move
// d = the per-element decay multiplier (decay_mult, model_manager_base.py:152-154), or 1.
struct OptK { int mode; float lr, b1, b2, eps, wd, a, b; };

__device__ __forceinline__ void opt_update(const OptK& k, float& p, float g, float& m, float& v, float d) {
  if (k.mode == 0) {
    const float gj = g + k.wd * d * p;
    m = k.b1 * m + (1.f - k.b1) * gj;
    v = k.b2 * v + (1.f - k.b2) * gj * gj;
    p -= k.lr * k.a * m / (sqrtf(v) * k.b + k.eps);
  } else {
    v = k.b2 * v + (1.f - k.b2) * g * g;
    m = k.b1 * m + (1.f - k.b1) * g;
    if (k.mode == 1) {
      p -= k.wd * d * k.lr * p;
      p -= k.a * k.lr * m / (sqrtf(v) + k.eps);
    } else if (k.mode == 2) {
      p -= k.wd * d * k.lr * p;
      p -= k.a * k.lr * m;
    }
  }
}

__device__ __forceinline__ void opt_update_range(const OptK& k, float* p, const float* g, float* m, float* v, const float* decay, long i, long n) {
  if (i + 3 < n) {
    f32x4_t pp = *reinterpret_cast<f32x4_t*>(p + i), gg = *reinterpret_cast<const f32x4_t*>(g + i);
    f32x4_t mm = *reinterpret_cast<f32x4_t*>(m + i), vv = *reinterpret_cast<f32x4_t*>(v + i);
    f32x4_t dd = decay ? *reinterpret_cast<const f32x4_t*>(decay + i) : f32x4_t{1.f, 1.f, 1.f, 1.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) { float pj = pp[j], mj = mm[j], vj = vv[j]; opt_update(k, pj, gg[j], mj, vj, dd[j]); pp[j] = pj; mm[j] = mj; vv[j] = vj; }
    if (k.mode != 3) *reinterpret_cast<f32x4_t*>(p + i) = pp;
    *reinterpret_cast<f32x4_t*>(m + i) = mm; *reinterpret_cast<f32x4_t*>(v + i) = vv;
  } else {
    for (long e = i; e < n; ++e) opt_update(k, p[e], g[e], m[e], v[e], decay ? decay[e] : 1.f);
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, const float* decay, long n, OptK k) {
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) opt_update_range(k, p, g, m, v, decay, i, n);
}

struct AdamSegs { float* p[4]; const float* g[4]; float* m[4]; float* v[4]; const float* decay[4]; const unsigned char* all1[4]; long n[4]; int blk0[5]; float* losses; };
__global__ __launch_bounds__(256) void adam4_kernel(AdamSegs sg, OptK k_arg, const OptK* k_dev) {
  const OptK k = k_dev ? *k_dev : k_arg;
  int s = 0;
#pragma unroll
  for (int t = 1; t < 4; ++t) if ((int)blockIdx.x >= sg.blk0[t]) s = t;
  // rider: total = contrastive + cycle-consistency (both final long before any update; was a 1-thread launch in front of the text backward)
  if (sg.losses && blockIdx.x == 0 && threadIdx.x == 0) sg.losses[0] = sg.losses[1] + sg.losses[2];
  const long n = sg.n[s];
  const int blk = (int)blockIdx.x - sg.blk0[s];
  const long i = ((long)blk * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  // a workgroup = 1 024 consecutive elements: where the caller marked the decay multiplier as 1.0 on all of them (everything but
  // the blocks that touch a bias vector) the mask is not read (coot_step_buffers.decay_block_all)
  const float* decay = (sg.all1[s] && sg.all1[s][blk]) ? nullptr : sg.decay[s];
  opt_update_range(k, sg.p[s], sg.g[s], sg.m[s], sg.v[s], decay, i, n);
}

// the scalars of the rules above (host, or thread 0 of step_state_kernel); optimizer: 0 = Adam, 1 = RAdam
__host__ __device__ OptK opt_scalars(int optimizer, int radam_degentosgd, float lr, float beta1, float beta2, float eps, float wd, int64_t step) {
  OptK k; k.lr = lr; k.b1 = beta1; k.b2 = beta2; k.eps = eps; k.wd = wd; k.a = 0.f; k.b = 0.f;
  const double b1t = pow((double)beta1, (double)step), b2t = pow((double)beta2, (double)step);
  if (optimizer == 0) {
    k.mode = 0; k.a = (float)(1.0 / (1.0 - b1t)); k.b = (float)(1.0 / sqrt(1.0 - b2t));
  } else {  // nntrainer/optimization.py:144-164
    const double nmax = 2.0 / (1.0 - (double)beta2) - 1.0;
    const double nsma = nmax - 2.0 * (double)step * b2t / (1.0 - b2t);
    if (nsma >= 5.0) {
      k.mode = 1;
      k.a = (float)(sqrt((1.0 - b2t) * (nsma - 4.0) / (nmax - 4.0) * (nsma - 2.0) / nsma * nmax / (nmax - 2.0)) / (1.0 - b1t));
    } else if (radam_degentosgd) {
      k.mode = 2; k.a = (float)(1.0 / (1.0 - b1t));
    } else {
      k.mode = 3;
    }
  }
  return k;
}

__global__ void loss_total_kernel(float* losses) { losses[0] = losses[1] + losses[2]; }

// Per-step scalars in DEVICE memory, for a train step that is captured once and replayed as a hipGraph (a replay cannot change
// kernel arguments): the dropout base seed, the optimizer step count with the scalars derived from it, the learning rate (the
// host rewrites that word when the schedule changes it).  step_state_kernel is the first node of the step: it advances the
// counters exactly as the host does between two eager steps, so eager and replayed steps draw the same masks and take the same
// update.  coot_step_set_device_state(ptr): non-null switches coot_train_step to this mode (its seed argument becomes a salt).
struct StepState { unsigned long long seed, step; float lr; int pad; OptK k; };
thread_local StepState* g_state_dev = nullptr;
__global__ void step_state_kernel(StepState* s, int optimizer, int degen, float b1, float b2, float eps, float wd, unsigned long long seed_inc) {
  s->step += 1; s->seed += seed_inc;
  s->k = opt_scalars(optimizer, degen, s->lr, b1, b2, eps, wd, (int64_t)s->step);
}

// Adam update of `count` parameter arenas in ONE launch (four dependent 13 us launches used to end the step)
int adam_nets(const coot_step_config& cfg, const coot_step_buffers& b, const int* nets, int count, int64_t step, hipStream_t st,
              float* losses = nullptr) {
  AdamSegs sg;
  sg.losses = losses;
  int blk = 0;
  for (int k = 0; k < 4; ++k) {
    const int i = nets[k < count ? k : count - 1];
    const long n = k < count ? (long)coot_net_param_numel(&cfg.net[i]) : 0;
    sg.p[k] = b.params[i]; sg.g[k] = b.grads[i]; sg.m[k] = b.adam_m[i]; sg.v[k] = b.adam_v[i]; sg.decay[k] = b.decay_mask[i]; sg.n[k] = n;
    sg.all1[k] = b.decay_mask[i] ? b.decay_block_all[i] : nullptr;
    sg.blk0[k] = blk;
    blk += (int)((n / 4 + 255) / 256);
  }
  sg.blk0[4] = blk;
  const OptK k = opt_scalars(cfg.optimizer, cfg.radam_degentosgd, cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay, step);
  hipLaunchKernelGGL(adam4_kernel, dim3(blk), dim3(256), 0, st, sg, k, g_state_dev ? (const OptK*)&g_state_dev->k : (const OptK*)nullptr);
  COOT_CHECK_LAUNCH("adam4");
  return 0;
}

// The cycle-consistency loss scores ONE valid position per video and direction (coot/loss_fn.py:306-314).  Normally drawn on the
// device from the step seed; a caller that has to reproduce a given draw (parity tests against the reference's th.multinomial
// sequence) injects the 2B indices with coot_step_set_cycle_indices.
thread_local const int64_t* g_cc_idx_inject = nullptr;
int draw_cycle_indices(const coot_step_batch& x, const coot_step_dims& d, uint64_t seed, long long* idx, hipStream_t s) {
  if (g_cc_idx_inject)
    return check_hip(hipMemcpyAsync(idx, g_cc_idx_inject, 2 * (size_t)d.B * sizeof(long long), hipMemcpyDeviceToDevice, s), "cycle indices");
  hipLaunchKernelGGL(sample_idx_kernel, dim3((2 * d.B + 255) / 256), dim3(256), 0, s, (const long long*)x.clip_num, (const long long*)x.sent_num,
                     d.B, (unsigned long long)seed, idx, (const unsigned long long*)g_step_seed_dev);
  COOT_CHECK_LAUNCH("sample_idx");
  return 0;
}

int early_update_fire(int side, int gi, hipStream_t st) {
  RUN(g_pipe.init());
  RUN(g_hops.hop(11 + side, st, g_pipe.stream));
  const int nets[1] = {gi};
  RUN(adam_nets(*g_early.cfg, *g_early.b, nets, 1, g_early.step, g_pipe.stream));
  if (g_early.repack) RUN(pack_nets(*g_early.cfg, *g_early.b, nets, 1, g_pipe.stream));
  return g_hops.record(13 + side, g_pipe.stream);
}

}  // namespace

extern "C" {

static int optimizer_step(int optimizer, int degen, float* params, const float* grads, float* m, float* v, const float* decay_mask, int64_t n,
                          float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, coot_stream_t stream) {
  COOT_REQUIRE(params && grads && m && v && step >= 1, "optimizer step: bad arguments");
  if (n <= 0) return 0;
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, decay_mask, (long)n,
                     opt_scalars(optimizer, degen, lr, beta1, beta2, eps, weight_decay, step));
  COOT_CHECK_LAUNCH("optimizer step");
  return 0;
}

int coot_adam_step(float* params, const float* grads, float* m, float* v, const float* decay_mask, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int64_t step, coot_stream_t stream) {
  return optimizer_step(0, 0, params, grads, m, v, decay_mask, n, lr, beta1, beta2, eps, weight_decay, step, stream);
}

int coot_radam_step(float* params, const float* grads, float* m, float* v, const float* decay_mask, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int64_t step, int degenerated_to_sgd, coot_stream_t stream) {
  return optimizer_step(1, degenerated_to_sgd, params, grads, m, v, decay_mask, n, lr, beta1, beta2, eps, weight_decay, step, stream);
}

size_t coot_step_workspace_bytes(const coot_step_config* cfg, const coot_step_dims* dims) {
  Bump A(nullptr, 0); StepWs W; layout_step(*cfg, *dims, A, W); return A.off + 512;
}

int coot_step_forward(const coot_step_config* cfg, const coot_step_buffers* b, const coot_step_batch* x, const coot_step_dims* d,
                      float* local_v, float* local_t, float* glob_v, float* glob_t, float* resh_v, float* resh_t, void* workspace,
                      size_t workspace_bytes, int train, uint64_t seed, int packs_fresh, coot_stream_t main_s, coot_stream_t side_v,
                      coot_stream_t side_t) {
  RUN(check_cfg(*cfg));
  COOT_REQUIRE((packs_fresh & ~(COOT_FWD_PACKS_FRESH | COOT_FWD_INPUT_STAGES | COOT_FWD_STAGE_ANNOUNCED)) == 0, "step_forward: unknown bits in packs_fresh (%d)", packs_fresh);
  Bump A(workspace, workspace_bytes); StepWs W; layout_step(*cfg, *d, A, W);
  const SidePacked pk = side_packed(*x, *d);
  COOT_REQUIRE(!A.overflow, "step: workspace too small (%zu < %zu)", workspace_bytes, A.off);
  hipStream_t sm = (hipStream_t)main_s, sv = (hipStream_t)side_v, st = (hipStream_t)side_t;
  RUN(g_pipe.ensure_beside(sm, sv, st));
  RUN(g_hops.hop(0, sm, sv));
  RUN(g_hops.hop(1, sm, st));
  const bool fresh = (packs_fresh & COOT_FWD_PACKS_FRESH) != 0;
  StageScope stage_scope;
  PipeStep ps;
  g_phase_pipe = PipeStep{};
  if (packs_fresh & COOT_FWD_INPUT_STAGES) RUN(pipe_begin(*cfg, *x, *d, sv, st, ps, (packs_fresh & COOT_FWD_STAGE_ANNOUNCED) != 0));
  if (ps.piped) coot_internal_set_input_stage(ps.SL.xv, ps.SL.pv, ps.hit ? 2 : 0);
  RUN(side_forward(*cfg, *b, 0, 1, x->vid_feat, x->vid_len, d->Lv, x->clip_feat, x->clip_len, d->Lc, x->clip_num, d->Cmax_clip, *d,
                   local_v, glob_v, resh_v, W.mask_v, W.lens_v, W.saved_lv, W.sz_lv, W.saved_gv, W.sz_gv, train, seed, sv, !fresh, &pk.v,
                   ps.prefetch ? 9 : -1));
  if (ps.piped) coot_internal_set_input_stage(ps.SL.xt, ps.SL.pt, ps.hit ? 2 : 0);
  RUN(side_forward(*cfg, *b, 2, 3, x->par_feat, x->par_len, d->Lp, x->sent_feat, x->sent_len, d->Ls, x->sent_num, d->Cmax_sent, *d,
                   local_t, glob_t, resh_t, W.mask_t, W.lens_t, W.saved_lt, W.sz_lt, W.saved_gt, W.sz_gt, train, seed + 1000, st, !fresh, &pk.t,
                   ps.prefetch ? 10 : -1));
  coot_internal_set_input_stage(nullptr, nullptr, 0);
  if (ps.prefetch) RUN(pipe_prefetch(*cfg, *b, ps, local_v, local_t, W.saved_lv, W.saved_lt, train, seed));
  g_phase_pipe = ps;  // coot_step_backward reads x^ of this batch in the same stage
  RUN(g_hops.hop(2, sv, sm));
  RUN(g_hops.hop(3, st, sm));
  return 0;
}

int coot_step_backward(const coot_step_config* cfg, const coot_step_buffers* b, const coot_step_batch* x, const coot_step_dims* d,
                       const float* local_v, const float* local_t, const float* resh_v, const float* resh_t, float* d_local_v,
                       float* d_local_t, const float* d_glob_v, const float* d_glob_t, const float* d_resh_v, const float* d_resh_t,
                       void* workspace, size_t workspace_bytes, int train, uint64_t seed, coot_stream_t main_s, coot_stream_t side_v,
                       coot_stream_t side_t) {
  RUN(check_cfg(*cfg));
  COOT_REQUIRE(!COOT_OPERAND_IS_F16, "coot_step_backward: the f16 operand build is forward-only (no GradScaler: coot/trainer_retrieval.py:277-285); coot_step_forward runs");
  Bump A(workspace, workspace_bytes); StepWs W; layout_step(*cfg, *d, A, W);
  const SidePacked pk = side_packed(*x, *d);
  COOT_REQUIRE(!A.overflow, "step: workspace too small");
  hipStream_t sm = (hipStream_t)main_s, sv = (hipStream_t)side_v, st = (hipStream_t)side_t;
  RUN(g_hops.hop(0, sm, sv));
  RUN(g_hops.hop(1, sm, st));
  StageScope stage_scope;
  const PipeStep ps = g_phase_pipe;  // the stage coot_step_forward kept this batch's x^ in (if any)
  g_phase_pipe = PipeStep{};
  if (ps.piped) coot_internal_set_input_stage(ps.SL.xv, ps.SL.pv, 0);
  RUN(side_backward(*cfg, *b, 0, 1, x->vid_feat, x->vid_len, d->Lv, x->clip_feat, x->clip_len, d->Lc, x->clip_num, d->Cmax_clip, *d, local_v,
                    resh_v, d_local_v, d_glob_v, d_resh_v, W.dhid_v, W.dfeat_v, W.saved_lv, W.sz_lv, W.saved_gv, W.sz_gv, W.scratch_v,
                    W.sz_sv, train, seed, sv, &pk.v));
  if (ps.piped) coot_internal_set_input_stage(ps.SL.xt, ps.SL.pt, 0);
  RUN(side_backward(*cfg, *b, 2, 3, x->par_feat, x->par_len, d->Lp, x->sent_feat, x->sent_len, d->Ls, x->sent_num, d->Cmax_sent, *d, local_t,
                    resh_t, d_local_t, d_glob_t, d_resh_t, W.dhid_t, W.dfeat_t, W.saved_lt, W.sz_lt, W.saved_gt, W.sz_gt, W.scratch_t,
                    W.sz_st, train, seed + 1000, st, &pk.t));
  RUN(g_hops.hop(2, sv, sm));
  RUN(g_hops.hop(3, st, sm));
  return 0;
}

// optimizer update of the four networks (video side on side_v, text side on side_t, one launch each) + optional repack of the
// bf16 weight packs; main is ordered after both
int coot_step_update(const coot_step_config* cfg, const coot_step_buffers* b, int64_t step, int repack, float* losses, coot_stream_t main_s,
                     coot_stream_t side_v, coot_stream_t side_t) {
  RUN(check_cfg(*cfg));
  COOT_REQUIRE(step >= 1, "step_update: step counts from 1");
  COOT_REQUIRE((repack & ~(COOT_UPDATE_REPACK | COOT_UPDATE_DEFER_TEXT_JOIN | COOT_UPDATE_SKIP_GLOBAL | COOT_UPDATE_GLOBAL_ONLY)) == 0,
               "step_update: unknown bits in repack (%d): a bit mask since ABI 5", repack);
  hipStream_t sm = (hipStream_t)main_s, sv = (hipStream_t)side_v, st = (hipStream_t)side_t;
  const bool do_pack = (repack & COOT_UPDATE_REPACK) != 0;
  if (repack & COOT_UPDATE_GLOBAL_ONLY) {
    // the two GLOBAL networks only, on main_s alone: their gradients are final (and, data parallel, reduced) a whole local backward
    // before the step's end — the caller runs this on its communication stream behind their bucket (as coot_train_step's early update)
    COOT_REQUIRE((repack & (COOT_UPDATE_SKIP_GLOBAL | COOT_UPDATE_DEFER_TEXT_JOIN)) == 0, "step_update: GLOBAL_ONLY excludes SKIP_GLOBAL / DEFER_TEXT_JOIN");
    const int gnets[2] = {1, 3};
    RUN(adam_nets(*cfg, *b, gnets, 2, step, sm));
    if (do_pack) RUN(pack_nets(*cfg, *b, gnets, 2, main_s));
    return 0;
  }
  RUN(g_hops.hop(0, sm, sv));
  RUN(g_hops.hop(1, sm, st));
  const int vnets[2] = {0, 1}, tnets[2] = {2, 3};
  const int per_side = (repack & COOT_UPDATE_SKIP_GLOBAL) ? 1 : 2;  // (SKIP_GLOBAL: a GLOBAL_ONLY call of this step already updated them)
  // total = contrastive + cycle-consistency rides on the VIDEO side's launch: with COOT_UPDATE_DEFER_TEXT_JOIN main_s is ordered after
  // that side only, and all three loss words are readable there (as in coot_train_step)
  RUN(adam_nets(*cfg, *b, vnets, per_side, step, sv, losses));
  if (do_pack) RUN(pack_nets(*cfg, *b, vnets, per_side, side_v));
  RUN(adam_nets(*cfg, *b, tnets, per_side, step, st));
  if (do_pack) RUN(pack_nets(*cfg, *b, tnets, per_side, side_t));
  RUN(g_hops.hop(4, sv, sm));
  if ((repack & COOT_UPDATE_DEFER_TEXT_JOIN) == 0) RUN(g_hops.hop(5, st, sm));
  return 0;
}

// idx[0 .. B) / idx[B .. 2B): one valid clip / sentence position per video (th.multinomial(mask, 1), coot/loss_fn.py:306-314)
int coot_sample_cycle_indices(const int64_t* clip_num, const int64_t* sent_num, int B, uint64_t seed, int64_t* idx, coot_stream_t stream) {
  COOT_REQUIRE(clip_num && sent_num && idx && B >= 0, "sample_cycle_indices: bad arguments");
  if (B == 0) return 0;
  hipLaunchKernelGGL(sample_idx_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const long long*)clip_num,
                     (const long long*)sent_num, B, (unsigned long long)seed, (long long*)idx);
  COOT_CHECK_LAUNCH("sample_idx");
  return 0;
}

int coot_train_step(const coot_step_config* cfg, const coot_step_buffers* b, const coot_step_batch* x, const coot_step_dims* d,
                    float* losses, void* workspace, size_t workspace_bytes, int train, uint64_t seed, int64_t step, int do_optimizer,
                    coot_stream_t main_s, coot_stream_t side_v, coot_stream_t side_t) {
  RUN(check_cfg(*cfg));
  COOT_REQUIRE(!COOT_OPERAND_IS_F16, "coot_train_step: the f16 operand build is forward-only (no GradScaler: coot/trainer_retrieval.py:277-285); coot_step_forward runs");
  COOT_REQUIRE(losses, "train_step: losses pointer");
  COOT_REQUIRE((do_optimizer & ~(COOT_STEP_OPTIMIZER | COOT_STEP_REPACK | COOT_STEP_PACKS_FRESH | COOT_STEP_DEFER_TEXT_JOIN | COOT_STEP_INPUT_STAGES |
                                 COOT_STEP_STAGE_ANNOUNCED)) == 0, "train_step: unknown bits in do_optimizer (%d)", do_optimizer);
  Bump A(workspace, workspace_bytes); StepWs W; layout_step(*cfg, *d, A, W);
  const SidePacked pk = side_packed(*x, *d);
  COOT_REQUIRE(!A.overflow, "train_step: workspace too small (%zu < %zu)", workspace_bytes, A.off);
  hipStream_t sm = (hipStream_t)main_s, sv = (hipStream_t)side_v, st = (hipStream_t)side_t;
  COOT_REQUIRE(sv != st, "train_step: the two side streams must differ");
  RUN(g_pipe.ensure_beside(sm, sv, st));
  const int D = cfg->net[0].hidden_dim;
  const bool optimize = (do_optimizer & COOT_STEP_OPTIMIZER) != 0, repack = optimize && (do_optimizer & COOT_STEP_REPACK) != 0;
  const bool pack_first = (do_optimizer & COOT_STEP_PACKS_FRESH) == 0;
  g_stamps.begin();
  g_stamps.mark("step starts", sm);
  if (g_state_dev) {  // replayable step: counters and optimizer scalars advance on the device, before the two sides fork
    COOT_REQUIRE(optimize, "train_step: the device step state is for optimizer steps");
    hipLaunchKernelGGL(step_state_kernel, dim3(1), dim3(1), 0, sm, g_state_dev, cfg->optimizer, cfg->radam_degentosgd, cfg->beta1, cfg->beta2,
                       cfg->eps, cfg->weight_decay, (unsigned long long)7919);
    COOT_CHECK_LAUNCH("step_state");
  }
  RUN(g_hops.hop(0, sm, sv));
  RUN(g_hops.hop(1, sm, st));
  const bool piped = (do_optimizer & COOT_STEP_INPUT_STAGES) != 0;
  StageScope stage_scope;
  PipeStep ps;
  COOT_REQUIRE(!(piped && g_state_dev), "train_step: input stages are not available in a replayable (captured) step");
  if (piped) RUN(pipe_begin(*cfg, *x, *d, sv, st, ps, (do_optimizer & COOT_STEP_STAGE_ANNOUNCED) != 0));
  const StageLayout& SL = ps.SL; const bool hit = ps.hit;
  const bool prefetch = ps.prefetch;
  if (piped) coot_internal_set_input_stage(SL.xv, SL.pv, hit ? 2 : 0);
  RUN(side_forward(*cfg, *b, 0, 1, x->vid_feat, x->vid_len, d->Lv, x->clip_feat, x->clip_len, d->Lc, x->clip_num, d->Cmax_clip, *d,
                   W.local_v, W.glob_v, W.resh_v, W.mask_v, W.lens_v, W.saved_lv, W.sz_lv, W.saved_gv, W.sz_gv, train, seed, sv, pack_first, &pk.v,
                   prefetch ? 9 : -1));
  if (piped) coot_internal_set_input_stage(SL.xt, SL.pt, hit ? 2 : 0);
  RUN(side_forward(*cfg, *b, 2, 3, x->par_feat, x->par_len, d->Lp, x->sent_feat, x->sent_len, d->Ls, x->sent_num, d->Cmax_sent, *d,
                   W.local_t, W.glob_t, W.resh_t, W.mask_t, W.lens_t, W.saved_lt, W.sz_lt, W.saved_gt, W.sz_gt, train, seed + 1000, st,
                   pack_first, &pk.t, prefetch ? 10 : -1));
  coot_internal_set_input_stage(nullptr, nullptr, 0);
  // x^ of the next batch, behind both local forward passes (the chip's memory system is idle from there to the local backward)
  if (prefetch) RUN(pipe_prefetch(*cfg, *b, ps, W.local_v, W.local_t, W.saved_lv, W.saved_lt, train, seed));
  // ---- losses.  The text side's forward is the later one (it gets the CUs the three times larger video side leaves: global forward
  // done at ~450 us against ~405 us, profiles/r04_step_timeline.txt) and the video side's backward the longer one (it ends ~200 us
  // after the text side's).  So the video stream carries as little as possible between the text side's last forward launch and its own
  // backward: the zero fill runs there BEFORE the join (in the gap the video side waits anyway), behind the join only the (vid, par)
  // terms (64 rows).  Everything else — the cycle-consistency loss, the contrastive terms on the clip / sentence embeddings and on
  // the context vectors — runs on the text stream next to the video side's global backward, which does not read their gradients
  // (slot 7 is waited for where d_resh_v / d_local_v are first read: behind it).  (Until round 4 the fill and the local terms sat on
  // the text stream in front of the join — a layout from when the text side finished first: join 26 us behind the text forward.)
  //
  // Zero fill (ONE launch): the vectors of the 4 gradient arenas (weight-matrix gradients are written, not accumulated, by this step's
  // backward — g_grad_write), the embedding-gradient block and the loss words.  Their previous readers: the last step's backward,
  // update and loss rider — on the video stream, or on the text stream BEFORE this step's local forward there (slot 10; also with a
  // deferred text join).  Everything that accumulates into them is ordered behind this launch: the (vid, par) terms on this stream,
  // the text stream's losses through slot 6, the text backward through slot 3.
  // (without a next batch to prepare there is no record on the text stream behind its local forward, and adding one costs more than
  // the fill: a marker in the middle of the forward's critical stream — profiles/README.md round 4; the fill then stays where it was,
  // at the end of the text stream's forward)
  const hipStream_t sz = prefetch ? sv : st;
  if (prefetch) RUN(g_hops.wait(10, sv));
  {
    const coot_net_config* cfgs[4] = {&cfg->net[0], &cfg->net[1], &cfg->net[2], &cfg->net[3]};
    float* extra[2] = {W.zero_begin, losses};
    const int64_t extra_n[2] = {(int64_t)(W.zero_bytes / sizeof(float)), 3};
    RUN(coot_nets_zero_grads_ex(4, cfgs, b->grads, g_grad_write, extra, extra_n, 2, (coot_stream_t)sz));
  }
  g_stamps.mark(prefetch ? "video: gradients zeroed" : "text: gradients zeroed", sz);
  // slot 6: the video side's embeddings exist (and, recorded behind the fill, every word the text stream's losses add to is zero)
  RUN(g_hops.record(6, sv));
  const bool cc = cfg->cc_weight != 0.f;
  RUN(g_hops.record(2, st));  // the text side's forward is done
  RUN(g_hops.wait(6, st));
  if (cc) {  // cycle-consistency -> losses[2]
    RUN(draw_cycle_indices(*x, *d, seed, W.idx, st));
    RUN(coot_cyclecons_fwd_bwd(W.resh_v, W.resh_t, x->clip_num, x->sent_num, (const int64_t*)W.idx, (const int64_t*)(W.idx + d->B), d->B,
                               d->Cmax_clip, d->Cmax_sent, D, cfg->cc_weight, 1.0f / (float)d->B, losses + 2, nullptr, nullptr, W.d_resh_v,
                               W.d_resh_t, side_t));
    g_stamps.mark("text: cycle-consistency done", st);
  }
  // (clip, sent) and (vid_ctx, par_ctx) terms -> losses[1] (one atomic add per call next to the (vid, par) terms' on the zeroed word:
  // two addends commute, the same bits in either order)
  RUN(coot_contrastive_fwd_bwd_part(&cfg->contr, d->B, d->Nc, 2 * D, D, W.glob_v, W.glob_t, W.local_v + (size_t)d->B * D,
                                    W.local_t + (size_t)d->B * D, W.local_v, W.local_t, losses + 1, W.d_glob_v, W.d_glob_t,
                                    W.d_local_v + (size_t)d->B * D, W.d_local_t + (size_t)d->B * D, W.d_local_v, W.d_local_t, W.loss_scratch,
                                    W.sz_loss, COOT_CONTRASTIVE_LOCAL, side_t));
  g_stamps.mark("text: local contrastive terms done", st);
  RUN(g_hops.record(7, st));
  RUN(g_hops.wait(2, sv));
  g_stamps.mark("video: text forward joined", sv);
  // (vid, par) terms -> losses[1]
  RUN(coot_contrastive_fwd_bwd_part(&cfg->contr, d->B, d->Nc, 2 * D, D, W.glob_v, W.glob_t, W.local_v + (size_t)d->B * D,
                                    W.local_t + (size_t)d->B * D, W.local_v, W.local_t, losses + 1, W.d_glob_v, W.d_glob_t,
                                    W.d_local_v + (size_t)d->B * D, W.d_local_t + (size_t)d->B * D, W.d_local_v, W.d_local_t, W.loss_scratch,
                                    W.sz_loss, COOT_CONTRASTIVE_GLOBAL, side_v));
  g_stamps.mark("video: contrastive done", sv);
  RUN(g_hops.hop(3, sv, st));  // text backward needs the (vid, par) gradients
  const int vnets[2] = {0, 1}, tnets[2] = {2, 3};
  // what the text stream produced for the video side's backward (the cycle-consistency gradient d_resh_v, the local contrastive
  // terms' gradients d_local_v) was recorded in slot 7; side_backward waits where it is first read
  g_resh_wait_slot = 7;
  // (not in deterministic mode: the fixed-point sums of an arena are flushed at the end of its side's backward; not in a captured step)
  const bool early = optimize && !det_on() && !g_state_dev;
  const bool early_v = early && (long)d->B * d->Lv + (long)d->Nc * d->Lc >= kEarlyMinTokens;
  const bool early_t = early && (long)d->B * d->Lp + (long)d->Nc * d->Ls >= kEarlyMinTokens;
  struct EarlyScope { ~EarlyScope() { g_early = EarlyUpdate{}; } } early_scope;
  g_early.on[0] = early_v; g_early.on[1] = early_t; g_early.cfg = cfg; g_early.b = b; g_early.step = step; g_early.repack = repack;
  (void)coot_net_grads_overwrite(g_grad_write);
  if (piped) coot_internal_set_input_stage(SL.xv, SL.pv, 0);  // the weight gradient of the input FC reads x^ there
  const int rc_v = side_backward(*cfg, *b, 0, 1, x->vid_feat, x->vid_len, d->Lv, x->clip_feat, x->clip_len, d->Lc, x->clip_num, d->Cmax_clip, *d,
                                 W.local_v, W.resh_v, W.d_local_v, W.d_glob_v, cc ? W.d_resh_v : nullptr, W.dhid_v, W.dfeat_v, W.saved_lv, W.sz_lv,
                                 W.saved_gv, W.sz_gv, W.scratch_v, W.sz_sv, train, seed, sv, &pk.v);
  g_resh_wait_slot = -1;
  if (rc_v) (void)coot_net_grads_overwrite(0);
  RUN(rc_v);
  // total = contrastive + cycle-consistency rides on the VIDEO side's update launch: both words are final on this stream (its backward
  // waited for the text stream's loss terms, slot 7), and with COOT_STEP_DEFER_TEXT_JOIN the caller's stream is ordered after the video
  // side only — all three loss words are readable there on return (on the text side's launch, losses[0] raced with a deferred join)
  if (det_on()) RUN(det_flush_range(losses, 3 * sizeof(float), sv));  // (the cycle-consistency word: 2 B addends, det.h)
  if (optimize) RUN(adam_nets(*cfg, *b, vnets, early_v ? 1 : 2, step, sv, losses));  // (early: the global network is being updated already)
  else {
    hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(1), 0, sv, losses);
    COOT_CHECK_LAUNCH("loss_total");
  }
  if (repack) RUN(pack_nets(*cfg, *b, vnets, early_v ? 1 : 2, side_v));
  if (early_v) RUN(g_hops.wait(13, sv));
  g_stamps.mark("video: updated", sv);
  if (piped) coot_internal_set_input_stage(SL.xt, SL.pt, 0);
  const int rc_t = side_backward(*cfg, *b, 2, 3, x->par_feat, x->par_len, d->Lp, x->sent_feat, x->sent_len, d->Ls, x->sent_num, d->Cmax_sent, *d, W.local_t,
                                 W.resh_t, W.d_local_t, W.d_glob_t, cc ? W.d_resh_t : nullptr, W.dhid_t, W.dfeat_t, W.saved_lt, W.sz_lt, W.saved_gt, W.sz_gt,
                                 W.scratch_t, W.sz_st, train, seed + 1000, st, &pk.t);
  (void)coot_net_grads_overwrite(0);
  coot_internal_set_input_stage(nullptr, nullptr, 0);
  RUN(rc_t);
  if (optimize) RUN(adam_nets(*cfg, *b, tnets, early_t ? 1 : 2, step, st));
  if (repack) RUN(pack_nets(*cfg, *b, tnets, early_t ? 1 : 2, side_t));
  if (early_t) RUN(g_hops.wait(14, st));
  g_stamps.mark("text: updated", st);
  RUN(g_hops.hop(4, sv, sm));
  if ((do_optimizer & COOT_STEP_DEFER_TEXT_JOIN) == 0) RUN(g_hops.hop(5, st, sm));  // else: the caller (or the next step's text side) orders it
  g_stamps.mark("step done", sm);
  return 0;
}

void coot_step_stamps_enable(int on) { g_stamps.on = on != 0; }
int coot_internal_stage_hits(void) { return g_pipe.hits; }
size_t coot_step_device_state_bytes(void) { return sizeof(StepState); }
int coot_step_set_device_state(void* state) {
  g_state_dev = (StepState*)state;
  g_step_seed_dev = state ? (const uint64_t*)&g_state_dev->seed : nullptr;
  return 0;
}
int coot_step_set_cycle_indices(const int64_t* idx) { g_cc_idx_inject = idx; return 0; }
size_t coot_step_input_stage_bytes(const coot_step_config* cfg, const coot_step_dims* dims) {
  if (!cfg || !dims) return 0;
  return stage_layout(*cfg, *dims, nullptr).bytes;
}
int coot_step_set_input_stages(void* stage0, void* stage1, size_t bytes_each) {
  COOT_REQUIRE((!stage0 && !stage1) || (stage0 && stage1 && stage0 != stage1), "set_input_stages: two distinct buffers, or both NULL");
  if (stage0 == g_pipe.stage[0] && stage1 == g_pipe.stage[1] && bytes_each == g_pipe.bytes) return 0;  // unchanged: what a stage holds stays valid
  g_pipe.stage[0] = stage0; g_pipe.stage[1] = stage1; g_pipe.bytes = bytes_each; g_pipe.ready = false; g_pipe.next_valid = false;
  return 0;
}
int coot_step_set_next_batch(const coot_step_batch* next, const coot_step_dims* next_dims) {
  COOT_REQUIRE((next == nullptr) == (next_dims == nullptr), "set_next_batch: batch and dims, or both NULL");
  g_pipe.next_valid = next != nullptr;
  if (next) { g_pipe.next_x = *next; g_pipe.next_d = *next_dims; }
  return 0;
}
int coot_step_set_global_done_events(void* ev_video, void* ev_text) { g_glob_done[0] = ev_video; g_glob_done[1] = ev_text; return 0; }

// caller-visible fence-free events (include/coot_hip.h: COOT_SYNC_EVENTS) + a ring for coot_stream_hop
namespace {
struct SyncEvents {
  static constexpr int RING = 16;
  hipEvent_t ev[COOT_SYNC_EVENTS + RING]; bool made = false; int next = 0;
  int init() {
    if (made) return 0;
    for (int i = 0; i < COOT_SYNC_EVENTS + RING; ++i)
      RUN(check_hip(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventDisableSystemFence), "hipEventCreate"));
    made = true; return 0;
  }
};
thread_local SyncEvents g_sync;
}  // namespace
int coot_event_record(int slot, coot_stream_t stream) {
  COOT_REQUIRE(slot >= 0 && slot < COOT_SYNC_EVENTS, "event_record: slot %d", slot);
  RUN(g_sync.init());
  return check_hip(hipEventRecord(g_sync.ev[slot], (hipStream_t)stream), "eventRecord");
}
int coot_event_wait(int slot, coot_stream_t stream) {
  COOT_REQUIRE(slot >= 0 && slot < COOT_SYNC_EVENTS, "event_wait: slot %d", slot);
  RUN(g_sync.init());
  return check_hip(hipStreamWaitEvent((hipStream_t)stream, g_sync.ev[slot], 0), "streamWait");
}
void* coot_event_handle(int slot) {
  if (slot < 0 || slot >= COOT_SYNC_EVENTS || g_sync.init()) return nullptr;
  return (void*)g_sync.ev[slot];
}
int coot_stream_hop(coot_stream_t from, coot_stream_t to) {
  if (from == to) return 0;
  RUN(g_sync.init());
  hipEvent_t e = g_sync.ev[COOT_SYNC_EVENTS + g_sync.next];
  g_sync.next = (g_sync.next + 1) % SyncEvents::RING;  // (a waiter captures the record it was enqueued behind: re-recording later is safe)
  RUN(check_hip(hipEventRecord(e, (hipStream_t)from), "eventRecord"));
  return check_hip(hipStreamWaitEvent((hipStream_t)to, e, 0), "streamWait");
}
int coot_streams_overlap(coot_stream_t a, coot_stream_t b) { return g_streams.overlap((hipStream_t)a, (hipStream_t)b); }
int coot_stream_create_concurrent(const coot_stream_t* others, int n_others, int priority, coot_stream_t* out, int* concurrent) {
  COOT_REQUIRE(out && n_others >= 0 && n_others <= 8 && (others || n_others == 0), "stream_create_concurrent: arguments");
  hipStream_t must[8];
  for (int i = 0; i < n_others; ++i) must[i] = (hipStream_t)others[i];
  hipStream_t s = nullptr;
  // beside the caller's streams — and, if the queues allow it, beside the library's own stream too
  const hipStream_t nice[1] = {g_pipe.stream};
  RUN(g_streams.create(must, n_others, nice, g_pipe.stream ? 1 : 0, priority, &s, concurrent));
  g_streams.live.push_back(s);
  *out = (coot_stream_t)s;
  return 0;
}
int coot_stream_destroy(coot_stream_t s) {
  for (size_t i = 0; i < g_streams.live.size(); ++i)
    if (g_streams.live[i] == (hipStream_t)s) {
      g_streams.live.erase(g_streams.live.begin() + i);
      RUN(check_hip(hipStreamSynchronize((hipStream_t)s), "streamSynchronize"));
      if (g_pipe.checked && (g_pipe.beside[0] == (hipStream_t)s || g_pipe.beside[1] == (hipStream_t)s)) g_pipe.checked = false;  // (the handle may be reused)
      return check_hip(hipStreamDestroy((hipStream_t)s), "streamDestroy");
    }
  set_error("stream_destroy: not a stream of coot_stream_create_concurrent (of this thread)");
  return 1;
}
int coot_internal_stream_counter(int which) { return which == 0 ? g_streams.tests : (which == 1 ? g_streams.rejected : g_streams.unresolved); }
void coot_step_grad_write(int on) { g_grad_write = on ? 1 : 0; }

// text table of the last step's stamps (ms since "step starts"); synchronises the device.  Returns the number of stamps.
int coot_debug_step_stamps(char* buf, int buf_bytes) {
  if (!buf || buf_bytes <= 0) return 0;
  buf[0] = 0;
  if (g_stamps.n == 0) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  int off = 0;
  for (int i = 0; i < g_stamps.n; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_stamps.ev[0], g_stamps.ev[i]) != hipSuccess) ms = -1.f;
    off += snprintf(buf + off, off < buf_bytes ? buf_bytes - off : 0, "%9.1f us  %s\n", ms * 1e3f, g_stamps.label[i]);
    if (off >= buf_bytes) break;
  }
  return g_stamps.n;
}

}  // extern "C"
