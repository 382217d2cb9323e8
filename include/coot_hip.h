/*
 * coot_hip.h — C ABI of libcoot_hip.so: the MI355X (gfx950) implementation of the COOT retrieval
 * training hot path.  Plain pointers and sizes only; every pointer is DEVICE memory unless stated
 * otherwise; every call enqueues asynchronously on the given hipStream_t (pass
 * torch.cuda.current_stream().cuda_stream) and never allocates, frees or synchronises.  Return value: 0 = ok,
 * negative = error (message via coot_last_error()).
 *
 * Retained pointers (ownership contract).  A compute call keeps NO pointer of its arguments after it returns (the
 * launches it enqueued read them, in stream order: the caller keeps the memory alive until those launches ran, as with any
 * asynchronous call).  The ONLY entry points that store a caller pointer beyond their own return are these setters; the memory
 * stays the caller's and must outlive the registration:
 *   coot_step_set_device_state(state)         thread-local; read by every later coot_train_step of the thread (and by every
 *                                             replay of a step captured while it was set) until reset with NULL
 *   coot_step_set_input_stages(s0, s1, n)     thread-local; written / read by steps that carry COOT_STEP_INPUT_STAGES or
 *                                             COOT_FWD_INPUT_STAGES until reset with (NULL, NULL, 0); a reset also forgets which
 *                                             stage holds which batch
 *   coot_step_set_next_batch(next, dims)      thread-local, ONE-SHOT: the two structs are COPIED at the call, the feature
 *                                             pointers inside them are read by the next coot_train_step / coot_step_forward of the
 *                                             thread (which consumes the announcement) and by nothing after it; (NULL, NULL) drops
 *                                             a pending announcement
 *   coot_step_set_global_done_events(ev, ev)  thread-local hipEvent_t handles, recorded by every later coot_step_backward of the
 *                                             thread until reset with (NULL, NULL)
 *   coot_step_set_cycle_indices(idx)          thread-local; read at every later coot_train_step / cycle-consistency phase of the
 *                                             thread until reset with NULL
 *   coot_det_configure(n, bases, bytes, shadow, ...)  PROCESS-GLOBAL; the base / size arrays are copied, the ranges and the shadow
 *                                             are consulted by every accumulating kernel of every thread until coot_det_configure(0,
 *                                             ...), which synchronises the device before it returns
 * A caller that frees such memory resets the registration first (RetrievalTrainer.close() does; tests/conftest.py does between
 * modules).  A captured step (hipStreamBeginCapture around coot_train_step) additionally bakes every argument pointer and the
 * registrations above into its nodes: all of them must stay valid for as long as the graph may be replayed.
 *
 * The reference (simon-ging/coot-videotext) is pure Python/PyTorch and has no FFI for this path;
 * each entry point names the reference code it replaces (file:line relative to the reference
 * root) — INTEGRATION.md shows the ctypes binding a maintainer adds.
 */
#ifndef COOT_HIP_H
#define COOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared between this push and the pop at the end of
 * the header leave libcoot_hip.so (tests/test_cpu_host.py checks the dynamic symbol table against this file). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef void* coot_stream_t; /* hipStream_t */

/* One COOT network = nntrainer/models/transformer_legacy.py:26-97 (TransformerConfig) restricted to
 * the options the shipped YAMLs use. */
typedef struct coot_net_config {
  int input_dim;      /* feature dim fed to norm_input                                    */
  int hidden_dim;     /* selfatn_config.hidden_dim (d_model)                              */
  int num_heads;      /* selfatn_config.num_heads                                         */
  int ff_dim;         /* selfatn_config.pointwise_ff_dim (0 => hidden_dim)                */
  int num_layers;     /* selfatn_config.num_layers                                        */
  int use_input_fc;   /* input_fc: Linear(input_dim, hidden_dim) + GELU                   */
  int use_context;    /* cross-attention context block (global nets)                      */
  int ctx_num_layers; /* crossatn_config.num_layers                                       */
  int pooler;         /* 0 = "atn" (GenPool), 1 = "avg_special" (TemporalAvgPool)         */
  int pool_hidden;    /* pooler_config.hidden_dim (0 => hidden_dim)                       */
  int pool_heads;     /* pooler_config.num_heads                                          */
  float dropout;      /* selfatn_config.dropout   (used only when train != 0)             */
  float ctx_dropout;  /* crossatn_config.dropout                                          */
  float pool_dropout; /* pooler_config.dropout                                            */
  int dtype;          /* COOT_DTYPE_BF16 (0, default): bf16 MFMA operands, fp32 accumulation / statistics — the fast path.
                         COOT_DTYPE_F32: the fp32 REFERENCE MODE of coot_net_fwd / coot_net_bwd — the reference's op sequence
                         (nntrainer/models/transformer_legacy.py:200-288, eval mode) with every activation, weight and
                         accumulation in fp32, exact erf GELU, nothing fused or folded: agrees with the reference to ~1e-6 of the
                         output scale, so a difference between the two modes is bf16 rounding and a difference to the reference in
                         F32 mode is a logic error.  coot_net_bwd in this mode is the derivative of exactly that sequence in fp32
                         (parameter gradients to ~1e-5 of the reference's).  A checker: eval only (train must be 0), never what
                         bench.py times.  `saved` keeps every intermediate of the forward (coot_net_saved_bytes accounts for it),
                         `scratch` the backward's temporaries (coot_net_scratch_bytes); wpack is not read.                       */
} coot_net_config;
#define COOT_DTYPE_BF16 0
#define COOT_DTYPE_F32 1
#define COOT_DTYPE_F16 2 /* IEEE half operands: the SECOND BUILD of these sources, libcoot_hip_f16.so (csrc/build.sh, -DCOOT_OPERAND_F16) — the
                            arithmetic of the reference's GPU path (fp16 autocast, coot/trainer_retrieval.py:264; BASELINE.json configs[3]).  The
                            16-bit operand format is a property of the build: libcoot_hip.so accepts COOT_DTYPE_BF16 (and _F32),
                            libcoot_hip_f16.so COOT_DTYPE_F16 (and _F32); coot_get_option("operand_f16") tells which one is loaded.  The f16
                            build is FORWARD-ONLY (coot_net_fwd, coot_step_forward, the loss forward): the reference trains its fp16 path under a
                            GradScaler (coot/trainer_retrieval.py:277-285), which is not built; its backward entry points refuse.            */

const char* coot_last_error(void);
/* ABI version of this header: struct layouts (coot_net_config gained `dtype`, coot_step_buffers `decay_block_all` in round 4) and the
 * meaning of flag words (coot_step_update's `repack` is a bit mask since round 4).  coot_version() returns the value the library was
 * built with; a binding compares the two when it loads the library (coot-videotext_amd/lib.py does) and refuses a mismatch. */
#define COOT_ABI_VERSION 7
int coot_version(void);
/* Option switches for A/B measurements and tests ("fused", "packed", "tn_dma", "grad_poison", ...: the names coot_set_option
 * accepts are listed in csrc/api.hip).  They are PROCESS-GLOBAL ints read at launch time without synchronisation: set them
 * while no other host thread is inside the library (one host thread per process and device is the supported model, SURVEY
 * 8b); the error string, coot_net_grads_overwrite, the step's device-state pointer and the injected cycle indices are
 * thread-local. */
int coot_set_option(const char* name, int value);
/* current value of "tn_dma" / "xcd_order" / "tn_mode" (tests restore what they switch) */
int coot_get_option(const char* name, int* value);
/* profiling aid: device buffer (>= 64 x uint64) receiving s_memtime stamps of block 0 of the fused chain kernels (NULL = off) */
int coot_debug_timestamps(void* dev_u64);
/* coot_set_option("step_stamps", 1): coot_train_step records HIP events at its phase boundaries; this call synchronises
 * the device and writes one line per boundary (microseconds since the start of the last step) into buf.  Profiling aid. */
int coot_debug_step_stamps(char* buf, int buf_bytes);
/* host-only (tests): the keep-scales (0 or 1 / keep-probability) the kernels draw for the n consecutive elements idx0, idx0 + 1, ...
 * of an element-wise dropout site (idx = token row * row width + column), resp. for the Lk attention probabilities of mask row
 * row32 = (sequence * heads + head) * Lq + query — evaluated on the host by the SAME functions (csrc/common.h) and the same
 * quantisation of p the device code uses.  out_host: HOST memory.  Pins oracle/dropout_masks.py (the masks injected into the
 * reference for train-mode parity; nn.Dropout sites of nntrainer/models/transformer_legacy.py:418,435,487,553,592-598 and
 * nntrainer/models/poolers.py:139-143) to this library. */
int coot_debug_dropout_scales(uint64_t seed, unsigned site, uint64_t idx0, int64_t n, float p, float* out_host);
int coot_debug_attn_dropout_scales(uint64_t seed, unsigned site, unsigned row32, int Lk, float p, float* out_host);

/* ---- parameter layout (flat fp32 arena per network; gradients use the same layout) -----------
 * Names and shapes are the reference state-dict names, e.g.
 * "tf.encoder_layers.0.self_attention_layer.sublayer.query_projection.weight". */
int64_t coot_net_param_numel(const coot_net_config* cfg);
int coot_net_param_count(const coot_net_config* cfg);
int coot_net_param_info(const coot_net_config* cfg, int index, char* name, int name_len, int64_t* offset,
                        int64_t shape[4], int* ndim);
int coot_net_out_dim(const coot_net_config* cfg);

/* ---- bf16 weight pack (once per optimizer step) ---------------------------------------------- */
size_t coot_net_wpack_bytes(const coot_net_config* cfg);
int coot_net_pack_weights(const coot_net_config* cfg, const float* params, void* wpack, coot_stream_t stream);
/* The same for n networks in ONE launch (the train step rebuilds the packs of a side's local + global network right after
 * their update: one launch instead of two dependent ones at the end of the step). */
int coot_nets_pack_weights(int n, const coot_net_config* const* cfgs, const float* const* params, void* const* wpacks,
                           coot_stream_t stream);

/* ---- one network forward / backward -----------------------------------------------------------
 * Replaces TransformerLegacy.forward (nntrainer/models/transformer_legacy.py:200-288) and its
 * autograd backward.  feats [N, L, input_dim] fp32 zero padded, lengths [N] int64 (valid rows),
 * hidden [N, hidden_dim] fp32 or NULL, pe = embedding.pe [>=L, hidden_dim] fp32.
 * pooled [N, out_dim] fp32; per_token [N, L, hidden_dim] fp32 or NULL.
 * `saved` carries activations from fwd to bwd (coot_net_saved_bytes), `scratch` is temporary
 * (coot_net_scratch_bytes).  train != 0 enables dropout with the given seed. */
/* Packed (variable-length) token rows, SURVEY 8f-2.  The reference pads every sequence of a batch to the batch maximum
 * (coot/dataset_retrieval.py:335-463) and runs all padded frames / words through the local networks.  With `packed` set, a local
 * network (input_fc + atn pooler) processes only the valid tokens: cu_seqlens[i] (DEVICE, int32, N + N2 + 1 entries, segment 1's
 * sequences first, cu_seqlens[0] = 0) is the first packed row of sequence i, total_tokens = cu_seqlens[N + N2] (HOST copy: it sizes
 * the launches).  The input stays the reference's padded feats (+ lengths): the input LayerNorm gathers the valid rows, the
 * token-tile chains are row independent, attention / pooling / positional encoding take the sequence boundaries from cu_seqlens.
 * Pooled outputs and all gradients equal the padded computation (padded rows carry exactly zero pooling weight, poolers.py:190).
 * NULL (or a network / size the packed path does not cover): the padded layout.  Forward and backward take the same decision.
 * source: where the token rows come from.  COOT_SOURCE_PADDED (0): feats / feats2 are the reference's zero-padded fp32 tensors
 * (coot/dataset_retrieval.py:335-463), the input LayerNorm gathers the valid rows.  COOT_SOURCE_PACKED_F32 / _BF16: `feats` IS the
 * packed matrix [total_tokens, input_dim] (rows in cu_seqlens order, fp32 resp. bf16 bits; feats2 is ignored) as
 * coot_collate_packed writes it — no padding ever crosses PCIe or is read from HBM.  A packed source needs the packed path (it
 * cannot fall back to the padded layout): the call fails where packed rows are not supported. */
#define COOT_SOURCE_PADDED 0
#define COOT_SOURCE_PACKED_F32 1
#define COOT_SOURCE_PACKED_BF16 2
typedef struct coot_packed_seqs { const int32_t* cu_seqlens; int total_tokens; int source; } coot_packed_seqs;
size_t coot_net_saved_bytes(const coot_net_config* cfg, int N, int L, int N2, int L2);
size_t coot_net_scratch_bytes(const coot_net_config* cfg, int N, int L, int N2, int L2);
/* Optional SECOND SEGMENT (feats2 [N2, L2, input_dim], lengths2 [N2]; N2 = 0 / NULL when unused): a second set
 * of sequences pushed through the same network in the same call (the reference calls net_video_local twice
 * per step, coot/model_retrieval.py:104 and :120).  pooled then holds N + N2 rows (segment 1 first). */
int coot_net_fwd(const coot_net_config* cfg, const float* params, const void* wpack, const float* pe,
                 const float* feats, const int64_t* lengths, int N, int L, const float* feats2,
                 const int64_t* lengths2, int N2, int L2, const float* hidden,
                 float* pooled, float* per_token, void* saved, size_t saved_bytes, void* scratch,
                 size_t scratch_bytes, int train, uint64_t seed, const uint64_t* seed_dev, coot_stream_t stream,
                 const coot_packed_seqs* packed /* NULL: padded rows */);
/* Dropout (train != 0) draws from a counter-based generator keyed by (seed + *seed_dev, site, element); seed_dev
 * is an optional DEVICE word the caller advances once per step — with it a captured HIP graph of the step draws new
 * masks on every replay; forward and backward of one step must see the same value.
 * grads: flat fp32 arena, ACCUMULATED (+=).  dhidden [N, hidden_dim] (written) or NULL.
 * dfeats [N, L, input_dim] fp32 (written; only supported when use_input_fc == 0) or NULL. */
int coot_net_bwd(const coot_net_config* cfg, const float* params, const void* wpack, const float* pe,
                 const float* feats, const int64_t* lengths, int N, int L, const float* feats2,
                 const int64_t* lengths2, int N2, int L2, const float* hidden,
                 const float* dpooled, float* grads, float* dhidden, float* dfeats, void* saved,
                 size_t saved_bytes, void* scratch, size_t scratch_bytes, int train, uint64_t seed,
                 const uint64_t* seed_dev, coot_stream_t stream,
                 const coot_packed_seqs* packed /* the forward's */);

/* coot_net_bwd accumulates into `grads`.  coot_net_grads_overwrite(1) (thread-local, until switched off): the weight MATRIX
 * gradients (input FC, QKV / output / feed-forward projections, pooling FCs — each the result of exactly one weight-gradient
 * problem of the pass) are WRITTEN instead; biases and LayerNorm parameters are still accumulated.  coot_nets_zero_grads zeroes
 * the gradient arenas of n networks: all of them (skip_matrices = 0), or only what is still accumulated in that mode
 * (skip_matrices = 1: < 1 % of the arena, one launch).  coot_train_step uses the pair: no 30 MB fill, no read-back of the
 * destination by the weight-gradient launches. */
int coot_net_grads_overwrite(int on);
int coot_nets_zero_grads(int n, const coot_net_config* const* cfgs, float* const* grads, int skip_matrices,
                         coot_stream_t stream);
/* host-only (tests): the (offset, elements) ranges of the arena that mode leaves to the backward; returns their number */
int coot_debug_written_matrices(const coot_net_config* cfg, int64_t* offsets, int64_t* sizes, int max_ranges);
/* the same launch also zeroes n_extra fp32 ranges (extra[i], extra_n[i] floats) */
int coot_nets_zero_grads_ex(int n, const coot_net_config* const* cfgs, float* const* grads, int skip_matrices,
                            float* const* extra, const int64_t* extra_n, int n_extra, coot_stream_t stream);

/* ---- clip -> video packing: the python loop of coot/model_retrieval.py:121-136 ---------------- */
int coot_pack_fwd(const float* emb, const int64_t* counts, int B, int Cmax, int D, float* out /*[B,Cmax,D]*/,
                  uint8_t* mask /*[B,Cmax] 1 = pad*/, int64_t* lens /*[B]*/, coot_stream_t stream);
int coot_pack_bwd(const float* dout, const int64_t* counts, int B, int Cmax, int D, float* demb /* += */,
                  coot_stream_t stream);

/* ---- losses ------------------------------------------------------------------------------------
 * compute_total_constrastive_loss (coot/trainer_retrieval.py:148-182) with ContrastiveLoss
 * (coot/loss_fn.py:63-100): six un-normalised embedding sets, loss accumulated into *loss,
 * gradients accumulated into d* (all NULL => forward only). */
typedef struct coot_contrastive_config {
  float margin;
  float weight_high, weight_high_internal, weight_low, weight_low_internal, weight_context,
      weight_context_internal;
} coot_contrastive_config;
size_t coot_contrastive_scratch_bytes(int n_high, int n_low, int d_high, int d_low);
int coot_contrastive_fwd_bwd(const coot_contrastive_config* cfg, int n_high, int n_low, int d_high, int d_low,
                             const float* vid_emb, const float* par_emb, const float* clip_emb,
                             const float* sent_emb, const float* vid_ctx, const float* par_ctx, float* loss,
                             float* d_vid_emb, float* d_par_emb, float* d_clip_emb, float* d_sent_emb,
                             float* d_vid_ctx, float* d_par_ctx, void* scratch, size_t scratch_bytes,
                             coot_stream_t stream);
/* The same, restricted to a part of the seven terms: COOT_CONTRASTIVE_GLOBAL = the terms on (vid_emb, par_emb) — the outputs of
 * the global networks (trainer_retrieval.py:168-171); COOT_CONTRASTIVE_LOCAL = the terms on (clip_emb, sent_emb) and
 * (vid_ctx, par_ctx) (:172-182), which need the local networks only.  Two calls with the two parts on the same scratch buffer
 * add up to the full call (disjoint scratch regions and gradient outputs, *loss added atomically) and may run on different
 * streams: coot_train_step computes the local part on the text stream next to the video side's global backward (which does not
 * read its gradients).  A part whose sets all fit the LDS (<= 128 rows, (rows + 16) (d + 8) bf16 <= 150 KB: the global part at the
 * paper shapes) takes two launches instead of three, with bit-identical results (coot_set_option("cl_small", 0): three). */
#define COOT_CONTRASTIVE_GLOBAL 1
#define COOT_CONTRASTIVE_LOCAL 2
int coot_contrastive_fwd_bwd_part(const coot_contrastive_config* cfg, int n_high, int n_low, int d_high, int d_low,
                                  const float* vid_emb, const float* par_emb, const float* clip_emb,
                                  const float* sent_emb, const float* vid_ctx, const float* par_ctx, float* loss,
                                  float* d_vid_emb, float* d_par_emb, float* d_clip_emb, float* d_sent_emb,
                                  float* d_vid_ctx, float* d_par_ctx, void* scratch, size_t scratch_bytes, int part,
                                  coot_stream_t stream);

/* fp32 REFERENCE MODE of coot_contrastive_fwd_bwd (the COOT_DTYPE_F32 of the losses; csrc/loss_f32.hip): the same arguments and
 * semantics — F.normalize, the seven ContrastiveLoss terms of coot/trainer_retrieval.py:148-182 / coot/loss_fn.py:63-100, *loss and
 * the six gradients ACCUMULATED — computed by plain fp32 FMA kernels in a fixed summation order: no MFMA, no bf16 operand, no
 * atomics.  With coot_net_config.dtype = COOT_DTYPE_F32 networks around it (and coot_cyclecons_fwd_bwd, which is fp32 VALU code in
 * both modes) the whole forward + loss + backward runs in the library in fp32: every parameter gradient of the reference's eval
 * fixtures to <= 1e-4 relative (tests/test_gpu_f32_mode.py).  A checker: one GPU, the full batch, never what bench.py times.
 * scratch: coot_contrastive_f32_scratch_bytes (NOT coot_contrastive_scratch_bytes). */
size_t coot_contrastive_f32_scratch_bytes(int n_high, int n_low, int d_high, int d_low);
int coot_contrastive_fwd_bwd_f32(const coot_contrastive_config* cfg, int n_high, int n_low, int d_high, int d_low,
                                 const float* vid_emb, const float* par_emb, const float* clip_emb,
                                 const float* sent_emb, const float* vid_ctx, const float* par_ctx, float* loss,
                                 float* d_vid_emb, float* d_par_emb, float* d_clip_emb, float* d_sent_emb,
                                 float* d_vid_ctx, float* d_par_ctx, void* scratch, size_t scratch_bytes,
                                 coot_stream_t stream);

/* The same loss for data-parallel training: sets[i] are the six GATHERED sets (row stride ld[i] floats: they may be column
 * slices of the all-gather buffers), the loss is the mean over the global batch — *loss receives THIS RANK'S SHARE of it (the
 * hinge terms of its rows against every column: the caller adds the shares of all ranks, e.g. in the gradient all-reduce;
 * scoring only its own strips keeps the per-rank cost linear in the global batch); gradients are produced only for this
 * rank's rows — [own_high0, own_high0 + own_high) of the per-video sets (0 vid_emb, 1 par_emb, 4 vid_ctx, 5 par_ctx),
 * [own_low0, own_low0 + own_low) of the per-clip sets (2 clip_emb, 3 sent_emb) — accumulated into the compact arrays
 * d_own[i] [own rows, d]. */
int coot_contrastive_fwd_bwd_dp(const coot_contrastive_config* cfg, int n_high, int n_low, int d_high, int d_low,
                                const float* const sets[6], const int64_t ld[6], float* loss, float* const d_own[6],
                                int own_high0, int own_high, int own_low0, int own_low, void* scratch,
                                size_t scratch_bytes, coot_stream_t stream);

/* The same, reading the rows straight from the BLOCKS of ONE all-gather (no repacking between the collective and the loss): every
 * rank contributes one block holding its six sets; set s of rank r has counts_high[r] (sets 0, 1, 4, 5) or counts_low[r]
 * (sets 2, 3) rows of stride ld[s] floats and starts at blocks + set_base[s * world + r] (floats, multiples of 4).  The global
 * batch is the concatenation of the ranks' rows in rank order; gradients for rank `rank`'s rows into d_own[s] [own rows, d];
 * *loss receives this rank's share.  world <= COOT_DP_MAX_RANKS (the block table travels in the kernel arguments).
 * counts_* / set_base / ld are host arrays.  Replaces the gather + concatenation of nn.DataParallel (nntrainer/trainer_base.py:126-129). */
#define COOT_DP_MAX_RANKS 16
int coot_contrastive_fwd_bwd_dp_blocks(const coot_contrastive_config* cfg, int world, int rank, const int64_t* counts_high,
                                       const int64_t* counts_low, int d_high, int d_low, const float* blocks,
                                       const int64_t* set_base, const int64_t ld[6], float* loss, float* const d_own[6],
                                       void* scratch, size_t scratch_bytes, coot_stream_t stream);

/* CycleConsistencyLoss.forward + get_total_loss(num_samples=1) (coot/loss_fn.py:143-319).
 * idx_* are the th.multinomial draws (one valid position per video).  loss += weight *
 * inv_batch * sum_b (l_clip[b, idx_clip[b]] + l_sent[b, idx_sent[b]]); rows_* optional [B, C]. */
int coot_cyclecons_fwd_bwd(const float* clip, const float* sent, const int64_t* clip_lens,
                           const int64_t* sent_lens, const int64_t* idx_clip, const int64_t* idx_sent, int B,
                           int Cc, int Cs, int D, float weight, float inv_batch, float* loss, float* rows_clip,
                           float* rows_sent, float* dclip, float* dsent, coot_stream_t stream);

/* ---- input side: batch collation into a staging arena ---------------------------------------------
 * One feature level of RetrievalDataset.collate_fn (coot/dataset_retrieval.py:335-463: the zero-padded tensor + the bool
 * mask the loops at :362-364, :378-380, :404-414, :438-452 fill): n sequences, seq[i] -> fp32 [rows[i], dim], are written to
 * dst [n, max_rows, dim] (fp32 copy, or bf16 round-to-nearest-even when dst_bf16 != 0), rows beyond rows[i] zeroed;
 * mask [n, max_rows] bytes (optional): 0 = data, 1 = padding.  Host-only: dst is normally a slice of ONE pinned arena that
 * carries all four levels, lengths and masks and goes to the device with one hipMemcpyAsync on a copy stream.  threads > 1
 * splits the sequences over that many host threads. */
int coot_collate_level(const float* const* seq, const int64_t* rows, int64_t n, int64_t dim, int64_t max_rows, int dst_bf16,
                       void* dst, uint8_t* mask, int threads);
/* Packed at the source ("produce packed varlen features directly in pinned memory, drop the zero padding"): the same n
 * sequences written BACK TO BACK, dst [sum rows, dim] (fp32 copy or bf16 round-to-nearest-even), and their row starts
 * cu_seqlens [n + 1] (int32, cu_seqlens[0] = 0) — exactly what coot_packed_seqs / coot_step_batch.cu_vis take, so one call per side
 * (the B videos followed by the Nc clips; the B paragraphs followed by the Nc sentences) produces the side's input.  Replaces the
 * padded blocks of RetrievalDataset.collate_fn (coot/dataset_retrieval.py:362-364, :378-380, :404-414, :438-452); the padded
 * batch is recoverable bit-exactly (dataset_retrieval.unpack_batch, tests/test_input_pipeline.py).  Host-only. */
int coot_collate_packed(const float* const* seq, const int64_t* rows, int64_t n, int64_t dim, int dst_bf16, void* dst,
                        int32_t* cu_seqlens, int threads);

/* ---- retrieval ranking on the device -----------------------------------------------------------
 * validate_epoch's metric tail (coot/trainer_retrieval.py:397-402, :425-436) + nntrainer/retrieval.py:31-98 for one pair
 * of embedding sets emb1, emb2 [N, d] fp32 (e.g. vid_emb / par_emb of the whole validation set), both directions:
 *   normalize != 0: rows are first divided by sqrt(sum x^2) (the reference's manual normalisation, no eps);
 *   ranks_12[i] = position of i in argsort(d[i])[::-1], d = emb1 . emb2^T (fp32 FMA chains in k order);  ranks_21: the same
 *   for d^T.  Exact ties are ordered as the reversal of a stable ascending sort (j > i ahead of i);
 *   metrics (optional, 14 floats): {r1, r5, r10, r50, medr, meanr, sum} for 1 -> 2, then for 2 -> 1 (VALKEYS order,
 *   R@K as fractions, medr = floor(median) + 1, meanr = mean + 1);
 *   sim_out (optional, [N, N]): the similarity matrix the ranks were counted on (testing aid).
 * The similarity matrix is never materialised otherwise.  workspace: coot_retrieval_workspace_bytes(N, d). */
size_t coot_retrieval_workspace_bytes(int N, int d);
int coot_retrieval_ranks(const float* emb1, const float* emb2, int N, int d, int normalize, int32_t* ranks_12, int32_t* ranks_21,
                         float* metrics, float* sim_out, void* workspace, size_t workspace_bytes, coot_stream_t stream);

/* ---- the whole training step as native code (coot/trainer_retrieval.py:253-291) -------------------------------
 * Networks are indexed 0 = net_video_local, 1 = net_video_global, 2 = net_text_local, 3 = net_text_global
 * (coot/configs_retrieval.py:182-189).  All buffers are caller-owned device memory. */
typedef struct coot_step_config {
  coot_net_config net[4];
  coot_contrastive_config contr;
  float cc_weight;                               /* train.loss_cycle_cons                                    */
  float lr, beta1, beta2, eps, weight_decay;     /* optimizer as built by nntrainer/optimization.py:45-74     */
  int optimizer;                                 /* 0 = torch.optim.Adam, 1 = the in-file RAdam (:79-181)     */
  int radam_degentosgd;                          /* RAdam degenerated_to_sgd (optimizer.radam_degentosgd)     */
} coot_step_config;
typedef struct coot_step_dims {
  int B, Nc, Lv, Lc, Lp, Ls, Cmax_clip, Cmax_sent;
  int tok_vis, tok_txt;   /* packed rows (coot_packed_seqs): valid frames of the B videos + Nc clips, valid words of the B paragraphs +
                             Nc sentences (host values; 0 = padded layout); used with coot_step_batch.cu_vis / cu_txt */
  int source;             /* COOT_SOURCE_*: PADDED = the four padded fp32 tensors of the reference's batch; PACKED_F32 / PACKED_BF16 =
                             coot_step_batch.vid_feat / par_feat point at the packed matrices [tok_vis, Dv] / [tok_txt, Dt] of
                             coot_collate_packed (videos then clips, paragraphs then sentences), clip_feat / sent_feat are unused */
} coot_step_dims;
typedef struct coot_step_buffers {
  float* params[4]; float* grads[4]; void* wpack[4];       /* flat arenas (coot_net_param_info layout), bf16 pack */
  float* adam_m[4]; float* adam_v[4];                      /* Adam moments, same layout                            */
  const float* decay_mask[4];                              /* 1.0 / 0.0 per element (decay_mult), or NULL = all 1  */
  const float* pe[4];                                      /* embedding.pe tables                                  */
  const uint8_t* decay_block_all[4];                       /* optional, one byte per 1 024 consecutive elements of the arena
                                                              (the last block may be partial): non-zero = decay_mask is 1.0 on
                                                              the whole block — the update then does not read the mask there
                                                              (the mask is 0 on the bias vectors only: > 99 % of the blocks;
                                                              30 MB of the step's HBM reads).  NULL = read the mask everywhere */
} coot_step_buffers;
typedef struct coot_step_batch {                           /* RetrievalDataBatchTuple (coot/dataset_retrieval.py:64-102) */
  const float *vid_feat, *clip_feat, *par_feat, *sent_feat;
  const int64_t *vid_len, *clip_len, *par_len, *sent_len, *clip_num, *sent_num;
  const int32_t *cu_vis, *cu_txt;  /* optional packed row starts [B + Nc + 1]: the B videos (paragraphs) first, then the Nc clips
                                      (sentences); NULL = padded layout */
} coot_step_batch;
size_t coot_step_workspace_bytes(const coot_step_config* cfg, const coot_step_dims* dims);
/* One optimisation step in one call: grads zeroed, both sides encoded on side_v / side_t, contrastive +
 * cycle-consistency losses (cycle indices drawn on the device), backward, Adam (`step` is the 1-based step count for the
 * bias correction).  losses[3] = {total, contrastive, cycle-consistency} (device, overwritten).
 * do_optimizer: bit mask of COOT_STEP_*.  side_v may be the same stream as main_s (recommended: the heavier video side then
 * runs without any cross-stream hop); side_t must differ from side_v.  On return main_s is ordered after both sides.
 * The call also uses one internal stream of the calling thread (created on first use): for the next batch's input LayerNorm
 * (COOT_STEP_INPUT_STAGES) and for the update of the GLOBAL networks, which starts behind their backward and runs next to the local
 * backward (sides whose local network has >= 8192 rows; not in deterministic mode, not in a captured step); side_v / side_t are
 * ordered after it before the call's last launches, so the ordering guarantees above hold unchanged. */
#define COOT_STEP_OPTIMIZER 1   /* Adam update of all four networks                                                          */
#define COOT_STEP_REPACK 2      /* rebuild the bf16 weight packs right after the update (off the next step's critical path)  */
#define COOT_STEP_PACKS_FRESH 4 /* wpack[] is current (previous step ran with REPACK and nothing else touched the parameters):
                                   skip the packing at the start of the step                                                */
#define COOT_STEP_INPUT_STAGES 16 /* x^ of the input LayerNorm lives in the caller's input stages (coot_step_set_input_stages below)      */
#define COOT_STEP_STAGE_ANNOUNCED 32 /* the caller asserts that `batch` IS the batch it announced with coot_step_set_next_batch and that its
                                   contents have not changed since: only then may the step use the x^ a previous step prepared.  Without the
                                   bit a prepared stage is never used, whatever the pointers say (a loader that refills a fixed-shape arena
                                   in place presents the SAME pointers and dims with other data)                              */
#define COOT_STEP_DEFER_TEXT_JOIN 8 /* on return main_s is ordered after the VIDEO side only: the text side's tail (its Adam update,
                                   weight packs) is still running on side_t; the three loss words are final on main_s.  The next coot_train_step with the
                                   same streams needs no join (its text side continues on side_t in order, its video side touches
                                   nothing the text tail writes): back-to-back steps overlap that tail (~30 us) with the next
                                   forward.  Before anything else reads the text networks' parameters / packs from another
                                   stream, the CALLER orders that stream after side_t.                                      */
int coot_train_step(const coot_step_config* cfg, const coot_step_buffers* bufs, const coot_step_batch* batch,
                    const coot_step_dims* dims, float* losses, void* workspace, size_t workspace_bytes, int train,
                    uint64_t seed, int64_t step, int do_optimizer, coot_stream_t main_stream, coot_stream_t side_v,
                    coot_stream_t side_t);
/* The same step split in phases, for data parallel training where torch.distributed collectives (all-gather of the
 * embeddings, all-reduce of the gradients) sit between them.  local_* = [B + Nc, D] (context rows first, then
 * clip / sentence embeddings), glob_* = [B, 2D], resh_* = [B, Cmax, D] (packed + zero padded). */
#define COOT_FWD_PACKS_FRESH 1
#define COOT_FWD_INPUT_STAGES 2
#define COOT_FWD_STAGE_ANNOUNCED 4   /* as COOT_STEP_STAGE_ANNOUNCED */
int coot_step_forward(const coot_step_config* cfg, const coot_step_buffers* bufs, const coot_step_batch* batch,
                      const coot_step_dims* dims, float* local_v, float* local_t, float* glob_v, float* glob_t,
                      float* resh_v, float* resh_t, void* workspace, size_t workspace_bytes, int train, uint64_t seed,
                      int packs_fresh /* bit mask: COOT_FWD_PACKS_FRESH = the bf16 weight packs are current (coot_step_update repacked
                                         them): skip the packing; COOT_FWD_INPUT_STAGES = as COOT_STEP_INPUT_STAGES of coot_train_step (the
                                         following coot_step_backward of this thread reads x^ in the same stage) */,
                      coot_stream_t main_stream, coot_stream_t side_v, coot_stream_t side_t);
int coot_step_backward(const coot_step_config* cfg, const coot_step_buffers* bufs, const coot_step_batch* batch,
                       const coot_step_dims* dims, const float* local_v, const float* local_t, const float* resh_v,
                       const float* resh_t, float* d_local_v, float* d_local_t, const float* d_glob_v,
                       const float* d_glob_t, const float* d_resh_v, const float* d_resh_t, void* workspace,
                       size_t workspace_bytes, int train, uint64_t seed, coot_stream_t main_stream, coot_stream_t side_v,
                       coot_stream_t side_t);
/* Replayable step (hipGraph): with a device state block set (coot_step_device_state_bytes() bytes: { uint64 seed; uint64
 * step; float lr; int pad; 32 bytes of optimizer scalars }), coot_train_step reads its per-step scalars from DEVICE memory — its
 * first node advances them (step += 1, seed += 7919, optimizer scalars from step and lr) exactly as the host does between two
 * eager steps; the seed argument becomes a salt (pass 0), the step argument is ignored.  The call can then be captured once
 * (hipStreamBeginCapture on `main`, side_v == main) and replayed: dependent launches cost 1.7 us in a replay against 3.1 us
 * launched one by one (tools/micro/launchgap.hip).  The host initialises seed / step / lr before the first step and rewrites
 * lr when the schedule changes it.  NULL returns to argument-driven steps.  Thread-local. */
size_t coot_step_device_state_bytes(void);
int coot_step_set_device_state(void* state);
/* Data parallel: hipEvent_t handles (or NULL) that coot_step_backward records on the video / text stream as soon as that side's
 * GLOBAL network backward is enqueued — its parameter gradients (networks 1 and 3) are final from there on, so a communication
 * stream can wait on the events and reduce them while the local backward (two thirds of the pass) still runs.  Thread-local,
 * stays set until changed. */
int coot_step_set_global_done_events(void* ev_video, void* ev_text);
/* Stream ordering without system-scope fences.  A default HIP event (what hipEventCreate, torch.cuda.Event and torch's wait_stream
 * use) performs a system-scope release at every record: an L2 writeback / invalidation that makes device memory visible to the host and
 * other devices — and evicts what the step's latency-bound kernels keep in L2.  Ordering two streams of ONE device needs none of it.
 * The library owns COOT_SYNC_EVENTS events created with hipEventDisableTiming | hipEventDisableSystemFence: coot_event_record /
 * coot_event_wait (slot 0 .. COOT_SYNC_EVENTS - 1; coot_event_handle returns the hipEvent_t, e.g. for
 * coot_step_set_global_done_events), coot_stream_hop = everything enqueued on `to` afterwards runs behind everything enqueued on `from`
 * before (internal event ring).  Thread-local, like the streams' owner. */
/* Streams that really run CONCURRENTLY.  HIP multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by
 * default) in the order they are created; two streams that land on one queue run their kernels one after the other whatever the
 * events between them say — with the step's video and text side on one queue the ActivityNet step costs 1.72 instead of 1.22 ms, and
 * one more stream created anywhere in the process before the trainer's is enough to get there (profiles/r06_stream_queues.txt).
 * coot_stream_create_concurrent creates (non-blocking) candidate streams until one's kernels overlap those of every stream in
 * others[0 .. n_others) (n_others <= 8; NULL entries = the null stream): a 100-us spin kernel on both must finish in the time of one.
 * It also keeps clear of the library's own stream (next batch's input LayerNorm, early update) when the queues allow it.
 * priority: 0 normal, > 0 the device's lowest, < 0 its highest.  *concurrent (may be NULL) = 1 if such a stream was found within 32
 * candidates, else 0 with the last candidate returned and a line on stderr (GPU_MAX_HW_QUEUES=1, a tool that serialises the
 * queues).  Release with coot_stream_destroy (synchronises the stream).  coot_streams_overlap(a, b): the test itself — 1 kernels overlap,
 * 0 they run one after the other, < 0 error.  All three SYNCHRONISE THE DEVICE: setup calls, not step calls.   Thread-local bookkeeping.  The library verifies its own stream the same way against (side_v, side_t) at the head of the
 * first coot_train_step / coot_step_forward that brings a new pair (one device synchronisation; never under capture).
 * coot_get_option: stream_overlap_tests / stream_candidates_rejected / stream_unresolved. */
int coot_stream_create_concurrent(const coot_stream_t* others, int n_others, int priority, coot_stream_t* out, int* concurrent);
int coot_stream_destroy(coot_stream_t stream);
int coot_streams_overlap(coot_stream_t a, coot_stream_t b);
#define COOT_SYNC_EVENTS 8
int coot_event_record(int slot, coot_stream_t stream);
int coot_event_wait(int slot, coot_stream_t stream);
void* coot_event_handle(int slot);
int coot_stream_hop(coot_stream_t from, coot_stream_t to);
/* Optimizer update of the four networks after the gradient all-reduce (cfg->optimizer; `step` 1-based): one launch per side on
 * side_v / side_t.  repack: bit mask — COOT_UPDATE_REPACK: the bf16 weight packs are rebuilt so that the next coot_step_forward may skip
 * the packing; COOT_UPDATE_DEFER_TEXT_JOIN: on return main_s is ordered after the VIDEO side only (as COOT_STEP_DEFER_TEXT_JOIN: the
 * text side's update tail overlaps the next step's forward; the caller orders whatever else touches the text networks or the gradient
 * arenas after side_t).
 * losses (may be NULL): the three loss words { total, contrastive, cycle-consistency } of the step; the video side's update launch
 * writes total = contrastive + cycle-consistency (a data-parallel caller keeps the two all-reduced words there: no extra launch). */
#define COOT_UPDATE_REPACK 1
#define COOT_UPDATE_DEFER_TEXT_JOIN 2
#define COOT_UPDATE_SKIP_GLOBAL 4  /* the two global networks are left alone: a COOT_UPDATE_GLOBAL_ONLY call of the same step updated them */
#define COOT_UPDATE_GLOBAL_ONLY 8  /* update (and repack) the two GLOBAL networks only, on main_s alone (side_v / side_t unused, losses untouched):
                                      their gradients are final behind the global backward (data parallel: behind their all-reduce bucket), a whole
                                      local backward before the step's end — as coot_train_step's early update of the global networks         */
int coot_step_update(const coot_step_config* cfg, const coot_step_buffers* bufs, int64_t step, int repack, float* losses,
                     coot_stream_t main_stream, coot_stream_t side_v, coot_stream_t side_t);
/* One valid clip / sentence position per video for the cycle-consistency loss (th.multinomial(mask, 1), coot/loss_fn.py:306-314),
 * drawn on the device: idx[0 .. B) from clip_num, idx[B .. 2B) from sent_num. */
int coot_sample_cycle_indices(const int64_t* clip_num, const int64_t* sent_num, int B, uint64_t seed, int64_t* idx,
                              coot_stream_t stream);
/* Injects the draw instead: idx[0 .. B) clip positions, idx[B .. 2B) sentence positions (device memory, read at every
 * coot_train_step / phase 4 until reset with NULL).  For reproducing a given th.multinomial sequence of the reference
 * (coot/loss_fn.py:306-314 consumes the global torch RNG, which no device kernel can replay).  Thread-local. */
int coot_step_set_cycle_indices(const int64_t* idx);
/* Software-pipelined input LayerNorm (COOT_STEP_INPUT_STAGES).  The input LayerNorm of the local networks
 * (nntrainer/models/transformer_legacy.py:224-239, norm_input) has no parameters of its own here — its gain / bias are folded into the
 * packed input-FC weights — so the normalised features x^ of batch t + 1 depend on nothing step t computes.  With two caller-owned
 * device buffers ("stages", each >= coot_step_input_stage_bytes() of the largest batch) a coot_train_step that carries the flag
 *   - keeps this batch's x^ in one stage (normalising it first unless the previous step already did), and
 *   - if a next batch was announced (coot_step_set_next_batch, one-shot: consumed by that step), normalises THAT batch into the
 *     other stage on an internal stream behind both sides' local forward passes — where the step runs its global networks and losses
 *     on a handful of CUs and the memory system is idle; the next coot_train_step on that batch (same pointers and dims) finds x^
 *     ready and starts with the input FC — provided the caller says so (COOT_STEP_STAGE_ANNOUNCED): equal pointers and dims alone do
 *     not identify a batch (arena slots are refilled in place), so without the bit the step normalises itself, after waiting for the
 *     internal stream's last launch (whose reads of the announced batch are then ordered before anything the caller enqueues next).
 * Every step still executes exactly one input LayerNorm per side (for the following batch); what a data loader's lookahead buys is
 * that it runs off the critical path.  Results are bit-identical to a step without stages.  The state (which stage holds what) is
 * thread-local and reset when the stage pointers change; steps without the flag never touch it.  Not available under graph capture. */
size_t coot_step_input_stage_bytes(const coot_step_config* cfg, const coot_step_dims* dims);
int coot_step_set_input_stages(void* stage0, void* stage1, size_t bytes_each);
int coot_step_set_next_batch(const coot_step_batch* next, const coot_step_dims* next_dims);
/* Deterministic mode (the reference tests run-to-run determinism: tests_nntrainer/integration_deter.py:18-66).  Everything in the
 * library is computed in a fixed order except the fp32 atomicAdds by which several workgroups add into one word (bias / LayerNorm
 * parameter gradients of a few kernels, the cycle-consistency loss word): their arrival order moves the last bits of a gradient, and
 * Adam amplifies last bits into different trajectories.  coot_det_configure registers up to 8 fp32 ranges (the gradient arenas, the
 * loss words) and a caller-owned shadow (coot_det_shadow_bytes) of 64-bit fixed-point accumulators: while it is set, every such add
 * into a registered range goes to the shadow as an INTEGER atomic (order independent) and coot_det_flush — called by coot_train_step /
 * coot_step_backward for the arenas and loss words they were given, by the caller for anything else — adds the sums into the fp32
 * words and clears the shadow.  With it two runs of the same steps are bit-identical.  n = 0 switches the mode off.  PROCESS-GLOBAL
 * (like the option switches); configure synchronises the device. */
size_t coot_det_shadow_bytes(int n, const size_t* bytes);
int coot_det_configure(int n, void* const* bases, const size_t* bytes, void* shadow, size_t shadow_bytes, coot_stream_t stream);
int coot_det_flush(const void* base, size_t bytes, coot_stream_t stream);
int coot_adam_step(float* params, const float* grads, float* m, float* v, const float* decay_mask, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int64_t step, coot_stream_t stream);
/* RAdam of nntrainer/optimization.py:79-181 on one flat arena: decoupled decay weight_decay * decay_mask,
 * rectified update when N_sma >= 5, otherwise SGD-with-momentum (degenerated_to_sgd) or no parameter update. */
int coot_radam_step(float* params, const float* grads, float* m, float* v, const float* decay_mask, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int64_t step, int degenerated_to_sgd,
                    coot_stream_t stream);

/* ---- kernel-level entry points (unit tests / microbenchmarks) ---------------------------------- */
/* C[M,N] (bf16 or fp32) = act(X[M,K] . W[N,K]^T + bias) (+ residual)   (bf16 operands as uint16) */
int coot_gemm_nt(const void* X, int64_t ldx, const void* W, int64_t ldw, int M, int N, int K, const float* bias,
                 int act, const void* residual_bf16, int64_t ldres, void* out, int64_t ldc, int out_f32,
                 coot_stream_t stream);
/* C[Mo,No] fp32 += sum_t A[t,Mo] * B[t,No].  workspace (coot_gemm_tn_workspace_bytes) selects the two-pass
 * split reduction; NULL falls back to fp32 atomics. */
size_t coot_gemm_tn_workspace_bytes(int T, int Mo, int No);
int coot_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int T, int Mo, int No, float* C,
                 int64_t ldc, void* workspace, size_t workspace_bytes, coot_stream_t stream);
/* n weight-gradient problems C_i[Mo_i,No_i] (+)= A_i^T . B_i in ONE launch (+ one split reduction) — the way a network's
 * backward pass issues them (coot_net_bwd); a_colsum (optional) += column sums of A_i (the bias gradient).  workspace: fp32
 * partial tiles, sum_i splits_i * Mo_i * No_i * 4 bytes with splits_i <= 8 (too small: the problems run one by one).
 * stamps (optional, (64 + 2 x workgroups) x uint64 on the device): phase times of tile (0,0) of problem 0 in shader clocks. */
typedef struct coot_tn_problem {
  const void* A; int64_t lda;   /* bf16 [T, Mo] */
  const void* B; int64_t ldb;   /* bf16 [T, No] */
  int T, Mo, No;
  float* C; int64_t ldc;        /* fp32 [Mo, No] */
  float* a_colsum;              /* fp32 [Mo] or NULL (groups == 1) */
  int overwrite;                /* 1: C = ..., 0: C += ... */
  int groups;                   /* >= 1: batch of problems, group z at A + z*zA, B + z*zB, C + z*zC (elements) */
  int64_t zA, zB, zC;
} coot_tn_problem;
int coot_gemm_tn_batch(const coot_tn_problem* problems, int n, void* workspace, size_t workspace_bytes,
                       uint64_t* stamps, coot_stream_t stream);
/* host-only: the workgroup -> (problem, tile) table of such a launch (tiles of one problem, split and group on one XCD: block b
 * runs on XCD b % 8) for n problems with gx x gy tiles, `groups` groups and `splits` splits each; out_item / out_local
 * [max_blocks] (-1 = empty slot); returns the grid size (0: the launch does not fit the table, linear order is used) */
int coot_debug_tn_xcd_map(int n, const int* gx, const int* gy, const int* groups, const int* splits, int* out_item,
                          int* out_local, int max_blocks);
/* one wave writes n pairs (100 MHz real-time counter, shader clock counter), one every interval_ticks real-time ticks:
 * run it on a side stream to see the shader clock the device delivers under the step's load (tools/clock_probe.py) */
int coot_debug_clock_monitor(uint64_t* out, int n, int interval_ticks, coot_stream_t stream);
int coot_ln_fwd(const float* x, int R, int D, const float* gain, const float* bias, void* y_bf16, float* y_f32,
                coot_stream_t stream);
int coot_attn_fwd(const void* qkv, int Nseq, int L, int H, int dh, const int64_t* lens, void* out, float* lse,
                  coot_stream_t stream);
/* HIP-event timing of every gemm_nt launch (the dominant kernel) while enabled; collect() synchronises the
 * recorded events and returns summed duration [ms], algorithmic flops (2*M*N*K) and launch count.
 * only_big_k != 0 restricts to the input-FC instances (K >= 1024). */
int coot_timing_enable(int on);
int coot_timing_collect(int only_big_k, double* ms, double* flops, int* launches);
/* probe of ds_read_b64_tr_b16: out[64*4] = what each lane reads from a 16x64 bf16 LDS tile holding its own index */
int coot_probe_tr16(uint16_t* out, coot_stream_t stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* COOT_HIP_H */
