// bf16 MFMA GEMMs for the COOT hot path (gfx950).
//   gemm_nt : C[M,N]  = epilogue( alpha * X[M,K] . W[N,K]^T )         (all Linear fwd + dX)
//   gemm_tn : C[Mo,No] += alpha * sum_t A[t,Mo]^T . B[t,No]            (all weight grads, split over t)
#pragma once
#include "common.h"

namespace coot {

struct GemmEpi {
  const float* bias = nullptr;  // [N]
  float alpha = 1.0f;
  int act = 0;  // 0 none | 1 v=gelu(v) | 2 v*=gelu'(aux)
  const bf16_t* aux = nullptr; long ldaux = 0;     // pre-activation for act==2
  bf16_t* save_pre = nullptr; long ldpre = 0;      // store v before activation (bf16)
  const bf16_t* res = nullptr; long ldres = 0;     // + residual (bf16)
  const float* res32 = nullptr; long ldres32 = 0;  // + residual (fp32)
  const float* pe = nullptr; int pe_L = 1;         // + pe[pos * N + col], pos = row % pe_L (rows < pe_T0)
  int pe_T0 = 0x7fffffff; int pe_L2 = 1;           //                      pos = (row - pe_T0) % pe_L2 (second segment)
  const float* rowscale = nullptr; const bf16_t* diag_src = nullptr; long lddiag = 0;  // + rowscale[row]*diag_src[row][col]
  float* colsum = nullptr;                         // += column sums of the final value (bias gradients)
  float* colsum_ws = nullptr; long ld_colsum_ws = 0;  // internal: per-row-block partials (set by the launcher)
  // dropout applied to (alpha*acc + bias) [act 0/1] or to the final product [act 2]
  unsigned drop_thr = 0; float drop_inv_keep = 1.0f; unsigned long long drop_seed = 0; unsigned drop_site = 0;
  const unsigned long long* drop_seed_ptr = nullptr;
  long drop_ld = 0;  // element index = row * drop_ld + col (+ z * drop_zoff)
  void* out = nullptr; long ldc = 0; int out_f32 = 0; int accumulate = 0;
};

struct GemmNT {
  const bf16_t* X = nullptr; long ldx = 0;  // [M,K] tokens
  const bf16_t* W = nullptr; long ldw = 0;  // [N,K] features
  int M = 0, N = 0, K = 0;
  const int* M_dev = nullptr;  // optional device-side row count (<= M): tiles beyond it exit
  // grouped launch: blockIdx.z = g adds these element offsets
  int groups = 1; long zX = 0, zW = 0, zOut = 0;  // zOut applies to out/aux/save_pre/res/bias/colsum columns
  int xcd_order = 1;  // set by the launcher (coot_set_option("xcd_order", 0/1))
  GemmEpi epi;
};

int launch_gemm_nt(const GemmNT& g, hipStream_t stream);
// Hook called with the weight operand of every launch_gemm_nt() of this thread before the launch (nullptr: none): the owner of
// the weight packs uses it to complete layouts it packs lazily (api.hip: per-op layouts of networks that run on the fused
// kernels).  Non-zero return aborts the launch.
typedef int (*gemm_weight_hook_t)(const void* W, hipStream_t stream);
void set_gemm_weight_hook(gemm_weight_hook_t fn);

struct GemmTN {
  const bf16_t* A = nullptr; long lda = 0;  // [T, Mo]
  const bf16_t* B = nullptr; long ldb = 0;  // [T, No]
  int T = 0, Mo = 0, No = 0;
  float alpha = 1.0f;
  float* C = nullptr; long ldc = 0;  // fp32 [Mo, No], accumulated with atomics
  int groups = 1; long zA = 0, zB = 0, zC = 0;
  float* ws = nullptr; size_t ws_floats = 0;  // optional split-reduction workspace (else the thread's default / atomics)
  // optional: a_colsum[m] += sum_t A[t][m] (the bias gradient that goes with dW = dY^T X: the kernel streams dY anyway).
  // groups must be 1; added with one fp32 atomic per column per split.
  float* a_colsum = nullptr;
  int overwrite = 0;  // C = alpha * (...) instead of += (saves the caller a zero fill; groups == 1)
  // microbenchmarks: tile (0, 0) of split 0 adds its shader-clock time per phase of the k loop (wide tiles only):
  // [0] k-steps, [1] issue of the operand loads, [2] LDS reads + MFMA, [3] wait for the loads + LDS stores, [4] barrier, [5] total
  unsigned long long* stamps = nullptr;
};

int launch_gemm_tn(const GemmTN& g, hipStream_t stream);
void set_tn_force_overwrite(bool on);
int tn_debug_xcd_map(int n, const int* gx, const int* gy, const int* groups, const int* splits, int* out_item, int* out_local, int max_blocks);  // every problem of this thread writes C instead of adding to it (gemm.hip)
size_t gemm_tn_workspace_floats(int T, int Mo, int No, int groups);
// Batching: between tn_batch_begin() and tn_batch_flush() every launch_gemm_tn() (without an explicit workspace) is only
// recorded; flush launches all recorded problems as ONE kernel (+ one split reduction) on `stream`.  The operands must
// stay untouched until the flush.  tn_batch_end() leaves the collecting mode.
void tn_batch_begin();
void get_tn_default_workspace(float** ws, size_t* floats);
int tn_batch_flush(hipStream_t stream, bool take_colsums = false);  // take_colsums: the thread's deferred column sums (rowops.h) run inside the reduce launch
void tn_batch_end();
// 1: wide tiles fed by LDS-DMA (gemm_tn_dma_kernel); 0: register-staged (A/B switch)
void set_tn_target_wgs(int n);
void set_tn_dma(int on);
int get_tn_dma();
void set_xcd_order(int bits);  // XCD-aware workgroup -> tile order: 1 = gemm_nt, 4 = short attention (A/B switch)
int get_xcd_order();
// default workspace used by launch_gemm_tn when GemmTN::ws is null (set by the orchestrator for one call)
void set_tn_default_workspace(float* ws, size_t floats);

// per-launch HIP-event timing of every MFMA GEMM kernel (bench roofline leg); events are recorded on the launch stream
enum { TIMING_NT = 2, TIMING_NT_SMALL = 3, TIMING_TN = 4, TIMING_FUSED = 5, TIMING_GLOB = 6, TIMING_INLN = 7 };  // 6: single-launch global network passes; 7: input LayerNorm (the `flops` field carries algorithmic BYTES: the HBM roofline of the input stream)
void gemm_timing_enable(int on);
int gemm_timing_collect(int selector, double* ms, double* flops, int* launches);
void* timing_begin(int kind, double flops, int big_k, hipStream_t stream);  // nullptr when timing is off
void timing_end(void* slot, hipStream_t stream);

// 0 = ds_read_b64_tr_b16 fragments (default), 1 = transposing LDS stores (fallback)
void set_tn_mode(int mode);
int get_tn_mode();

}  // namespace coot
