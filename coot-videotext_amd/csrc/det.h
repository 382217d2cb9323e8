// Deterministic accumulation (coot_det_configure; the reference tests run-to-run determinism: tests_nntrainer/integration_deter.py:18-66).
//
// Everything in this library is computed in a fixed order EXCEPT the few places where several workgroups add into one fp32 word with
// atomicAdd (bias / LayerNorm-parameter gradients of the global networks' single-launch backward, the bias gradients the weight-
// gradient GEMM's loader waves collect per split, the pooling bias gradient, the folded input-LayerNorm gradients, the per-op path's
// column sums, the cycle-consistency loss word): float addition does not associate, so the arrival order of the atomics moves the last
// bits, and Adam (eps = 1e-8: an update is +-lr whatever the gradient's size) turns last bits into different trajectories.
// Deterministic mode gives every REGISTERED fp32 range (the gradient arenas, the loss words) a shadow of 64-bit fixed-point
// accumulators (2^-40 units): acc_add() adds round(v 2^40) there with an INTEGER atomic — integer addition associates, the sum does not
// depend on the order — and coot_det_flush adds the sums into the fp32 words (one float addition per word, in stream order behind every
// adder) and clears the shadow.  Addresses outside the registered ranges, and everything while the mode is off, take the plain float
// atomic.  Resolution 9.1e-13 absolute, range +-8.4e6 per word: below fp32 round-off for any gradient entry above 1e-5 in magnitude.
// An addend that is NaN, Inf or >= 2^22 in magnitude bypasses the shadow (plain float atomic on the fp32 word): a diverged step shows
// up as NaN / Inf in the gradients and the loss exactly as without the mode (the run is no longer bit-reproducible then — it is lost anyway).
// Every such addend is COUNTED (a word in the shadow's padding; coot_get_option("det_bypasses") reads it, synchronising): a caller
// that relies on bit-reproducibility checks that the count stayed 0 (RetrievalTrainer.det_bypass_count).
#pragma once
#include "common.h"

namespace coot {

struct DetRange { const char* base; size_t bytes; long long* shadow; };
struct DetTable { int n; DetRange r[8]; unsigned* bypass; };  // bypass: counts the addends that left the fixed-point path (below), in the shadow's padding
constexpr double kDetScale = 1099511627776.0;  // 2^40
constexpr float kDetMaxAddend = 4194304.0f;    // 2^22: one addend; the 64-bit word itself holds sums up to +-8.4e6

// one table per translation unit (the library is built without relocatable device code): COOT_DET_DEFINE_SETTER(name) defines
// det_set_table_<name>(), det.hip installs the same table in all of them
static __device__ DetTable g_det_dev = {0, {}, nullptr};

__device__ __forceinline__ void acc_add(float* p, float v) {
  const int n = g_det_dev.n;
  for (int i = 0; i < n; ++i) {
    const size_t off = (size_t)(reinterpret_cast<const char*>(p) - g_det_dev.r[i].base);  // (wraps to a huge value below the base)
    if (off < g_det_dev.r[i].bytes) {
      // outside the fixed-point range (|v| >= 2^22: a single addend that large is an exploding gradient; words that accumulate up to
      // +-8.4e6 stay exact) or not finite: llrint would be undefined (LLONG_MIN in practice — a finite garbage word that hides the
      // NaN from the trainer's loss checks).  Such an addend takes the plain float atomic: NaN / Inf reach the fp32 word.
      if (!(fabsf(v) < kDetMaxAddend)) {  // counted: a run that took this exit is no longer bit-reproducible (coot_get_option("det_bypasses"))
        atomicAdd(g_det_dev.bypass, 1u);
        break;
      }
      atomicAdd(reinterpret_cast<unsigned long long*>(g_det_dev.r[i].shadow + (off >> 2)), (unsigned long long)llrint((double)v * kDetScale));
      return;
    }
  }
  atomicAdd(p, v);
}

#define COOT_DET_DEFINE_SETTER(name)                                                                              \
  int det_set_table_##name(const DetTable& t) {                                                                   \
    return check_hip(hipMemcpyToSymbol(HIP_SYMBOL(g_det_dev), &t, sizeof(DetTable)), "det table (" #name ")");   \
  }

int det_set_table_fused(const DetTable& t);
int det_set_table_gemm(const DetTable& t);
int det_set_table_loss(const DetTable& t);
int det_set_table_pool(const DetTable& t);
int det_set_table_rowops(const DetTable& t);
// adds the shadow sums of the registered words inside [base, base + bytes) into them and clears the shadow (no-op while the mode is off)
int det_flush_range(const void* base, size_t bytes, hipStream_t st);
bool det_on();
int det_bypass_count();

}  // namespace coot
