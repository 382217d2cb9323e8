// Input side of the path: host-side batch collation into a staging arena.
//
// RetrievalDataset.collate_fn (coot/dataset_retrieval.py:335-463) builds four zero-padded fp32 tensors with Python loops of
// tensor slice assignments and hands them to pin_memory / .cuda() one by one.  Here one call per feature level writes the
// padded block (fp32, or bf16 with round-to-nearest-even: half the PCIe bytes) and its mask straight into the caller's
// arena — normally ONE pinned allocation holding all four levels, lengths and masks, moved with ONE hipMemcpyAsync on a copy
// stream (dataset_retrieval.py: BatchArena / DeviceLoader).  Pure host code: no kernel, no stream.
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/coot_hip.h"
#include "common.h"

namespace {

// fp32 -> bf16, round to nearest even; NaN stays NaN (canonical quiet NaN: rounding the payload could carry into infinity).
// Branch-free so that the row loop vectorises; compiled for AVX-512 / AVX2 / baseline and picked at load time (the staging
// thread converts 65 M elements per ActivityNet batch: the scalar loop was the bound of the bf16 staging path).
#if defined(__HIP_DEVICE_COMPILE__)
#define COOT_HOST_CLONES  // hipcc also parses this file for gfx950: host function multiversioning does not exist there
#else
#define COOT_HOST_CLONES __attribute__((target_clones("arch=x86-64-v4", "arch=x86-64-v3", "default")))
#endif
COOT_HOST_CLONES void convert_bf16_rne(const float* __restrict__ s, uint16_t* __restrict__ o, int64_t n) {
#ifdef COOT_OPERAND_F16  // the f16 build stages IEEE half (the name keeps "bf16": the 16-bit operand of the build, common.h)
  for (int64_t k = 0; k < n; ++k) {
    const _Float16 h = (_Float16)s[k];
    std::memcpy(o + k, &h, 2);
  }
  return;
#endif
  for (int64_t k = 0; k < n; ++k) {
    uint32_t u;
    std::memcpy(&u, s + k, 4);
    const uint32_t r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    o[k] = (u & 0x7fffffffu) > 0x7f800000u ? (uint16_t)0x7fc0 : (uint16_t)r;
  }
}

// Persistent host workers.  A batch is ~1-2 ms of copy / conversion work per level; starting 16 std::threads per call cost as much as
// the work they did (tools/bench_input.py: 32 and 64 threads were SLOWER than 16).  The pool is created on first use, grows to the
// largest thread count asked for, and lives until the library is unloaded; run() hands out [0, n) job indices, the caller works too.
class HostPool {
 public:
  ~HostPool() {
    if (pid_ != getpid()) return;  // a forked child: the workers never existed here
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : *workers_) t.join();
  }
  void run(int njobs, const std::function<void(int)>& job) {
    if (njobs <= 1) { if (njobs == 1) job(0); return; }
    std::unique_lock<std::mutex> call(call_mu_);  // one parallel region at a time (collation calls come from one loader thread)
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (pid_ != getpid()) {  // first use after a fork (DataLoader worker processes): threads do not survive it — start over
        workers_ = new std::vector<std::thread>();  // (the parent's std::thread objects are left alone: not joinable from here)
        pid_ = getpid();
      }
      while ((int)workers_->size() < njobs - 1) workers_->emplace_back([this] { loop(); });
      job_ = &job; next_ = 0; njobs_ = njobs; pending_ = njobs; ++epoch_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      int i;
      const std::function<void(int)>* j;
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (!job_ || next_ >= njobs_) return;
        i = next_++; j = job_;
      }
      (*j)(i);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && job_ && next_ < njobs_); });
        if (stop_) return;
        seen = epoch_;
      }
      work();
    }
  }
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread>* workers_ = new std::vector<std::thread>();
  pid_t pid_ = getpid();
  const std::function<void(int)>* job_ = nullptr;
  int next_ = 0, njobs_ = 0, pending_ = 0;
  unsigned long epoch_ = 0;
  bool stop_ = false;
};
HostPool& host_pool() { static HostPool p; return p; }

void collate_range(const float* const* seq, const int64_t* rows, int64_t i0, int64_t i1, int64_t dim, int64_t max_rows, int dst_bf16,
                   void* dst, uint8_t* mask) {
  const size_t esz = dst_bf16 ? 2 : 4;
  for (int64_t i = i0; i < i1; ++i) {
    const int64_t r = rows[i];
    char* d = (char*)dst + (size_t)i * max_rows * dim * esz;
    if (dst_bf16) {
      convert_bf16_rne(seq[i], (uint16_t*)d, r * dim);
    } else if (r > 0) {
      std::memcpy(d, seq[i], (size_t)r * dim * 4);
    }
    std::memset(d + (size_t)r * dim * esz, 0, (size_t)(max_rows - r) * dim * esz);
    if (mask) {
      std::memset(mask + i * max_rows, 0, (size_t)r);
      std::memset(mask + i * max_rows + r, 1, (size_t)(max_rows - r));
    }
  }
}

}  // namespace

extern "C" int coot_collate_level(const float* const* seq, const int64_t* rows, int64_t n, int64_t dim, int64_t max_rows, int dst_bf16,
                                  void* dst, uint8_t* mask, int threads) {
  COOT_REQUIRE(n >= 0 && dim > 0 && max_rows >= 0, "coot_collate_level: bad sizes n=%ld dim=%ld max_rows=%ld", (long)n, (long)dim, (long)max_rows);
  COOT_REQUIRE(n == 0 || (seq && rows && dst), "coot_collate_level: null pointer");
  for (int64_t i = 0; i < n; ++i) {
    COOT_REQUIRE(rows[i] >= 0 && rows[i] <= max_rows, "coot_collate_level: sequence %ld has %ld rows, max_rows is %ld", (long)i, (long)rows[i], (long)max_rows);
    COOT_REQUIRE(rows[i] == 0 || seq[i], "coot_collate_level: sequence %ld is null", (long)i);
  }
  const int64_t bytes = n * max_rows * dim * (dst_bf16 ? 2 : 4);
  int nt = threads < 1 ? 1 : threads;
  if (nt > n) nt = (int)(n > 0 ? n : 1);
  if (bytes < (1 << 20)) nt = 1;  // thread start-up costs more than a 1 MB copy
  if (nt == 1) {
    collate_range(seq, rows, 0, n, dim, max_rows, dst_bf16, dst, mask);
    return 0;
  }
  host_pool().run(nt, [&](int t) { collate_range(seq, rows, n * t / nt, n * (t + 1) / nt, dim, max_rows, dst_bf16, dst, mask); });
  return 0;
}

namespace {
void pack_range(const float* const* seq, const int64_t* rows, const int32_t* cu, int64_t i0, int64_t i1, int64_t dim, int dst_bf16, void* dst) {
  const size_t esz = dst_bf16 ? 2 : 4;
  for (int64_t i = i0; i < i1; ++i) {
    const int64_t r = rows[i];
    if (r <= 0) continue;
    char* d = (char*)dst + (size_t)cu[i] * dim * esz;
    if (dst_bf16) convert_bf16_rne(seq[i], (uint16_t*)d, r * dim);
    else std::memcpy(d, seq[i], (size_t)r * dim * 4);
  }
}
}  // namespace

extern "C" int coot_collate_packed(const float* const* seq, const int64_t* rows, int64_t n, int64_t dim, int dst_bf16, void* dst,
                                   int32_t* cu_seqlens, int threads) {
  COOT_REQUIRE(n >= 0 && dim > 0 && cu_seqlens, "coot_collate_packed: bad arguments n=%ld dim=%ld", (long)n, (long)dim);
  COOT_REQUIRE(n == 0 || (seq && rows && dst), "coot_collate_packed: null pointer");
  int64_t tot = 0;
  cu_seqlens[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    COOT_REQUIRE(rows[i] >= 0 && (rows[i] == 0 || seq[i]), "coot_collate_packed: sequence %ld: %ld rows / null", (long)i, (long)rows[i]);
    tot += rows[i];
    COOT_REQUIRE(tot < (int64_t)1 << 31, "coot_collate_packed: more than 2^31 token rows");
    cu_seqlens[i + 1] = (int32_t)tot;
  }
  int nt = threads < 1 ? 1 : threads;
  if (nt > n) nt = (int)(n > 0 ? n : 1);
  if (tot * dim * (dst_bf16 ? 2 : 4) < (1 << 20)) nt = 1;
  if (nt == 1) { pack_range(seq, rows, cu_seqlens, 0, n, dim, dst_bf16, dst); return 0; }
  // threads take runs of sequences with about the same number of ROWS (sequence lengths are ragged)
  std::vector<int64_t> cut(nt + 1, n);
  cut[0] = 0;
  for (int t = 1; t < nt; ++t) {
    const int64_t want = tot * t / nt;
    int64_t i = cut[t - 1];
    while (i < n && cu_seqlens[i] < want) ++i;
    cut[t] = i;
  }
  host_pool().run(nt, [&](int t) { pack_range(seq, rows, cu_seqlens, cut[t], cut[t + 1], dim, dst_bf16, dst); });
  return 0;
}
