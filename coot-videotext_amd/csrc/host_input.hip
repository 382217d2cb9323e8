// Input side of the path: host-side batch collation into a staging arena.
//
// RetrievalDataset.collate_fn (coot/dataset_retrieval.py:335-463) builds four zero-padded fp32 tensors with Python loops of
// tensor slice assignments and hands them to pin_memory / .cuda() one by one.  Here one call per feature level writes the
// padded block (fp32, or bf16 with round-to-nearest-even: half the PCIe bytes) and its mask straight into the caller's
// arena — normally ONE pinned allocation holding all four levels, lengths and masks, moved with ONE hipMemcpyAsync on a copy
// stream (dataset_retrieval.py: BatchArena / DeviceLoader).  Pure host code: no kernel, no stream.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/coot_hip.h"
#include "common.h"

namespace {

inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;  // NaN stays NaN (canonical quiet NaN; rounding the payload could carry into infinity)
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

void collate_range(const float* const* seq, const int64_t* rows, int64_t i0, int64_t i1, int64_t dim, int64_t max_rows, int dst_bf16,
                   void* dst, uint8_t* mask) {
  const size_t esz = dst_bf16 ? 2 : 4;
  for (int64_t i = i0; i < i1; ++i) {
    const int64_t r = rows[i];
    char* d = (char*)dst + (size_t)i * max_rows * dim * esz;
    if (dst_bf16) {
      const float* s = seq[i];
      uint16_t* o = (uint16_t*)d;
      for (int64_t k = 0; k < r * dim; ++k) o[k] = f32_to_bf16_rne(s[k]);
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
  std::vector<std::thread> pool;
  pool.reserve(nt - 1);
  for (int t = 1; t < nt; ++t)
    pool.emplace_back(collate_range, seq, rows, n * t / nt, n * (t + 1) / nt, dim, max_rows, dst_bf16, dst, mask);
  collate_range(seq, rows, 0, n / nt, dim, max_rows, dst_bf16, dst, mask);
  for (auto& th : pool) th.join();
  return 0;
}
