// What a cross-stream dependency costs on the stream that waits (the train step has 2-3 of them on its critical path).
// Stream A: k1 (stamps its end) -> [dependency on stream B] -> k2 (stamps its start); gap = k2.start - k1.end in us (s_memrealtime,
// 100 MHz).  Variants: none; hipStreamWaitEvent on an event stream B recorded (a) long ago (already complete when A reaches the wait),
// (b) right behind a kernel on B that finishes ~when k1 does; hipStreamWaitValue32 on a flag a kernel of B wrote.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/hopgap.hip -o tools/micro/hopgap && tools/micro/hopgap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void work(unsigned long long* stamp, int slot, int spin_us, unsigned* flag, unsigned flagval) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    stamp[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    if (flag) { __threadfence_system(); *(volatile unsigned*)flag = flagval; }
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  unsigned long long* stamp; CK(hipMalloc(&stamp, 64 * 8));
  unsigned* flag = nullptr;
  bool have_flag = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory) == hipSuccess;
  if (have_flag) CK(hipMemset(flag, 0, 64));
  hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const char* names[] = {"no dependency", "wait: event of B complete long ago", "wait: event behind a B kernel ending with k1", "hipStreamWaitValue32 on a flag written by B's kernel"};
  for (int variant = 0; variant < 4; ++variant) {
    if (variant == 3 && !have_flag) { printf("%-55s: signal memory not available\n", names[3]); continue; }
    std::vector<double> gaps;
    for (int it = 0; it < 30; ++it) {
      if (variant == 1) { hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, B, stamp, 4, 5, nullptr, 0u); CK(hipEventRecord(ev, B)); CK(hipStreamSynchronize(B)); }
      hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, A, stamp, 0, 200, nullptr, 0u);   // k1: 200 us (the host runs ahead)
      if (variant == 2) { hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, B, stamp, 4, 190, nullptr, 0u); CK(hipEventRecord(ev, B)); }
      if (variant == 3) hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, B, stamp, 4, 190, flag, (unsigned)(it + 1));
      if (variant == 1 || variant == 2) CK(hipStreamWaitEvent(A, ev, 0));
      if (variant == 3) CK(hipStreamWaitValue32(A, flag, (unsigned)(it + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
      hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, A, stamp, 1, 5, nullptr, 0u);     // k2
      CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
      unsigned long long h[4]; CK(hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost));
      if (it >= 5) gaps.push_back((double)(h[2] - h[1]) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("%-55s: gap median %.1f us (min %.1f, max %.1f)\n", names[variant], gaps[gaps.size() / 2], gaps.front(), gaps.back());
  }
  return 0;
}
