// Dependent-launch gap on this runtime: a chain of N tiny kernels (each needs the previous one's result) launched
// (a) one by one into a stream, host running ahead, (b) as a captured hipGraph replayed with one hipGraphLaunch.
// The global-network passes of the train step are such chains.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launchgap.hip -o tools/micro/launchgap && tools/micro/launchgap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void tiny(float* x, int wgs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  x[i] = x[i] * 1.0001f + 1.0f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int N = 200;
  float* x; CK(hipMalloc(&x, 256 * 1024 * sizeof(float)));
  CK(hipMemset(x, 0, 256 * 1024 * sizeof(float)));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int wgs : {1, 64, 256}) {
    for (int rep = 0; rep < 3; ++rep) {  // eager
      CK(hipEventRecord(a, st));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, x, wgs);
      CK(hipEventRecord(b, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 2) printf("wgs=%3d eager : %.2f us per dependent launch\n", wgs, ms * 1e3f / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, x, wgs);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(b, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 2) printf("wgs=%3d graph : %.2f us per dependent launch\n", wgs, ms * 1e3f / N);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
