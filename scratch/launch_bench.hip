// microbenchmark: cost of dependent kernel boundaries, eager vs hipGraph, on this box
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_empty(float* p) { if (p == nullptr) return; }
__global__ void k_touch(float4* p, unsigned n) { unsigned i = blockIdx.x * 256 + threadIdx.x; if (i < n) { float4 v = p[i]; v.x += 1.f; p[i] = v; } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float4* d; unsigned n = 100000; CK(hipMalloc(&d, n * sizeof(float4))); CK(hipMemset(d, 0, n * sizeof(float4)));
  const int N = 2000;
  for (int variant = 0; variant < 3; ++variant) {
    auto enqueue = [&](int cnt) {
      for (int i = 0; i < cnt; ++i) {
        if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, (float*)d);
        else if (variant == 1) hipLaunchKernelGGL(k_touch, dim3((n + 255) / 256), dim3(256), 0, st, d, n);
        else hipLaunchKernelGGL(k_touch, dim3((18000 + 255) / 256), dim3(256), 0, st, d, 18000u);
      }
    };
    enqueue(100); CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::high_resolution_clock::now();
    enqueue(N); CK(hipStreamSynchronize(st));
    double eager = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); enqueue(N); CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    double graph = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (5.0 * N);
    // per-dispatch stamps
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double stamp = 0;
    for (int i = 0; i < 200; ++i) {
      if (variant == 0) hipExtLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, a, b, 0, (float*)d);
      else hipExtLaunchKernelGGL(k_touch, dim3(((variant == 1 ? n : 18000u) + 255) / 256), dim3(256), 0, st, a, b, 0, d, variant == 1 ? n : 18000u);
      CK(hipStreamSynchronize(st)); float ms; CK(hipEventElapsedTime(&ms, a, b)); stamp += ms * 1e3;
    }
    printf("variant %d: eager %.2f us/launch, graph %.2f us/launch, ext-event stamp %.2f us\n", variant, eager, graph, stamp / 200);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
