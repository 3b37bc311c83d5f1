// launchgap.hip -- launch-to-launch time of dependent trivial kernels in one stream, by how the stream was
// created (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1;} } while (0)
__global__ void k_tiny(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
  float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("priority range lo=%d hi=%d\n", lo, hi);
  struct { const char* name; int kind; } cfgs[] = {{"null stream", 0}, {"hipStreamCreate", 1}, {"NonBlocking", 2},
                                                  {"NonBlocking + high priority", 3}, {"NonBlocking + low priority", 4},
                                                  {"NonBlocking + priority 0", 5}};
  for (auto& c : cfgs) {
    hipStream_t st = nullptr;
    if (c.kind == 1) CK(hipStreamCreate(&st));
    if (c.kind == 2) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (c.kind == 3) CK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
    if (c.kind == 4) CK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo));
    if (c.kind == 5) CK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, 0));
    for (int wgs : {1, 256}) {
      const int n = 2000;
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_tiny, dim3(wgs), dim3(256), 0, st, d, wgs * 256);
      CK(hipStreamSynchronize(st));
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      auto t0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(a, st));
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_tiny, dim3(wgs), dim3(256), 0, st, d, wgs * 256);
      CK(hipEventRecord(b, st));
      auto t1 = std::chrono::steady_clock::now();
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("%-30s %4d WGs: %.2f us per launch on the device, %.2f us host enqueue\n", c.name, wgs, ms * 1e3 / n,
             std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
    }
  }
  return 0;
}
