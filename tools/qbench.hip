// qbench: do two in-order streams of small dependent kernels overlap on this GPU?
// usage: qbench [wgs] [iters_in_kernel] [launches]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(float* x, int iters) {
  float v = x[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  x[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
static double run(std::vector<hipStream_t>& ss, std::vector<float*>& bufs, int wgs, int iters, int n) {
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i)
    for (size_t s = 0; s < ss.size(); ++s)
      hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, ss[s], bufs[s], iters);
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 128, iters = argc > 2 ? atoi(argv[2]) : 4000,
            n = argc > 3 ? atoi(argv[3]) : 500;
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  for (int mode = 0; mode < 3; ++mode) {   // 0: default-priority streams, 1: all high, 2: high + low
    for (int ns = 1; ns <= 4; ++ns) {
      std::vector<hipStream_t> ss(ns);
      std::vector<float*> bufs(ns);
      for (int s = 0; s < ns; ++s) {
        if (mode == 0) hipStreamCreateWithFlags(&ss[s], hipStreamNonBlocking);
        else hipStreamCreateWithPriority(&ss[s], hipStreamNonBlocking, mode == 1 ? hi : (s & 1 ? lo : hi));
        hipMalloc(&bufs[s], (size_t)wgs * 256 * 4);
        hipMemset(bufs[s], 0, (size_t)wgs * 256 * 4);
      }
      run(ss, bufs, wgs, iters, 20);
      const double ms = run(ss, bufs, wgs, iters, n);
      printf("mode %d streams %d: %.3f ms total, %.2f us per launch per stream\n", mode, ns, ms,
             ms * 1e3 / n);
      for (int s = 0; s < ns; ++s) { hipStreamDestroy(ss[s]); hipFree(bufs[s]); }
    }
  }
  return 0;
}
