// gbar: cost of a grid barrier in a persistent kernel, three variants, agent-scope fences
// usage: gbar [wgs] [rounds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define SPIN_MAX (1 << 22)
// mode 3: like 2 but NO fences: the published data itself moves with agent-scope relaxed atomics (sc1).
// mode 0: one counter.  mode 1: 8 counters (wg % 8 ~ XCD) + waiters sum them.  mode 2: one flag per WG.
template <int MODE>
__global__ __launch_bounds__(256) void persist(unsigned* ctr, float* buf, int rounds, int* err) {
  const unsigned G = gridDim.x;
  float v = buf[blockIdx.x * 256 + threadIdx.x];
  for (int r = 0; r < rounds; ++r) {
    if (MODE == 3)
      __hip_atomic_store(buf + ((blockIdx.x + r) % G) * 256 + threadIdx.x, v + 1.f, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    else
      buf[((blockIdx.x + r) % G) * 256 + threadIdx.x] = v + 1.f;   // something to publish
    __builtin_amdgcn_s_waitcnt(0);   // the wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x < 64) {
      const unsigned l = threadIdx.x;
      if (MODE != 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (MODE == 0) {
        if (l == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(r + 1) * G;
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_MAX) { *err = 1; break; }
        }
      } else if (MODE == 1) {
        if (l == 0)
          __hip_atomic_fetch_add(ctr + 32 * (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(r + 1) * G;
        int spins = 0;
        for (;;) {
          unsigned c = l < 8 ? __hip_atomic_load(ctr + 32 * l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
          c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64);
          if (__shfl(c, 0, 64) >= target) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_MAX) { *err = 1; break; }
        }
      } else {
        if (l == 0) __hip_atomic_store(ctr + blockIdx.x, (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        for (;;) {
          bool ok = true;
          for (unsigned i = l; i < G; i += 64)
            ok &= __hip_atomic_load(ctr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(r + 1);
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_MAX) { *err = 1; break; }
        }
      }
      if (MODE != 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (MODE == 3)
      v = __hip_atomic_load(buf + ((blockIdx.x + r + 1) % G) * 256 + threadIdx.x, __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
    else
      v = buf[((blockIdx.x + r + 1) % G) * 256 + threadIdx.x];      // read a neighbour's value
  }
  buf[blockIdx.x * 256 + threadIdx.x] = v;
}
int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256, rounds = argc > 2 ? atoi(argv[2]) : 1000;
  unsigned* ctr; float* buf; int* err;
  hipMalloc(&ctr, 4096); hipMalloc(&buf, (size_t)wgs * 256 * 4); hipMalloc(&err, 4);
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(ctr, 0, 4096); hipMemset(buf, 0, (size_t)wgs * 256 * 4); hipMemset(err, 0, 4);
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) hipLaunchKernelGGL(persist<0>, dim3(wgs), dim3(256), 0, 0, ctr, buf, rounds, err);
      if (mode == 1) hipLaunchKernelGGL(persist<1>, dim3(wgs), dim3(256), 0, 0, ctr, buf, rounds, err);
      if (mode == 2) hipLaunchKernelGGL(persist<2>, dim3(wgs), dim3(256), 0, 0, ctr, buf, rounds, err);
      if (mode == 3) hipLaunchKernelGGL(persist<3>, dim3(wgs), dim3(256), 0, 0, ctr, buf, rounds, err);
      hipDeviceSynchronize();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      int e = 0; float b0 = 0;
      hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
      hipMemcpy(&b0, buf, 4, hipMemcpyDeviceToHost);
      printf("mode %d wgs %d rounds %d: %.3f ms, %.2f us per barrier, err %d, buf0 %.0f (expect %d)\n", mode,
             wgs, rounds, ms, ms * 1e3 / rounds, e, b0, rounds);
    }
  }
  return 0;
}
