// encbench.hip -- stand-alone timing of the weight-stationary persistent encoder (enc_ws.hip).
//   ./encbench [B] [TL]      development tool
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../rau_vqa_amd/csrc/kernels.h"
using namespace rau;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() % 2001) / 1000.f - 1.f);
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, TL = argc > 2 ? atoi(argv[2]) : 26, R = 512;
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t BR = (size_t)B * R;
  EncWsParams q{};
  q.B = B; q.R = R; q.TL = TL;
  float* G1src = dev_rand((size_t)TL * BR * 4, 0.5f);
  CK(hipMalloc(&q.G1, (size_t)TL * BR * 16)); CK(hipMalloc(&q.G2, (size_t)TL * BR * 16));
  for (float** p : {&q.h1, &q.c1, &q.h2, &q.c2}) { CK(hipMalloc(p, (TL + 1) * BR * 4)); CK(hipMemset(*p, 0, (TL + 1) * BR * 4)); }
  for (float** p : {&q.tc1, &q.tc2, &q.x2}) CK(hipMalloc(p, (size_t)TL * BR * 4));
  q.Wh1 = dev_rand((size_t)4 * R * R, 0.08f); q.Wi2 = dev_rand((size_t)4 * R * R, 0.08f); q.Wh2 = dev_rand((size_t)4 * R * R, 0.08f);
  q.bi2 = dev_rand(4 * R, 0.08f); q.bh2 = dev_rand(4 * R, 0.08f);
  q.mask = nullptr; q.mscale = 1.f;
  CK(hipMalloc(&q.cnt, 64)); int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4)); q.err = err;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; ++rep) {
    float tot = 0;
    const int n = 10;
    for (int i = 0; i < n; ++i) {
      CK(hipMemcpyAsync(q.G1, G1src, (size_t)TL * BR * 16, hipMemcpyDeviceToDevice, st));
      CK(hipEventRecord(a, st));
      CK(enc_ws_forward(st, GATES_DEEP, q));
      CK(hipEventRecord(b, st));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); tot += ms;
    }
    int e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    printf("B=%d TL=%d: %.1f us per launch, %.2f us per token step (err %d, %d workgroups)\n", B, TL, tot / n * 1e3,
           tot / n * 1e3 / TL, e, enc_ws_workgroups(B));
  }
  return 0;
}
