// attcheck.hip -- dumps att_fwd_fused's outputs on random data (development tool): run once with and
// once without RAU_ATT_DMA_OFF and compare the files.   ./attcheck nB M A S out.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../rau_vqa_amd/csrc/kernels.h"
using namespace rau;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static std::vector<std::vector<float>> keep;
static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() % 2001) / 1000.f - 1.f);
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  keep.push_back(h);
  return d;
}
int main(int argc, char** argv) {
  const int nB = atoi(argv[1]), M = atoi(argv[2]), A = atoi(argv[3]), S = atoi(argv[4]);
  srand(7);
  hipStream_t st; CK(hipStreamCreate(&st));
  float* P = dev_rand((size_t)nB * A * S, 1.f), *u = dev_rand((size_t)nB * A, 0.5f), *ws = dev_rand(A, 0.3f);
  float* bs = dev_rand(1, 0.1f), *zm = dev_rand((size_t)nB * S, 0.5f), *I = dev_rand((size_t)nB * M * S, 0.9f);
  float* qf = dev_rand((size_t)nB * M, 0.5f);
  float *a, *jv; CK(hipMalloc(&a, (size_t)nB * S * 4)); CK(hipMalloc(&jv, (size_t)nB * M * 4));
  CK(hipMemset(a, 0, (size_t)nB * S * 4)); CK(hipMemset(jv, 0, (size_t)nB * M * 4));
  CK(att_fwd_fused(st, nB, M, A, S, P, u, ws, bs, zm, I, qf, nullptr, a, jv));
  CK(hipStreamSynchronize(st));
  std::vector<float> h((size_t)nB * (S + M));
  CK(hipMemcpy(h.data(), a, (size_t)nB * S * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h.data() + (size_t)nB * S, jv, (size_t)nB * M * 4, hipMemcpyDeviceToHost));
  FILE* f = fopen(argv[5], "wb"); fwrite(h.data(), 4, h.size(), f); fclose(f);
  // host reference in double
  const std::vector<float>&hP = keep[0], &hu = keep[1], &hws = keep[2], &hbs = keep[3], &hzm = keep[4], &hI = keep[5], &hqf = keep[6];
  double ea = 0, ej = 0;
  for (int b = 0; b < nB; ++b) {
    std::vector<double> e(S), av(S);
    double mx = -1e300;
    for (int s = 0; s < S; ++s) {
      double v = hbs[0] + hzm[(size_t)b * S + s];
      for (int k = 0; k < A; ++k) v += hws[k] * tanh((double)hP[((size_t)b * A + k) * S + s] + hu[(size_t)b * A + k]);
      e[s] = v; if (v > mx) mx = v;
    }
    double den = 0; for (int s = 0; s < S; ++s) { av[s] = exp(e[s] - mx); den += av[s]; }
    for (int s = 0; s < S; ++s) { av[s] /= den; ea = fmax(ea, fabs(av[s] - h[(size_t)b * S + s])); }
    for (int m = 0; m < M; ++m) {
      double v = hqf[(size_t)b * M + m];
      for (int s = 0; s < S; ++s) v += hI[((size_t)b * M + m) * S + s] * av[s];
      ej = fmax(ej, fabs(v - h[(size_t)nB * S + (size_t)b * M + m]));
    }
  }
  printf("max |a - ref| %.3g   max |jv - ref| %.3g\n", ea, ej);
  return 0;
}
