// Stand-alone check + timing of the recurrence's skinny Linear GEMMs (gemm_nt / gemm_nn with a
// split-K slab): result against a double-precision host product, then the average of 200 launches.
// RAU_SKINNY_DMA_OFF=1 selects the register-staged tile for an A/B.
//   usage: linbench [M] [deep] [bf16] [woff]   deep: 32-deep stages;  bf16: operands rounded in registers, bf16 MFMA;
//          woff: the weights start that many floats behind a 16-byte boundary (the flat parameter vector does that)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../rau_vqa_amd/csrc/kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 256;
  if (argc > 2 && atoi(argv[2])) { rau::skinny_dma_set_deep(1); printf("32-deep stages\n"); }
  if (argc > 3 && atoi(argv[3])) { rau::lin_set_bf16(1); printf("bf16 products\n"); }
  const int woff = argc > 4 ? atoi(argv[4]) : 0;   // W starts `woff` floats behind a 16-byte boundary
  if (woff) printf("W offset by %d floats\n", woff);
  struct Shape { int nn, N, K; };
  const Shape shapes[] = {{0, 2048, 512}, {0, 2048, 1024}, {0, 2048, 1536}, {0, 512, 512}, {0, 1024, 512},
                          {1, 512, 2048}, {1, 1024, 2048}, {1, 512, 512}, {1, 1536, 2048}, {0, 512, 2048},
                          // ragged: attprob's 196 positions as K (forward, dgrad) and as N, the 200-wide embedding
                          {0, 512, 196}, {1, 512, 196}, {1, 196, 512}, {0, 2048, 200}, {1, 200, 2048}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t slab_floats = (size_t)16 << 20;
  float* slab;
  CK(hipMalloc(&slab, slab_floats * 4)); rau::split_ws_register(slab, (slab_floats * 4) / 4);
  for (const Shape& sh : shapes) {
    const int N = sh.N, K = sh.K;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hC((size_t)M * N);
    unsigned seed = 1234 + N + K;
    for (auto& v : hA) v = frand(seed);
    for (auto& v : hW) v = frand(seed) * 0.1f;
    for (auto& v : hb) v = frand(seed);
    float *A, *W, *b, *C, *Wbase;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&Wbase, (hW.size() + 8) * 4));
    W = Wbase + woff;
    CK(hipMalloc(&b, hb.size() * 4)); CK(hipMalloc(&C, hC.size() * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    rau::LinOpts o;
    o.bias = b;
    o.slab = slab;
    o.slab_floats = slab_floats;
    auto run = [&]() {
      // nt: W [N][K];  nn: W [K][N]
      return sh.nn ? rau::gemm_nn(st, M, N, K, A, K, W, N, C, N, o) : rau::gemm_nt(st, M, N, K, A, K, W, K, C, N, o);
    };
    CK(run());
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int m = 0; m < M; m += 7)
      for (int n = 0; n < N; n += 5) {
        double acc = hb[n];
        for (int k = 0; k < K; ++k)
          acc += (double)hA[(size_t)m * K + k] * (sh.nn ? hW[(size_t)k * N + n] : hW[(size_t)n * K + k]);
        const double e = std::fabs(acc - hC[(size_t)m * N + n]);
        if (e > maxerr) maxerr = e;
      }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) CK(run());
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 200; ++i) CK(run());
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / 200;
    printf("%s M=%d N=%d K=%d  max|err|=%.2e  %.1f us/GEMM(+reduce)  %.1f TFLOP/s\n", sh.nn ? "nn" : "nt", M, N, K,
           maxerr, us, 2.0 * M * N * K / us * 1e-6);
    hipFree(A); hipFree(Wbase); hipFree(b); hipFree(C);
  }
  return 0;
}
