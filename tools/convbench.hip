// convbench.hip -- stand-alone check + timing of the wide-tile conv kernels (conv_wide.hip) against
// the round-2 tilings (gemm_conv.hip / gemm_sample.hip) on random data.  Development tool.
//   ./convbench [B] [what]      what: fwd | pre | dz | d2048 | all (default)
// Prints, per shape: bitwise agreement with the old kernel, avg us and TFLOP/s of both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <cmath>
#include "../rau_vqa_amd/csrc/kernels.h"
#include "exp_fwd16.h"
#include "exp_wide2.h"
using namespace rau;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static float* dev_rand(size_t n, float scale = 1.f, bool nonneg = false) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) {
    float v = (rand() % 2001) / 1000.f - 1.f;
    if (nonneg) v = v < 0 ? 0.f : 2.f * v;      // half of the elements dropped, the rest doubled
    h[i] = scale * v;
  }
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}
static double timeit(hipStream_t st, int iters, const std::function<hipError_t()>& f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK(f());
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int i = 0; i < iters; ++i) CK(f());
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / iters;
}
static void report(const char* name, double us, double flops) {
  printf("  %-34s %9.1f us  %7.1f TFLOP/s (%.3f of 157.3)\n", name, us, flops / us / 1e6,
         flops / us / 1e6 / 157.3);
  fflush(stdout);
}
static void compare(const char* what, const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  size_t nd = 0; double md = 0, mx = 0;
  for (size_t i = 0; i < n; ++i) {
    if (memcmp(&ha[i], &hb[i], 4)) ++nd;
    double d = fabs((double)ha[i] - hb[i]); if (d > md) md = d;
    if (fabs(hb[i]) > mx) mx = fabs(hb[i]);
  }
  printf("  %-34s %zu of %zu words differ, max |diff| %.3g (max |ref| %.3g)%s\n", what, nd, n, md, mx,
         nd == 0 ? "  BITWISE OK" : (md <= 1e-5 * mx ? "  close" : "  MISMATCH"));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256;
  const char* what = argc > 2 ? argv[2] : "all";
  auto want = [&](const char* w) { return !strcmp(what, "all") || !strcmp(what, w); };
  const int S = 196, M = 512, A = 256;
  hipStream_t st; CK(hipStreamCreate(&st));
  srand(1234);

  for (int D : {512, 2048}) {
    if (D == 512 && !want("fwd")) continue;
    if (D == 2048 && !want("d2048")) continue;
    for (int nh : {1, 2}) {
      const int nB = nh * B;
      printf("conv_embed_fwd  D=%d  nB=%d (M=%d, tanh)\n", D, nB, M);
      float* X = dev_rand((size_t)nB * D * S, 0.5f, true);
      float* WiT = dev_rand((size_t)D * M, 0.08f), *bi = dev_rand(M, 0.08f);
      float *I0, *I1; CK(hipMalloc(&I0, (size_t)nB * M * S * 4)); CK(hipMalloc(&I1, (size_t)nB * M * S * 4));
      CK(hipMemset(I0, 0xff, (size_t)nB * M * S * 4)); CK(hipMemset(I1, 0xee, (size_t)nB * M * S * 4));
      const double fl = 2.0 * M * (double)nB * S * D;
      CK(conv_embed_fwd(st, nB, D, S, M, X, WiT, bi, I0, 0, 0));
      CK(conv_wide(st, 0, nB, M, D, S, WiT, M, X, (long)D * S, I1, (long)M * S, bi, 1, nullptr, nullptr, nullptr, nullptr, 0, 2));
      CK(hipStreamSynchronize(st));
      compare("wide(2/CU) vs round-2 kernel", I1, I0, (size_t)nB * M * S);
      CK(hipMemset(I1, 0xee, (size_t)nB * M * S * 4));
      CK(conv_wide(st, 0, nB, M, D, S, WiT, M, X, (long)D * S, I1, (long)M * S, bi, 1, nullptr, nullptr, nullptr, nullptr, 0, 1));
      CK(hipStreamSynchronize(st));
      compare("wide(1/CU) vs round-2 kernel", I1, I0, (size_t)nB * M * S);
      CK(hipMemset(I1, 0xff, (size_t)nB * M * S * 4));
      CK(hipStreamSynchronize(st));
      if (conv_wide2_ok(M, D, S, M)) {   // round 4: 128 rows x two samples (conv_wide2.hip)
        CK(hipMemset(I1, 0xcc, (size_t)nB * M * S * 4));
        CK(conv_wide2(st, nB, M, D, S, WiT, M, X, (long)D * S, I1, (long)M * S, bi, 1, 1));
        CK(hipStreamSynchronize(st));
        compare("wide2 128x392 vs round-2 kernel", I1, I0, (size_t)nB * M * S);
      }
      for (int rep = 0; rep < 2; ++rep) {
        if (conv_wide2_ok(M, D, S, M)) {
          report("wide2 128x392, 1 per CU", timeit(st, 10, [&] { return conv_wide2(st, nB, M, D, S, WiT, M, X, (long)D * S, I1, (long)M * S, bi, 1, 1); }), fl);
          report("wide2 128x392, 2 per CU", timeit(st, 10, [&] { return conv_wide2(st, nB, M, D, S, WiT, M, X, (long)D * S, I1, (long)M * S, bi, 1, 2); }), fl);
        }
        report("round-2 flattened 128x128", timeit(st, 10, [&] { return conv_embed_fwd(st, nB, D, S, M, X, WiT, bi, I0, 0, 0); }), fl);
        report("round-2 per-sample 128x208", timeit(st, 10, [&] { return conv_sample(st, 0, nB, M, D, S, WiT, M, X, (long)D * S, I0, (long)M * S, bi, 1, nullptr, nullptr); }), fl);
        report("wide 64x784, 2 per CU", timeit(st, 10, [&] { return conv_wide(st, 0, nB, M, D, S, WiT, M, X, (long)D * S, I1, (long)M * S, bi, 1, nullptr, nullptr, nullptr, nullptr, 0, 2); }), fl);
        report("wide 64x784, 1 per CU", timeit(st, 10, [&] { return conv_wide(st, 0, nB, M, D, S, WiT, M, X, (long)D * S, I1, (long)M * S, bi, 1, nullptr, nullptr, nullptr, nullptr, 0, 1); }), fl);
      }
      CK(hipFree(X)); CK(hipFree(WiT)); CK(hipFree(bi)); CK(hipFree(I0)); CK(hipFree(I1));
    }
  }

  if (want("pre")) {
    for (int nh : {1, 2}) {
      const int nB = nh * B;
      printf("conv_att_pre  nB=%d (A=%d rows, K=%d)\n", nB, A, M);
      float* I = dev_rand((size_t)nB * M * S, 0.9f);
      float* WpT = dev_rand((size_t)M * A, 0.08f), *bp = dev_rand(A, 0.08f);
      float *P0, *P1; CK(hipMalloc(&P0, (size_t)nB * A * S * 4)); CK(hipMalloc(&P1, (size_t)nB * A * S * 4));
      CK(hipMemset(P0, 0xff, (size_t)nB * A * S * 4)); CK(hipMemset(P1, 0xee, (size_t)nB * A * S * 4));
      const double fl = 2.0 * A * (double)nB * S * M;
      CK(conv_att_pre(st, nB, M, S, A, I, WpT, bp, P0, 0, 0));
      CK(conv_wide(st, 0, nB, A, M, S, WpT, A, I, (long)M * S, P1, (long)A * S, bp, 0, nullptr, nullptr, nullptr, nullptr, 0, 2));
      CK(hipStreamSynchronize(st));
      compare("wide vs round-2 kernel", P1, P0, (size_t)nB * A * S);
      if (conv_wide2_ok(A, M, S, A)) {
        CK(hipMemset(P1, 0xcc, (size_t)nB * A * S * 4));
        CK(conv_wide2(st, nB, A, M, S, WpT, A, I, (long)M * S, P1, (long)A * S, bp, 0, 1));
        CK(hipStreamSynchronize(st));
        compare("wide2 128x392 vs round-2 kernel", P1, P0, (size_t)nB * A * S);
      }
      for (int rep = 0; rep < 2; ++rep) {
        if (conv_wide2_ok(A, M, S, A)) {
          report("wide2 128x392, 1 per CU", timeit(st, 10, [&] { return conv_wide2(st, nB, A, M, S, WpT, A, I, (long)M * S, P1, (long)A * S, bp, 0, 1); }), fl);
          report("wide2 128x392, 2 per CU", timeit(st, 10, [&] { return conv_wide2(st, nB, A, M, S, WpT, A, I, (long)M * S, P1, (long)A * S, bp, 0, 2); }), fl);
        }
        report("round-2 flattened 128x128", timeit(st, 10, [&] { return conv_att_pre(st, nB, M, S, A, I, WpT, bp, P0, 0, 0); }), fl);
        report("wide 64x784, 2 per CU", timeit(st, 10, [&] { return conv_wide(st, 0, nB, A, M, S, WpT, A, I, (long)M * S, P1, (long)A * S, bp, 0, nullptr, nullptr, nullptr, nullptr, 0, 2); }), fl);
        report("wide 64x784, 1 per CU", timeit(st, 10, [&] { return conv_wide(st, 0, nB, A, M, S, WpT, A, I, (long)M * S, P1, (long)A * S, bp, 0, nullptr, nullptr, nullptr, nullptr, 0, 1); }), fl);
      }
      CK(hipFree(I)); CK(hipFree(WpT)); CK(hipFree(bp)); CK(hipFree(P0)); CK(hipFree(P1));
    }
  }

  if (want("dz")) {
    for (int nh : {1, 2}) {
      const int nB = nh * B;
      printf("conv_att_dgrad_dz  nB=%d (M=%d rows, K=%d)\n", nB, M, A);
      float* dS = dev_rand((size_t)nB * A * S, 0.01f);
      float* Wp = dev_rand((size_t)A * M, 0.08f);
      float* dj = dev_rand((size_t)nB * M, 0.01f), *av = dev_rand((size_t)nB * S, 0.01f);
      float* I = dev_rand((size_t)nB * M * S, 0.9f);
      float *Z0, *Z1, *r0, *r1;
      CK(hipMalloc(&Z0, (size_t)nB * M * S * 4)); CK(hipMalloc(&Z1, (size_t)nB * M * S * 4));
      CK(hipMalloc(&r0, (size_t)nB * M * 4)); CK(hipMalloc(&r1, (size_t)nB * M * 4));
      CK(hipMemset(Z0, 0xff, (size_t)nB * M * S * 4)); CK(hipMemset(Z1, 0xee, (size_t)nB * M * S * 4));
      const double fl = 2.0 * M * (double)nB * S * A;
      CK(conv_sample(st, 2, nB, M, A, S, Wp, M, dS, (long)A * S, Z0, (long)M * S, nullptr, 0, dj, av, I, r0, 0));
      CK(conv_wide(st, 2, nB, M, A, S, Wp, M, dS, (long)A * S, Z1, (long)M * S, nullptr, 0, dj, av, I, r1, 0, 2));
      CK(hipStreamSynchronize(st));
      compare("dZ: wide vs round-2 per-sample", Z1, Z0, (size_t)nB * M * S);
      compare("row sums", r1, r0, (size_t)nB * M);
      if (dgrad_dma_ok(M, A, S, M)) {   // round 4: per-sample tiles fed by LDS-DMA (dgrad_dma.hip)
        CK(hipMemset(Z1, 0xdd, (size_t)nB * M * S * 4)); CK(hipMemset(r1, 0xdd, (size_t)nB * M * 4));
        CK(dgrad_dma(st, nB, M, A, S, Wp, M, dS, (long)A * S, Z1, (long)M * S, dj, av, I, r1));
        CK(hipStreamSynchronize(st));
        compare("dZ: dgrad_dma vs round-2 per-sample", Z1, Z0, (size_t)nB * M * S);
        compare("row sums (dgrad_dma)", r1, r0, (size_t)nB * M);
      }
      for (int rep = 0; rep < 2; ++rep) {
        report("round-2 per-sample 128x208", timeit(st, 10, [&] { return conv_sample(st, 2, nB, M, A, S, Wp, M, dS, (long)A * S, Z0, (long)M * S, nullptr, 0, dj, av, I, r0, 0); }), fl);
        report("wide 64x784, 2 per CU", timeit(st, 10, [&] { return conv_wide(st, 2, nB, M, A, S, Wp, M, dS, (long)A * S, Z1, (long)M * S, nullptr, 0, dj, av, I, r1, 0, 2); }), fl);
        report("wide 64x784, 1 per CU", timeit(st, 10, [&] { return conv_wide(st, 2, nB, M, A, S, Wp, M, dS, (long)A * S, Z1, (long)M * S, nullptr, 0, dj, av, I, r1, 0, 1); }), fl);
        if (dgrad_dma_ok(M, A, S, M))
          report("dgrad_dma per-sample 128x208 (RAU_DGRAD_DMA=2|3: ring depth)", timeit(st, 10, [&] { return dgrad_dma(st, nB, M, A, S, Wp, M, dS, (long)A * S, Z1, (long)M * S, dj, av, I, r1); }), fl);
      }
      CK(hipFree(dS)); CK(hipFree(Wp)); CK(hipFree(dj)); CK(hipFree(av)); CK(hipFree(I));
      CK(hipFree(Z0)); CK(hipFree(Z1)); CK(hipFree(r0)); CK(hipFree(r1));
    }
  }
  if (want("wgrad")) {
    // conv weight gradients: conv_att_wgrad (dWp [A][M]) and conv_embed_wgrad with dZ final (dWi [M][D]);
    // set RAU_WGRAD_DMA_OFF=1 to time the round-2 register-staged kernel instead of wgrad_dma.hip
    {   // correctness against a host reference in double, 8 samples
      const int nb = 8, D = 512;
      std::vector<float> hA((size_t)nb * M * S), hB((size_t)nb * D * S);
      for (auto& v : hA) v = 0.01f * ((rand() % 2001) / 1000.f - 1.f);
      for (auto& v : hB) v = (rand() % 2001) / 1000.f - 1.f;
      float *dA, *dB, *dW, *slab;
      CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4));
      CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
      CK(hipMalloc(&dW, (size_t)M * D * 4)); CK(hipMemset(dW, 0, (size_t)M * D * 4));
      CK(hipMalloc(&slab, conv_wgrad_slab_floats(nb, M, D, S) * 4)); rau::split_ws_register(slab, (conv_wgrad_slab_floats(nb, M, D, S) * 4) / 4);
      CK(conv_embed_wgrad(st, nb, D, S, M, dA, nullptr, dB, dW, slab, 0, nullptr, 1));
      CK(hipStreamSynchronize(st));
      std::vector<float> hW((size_t)M * D);
      CK(hipMemcpy(hW.data(), dW, hW.size() * 4, hipMemcpyDeviceToHost));
      double md = 0, mx = 0;
      for (int m = 0; m < M; m += 7)
        for (int d = 0; d < D; d += 5) {
          double v = 0;
          for (int b = 0; b < nb; ++b)
            for (int s2 = 0; s2 < S; ++s2) v += (double)hA[((size_t)b * M + m) * S + s2] * hB[((size_t)b * D + d) * S + s2];
          md = fmax(md, fabs(v - hW[(size_t)m * D + d])); mx = fmax(mx, fabs(v));
        }
      printf("conv_embed_wgrad nB=8 vs host double: max |diff| %.3g (max |ref| %.3g)\n", md, mx);
      CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dW)); CK(hipFree(slab));
    }
    for (int nh : {1, 2}) {
      const int nB = nh * B, D = 512;
      float* dS = dev_rand((size_t)nB * A * S, 0.01f), *I = dev_rand((size_t)nB * M * S, 0.9f);
      float* dZ = dev_rand((size_t)nB * M * S, 0.01f), *X = dev_rand((size_t)nB * D * S, 0.5f, true);
      float *dWp, *dWi, *slab;
      CK(hipMalloc(&dWp, (size_t)A * M * 4)); CK(hipMalloc(&dWi, (size_t)M * D * 4));
      size_t sl = conv_wgrad_slab_floats(nB, M, D, S); if (conv_wgrad_slab_floats(nB, A, M, S) > sl) sl = conv_wgrad_slab_floats(nB, A, M, S);
      CK(hipMalloc(&slab, sl * 4)); rau::split_ws_register(slab, (sl * 4) / 4);
      printf("conv weight gradients  nB=%d\n", nB);
      for (int rep = 0; rep < 2; ++rep) {
        report("conv_att_wgrad (256 x 512)", timeit(st, 10, [&] { return conv_att_wgrad(st, nB, M, S, A, dS, I, dWp, slab, 0); }), 2.0 * A * M * (double)nB * S);
        report("conv_embed_wgrad (512 x 512)", timeit(st, 10, [&] { return conv_embed_wgrad(st, nB, D, S, M, dZ, nullptr, X, dWi, slab, 0, nullptr, 1); }), 2.0 * M * D * (double)nB * S);
      }
      CK(hipFree(dS)); CK(hipFree(I)); CK(hipFree(dZ)); CK(hipFree(X)); CK(hipFree(dWp)); CK(hipFree(dWi)); CK(hipFree(slab));
    }
  }
  if (want("wgrad16")) {
    // bf16-operand i_embed weight gradient (configs[2]): conv_embed_wgrad_b16 -> wgrad16.hip;
    // RAU_WGRAD16_OFF=1 times the round-2 128x128x32 register-staged tile instead
    auto to_bf16 = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); };
    auto from_bf16 = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int D : {512, 2048}) {
      {
        const int nb = 5;
        std::vector<uint16_t> hA((size_t)nb * M * S), hB((size_t)nb * D * S);
        for (auto& v : hA) v = to_bf16(0.01f * ((rand() % 2001) / 1000.f - 1.f));
        for (auto& v : hB) v = to_bf16((rand() % 2001) / 1000.f - 1.f);
        uint16_t *dA, *dB; float *dW, *slab;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&dW, (size_t)M * D * 4)); CK(hipMemset(dW, 0, (size_t)M * D * 4));
        size_t sl = conv_wgrad_slab_floats(nb, M, D, S); if (wgrad16_slab_floats(nb, M, D, S) > sl) sl = wgrad16_slab_floats(nb, M, D, S);
        CK(hipMalloc(&slab, sl * 4)); rau::split_ws_register(slab, (sl * 4) / 4);
        CK(conv_embed_wgrad_b16(st, nb, D, S, M, dA, dB, dW, slab));
        CK(hipStreamSynchronize(st));
        std::vector<float> hW((size_t)M * D);
        CK(hipMemcpy(hW.data(), dW, hW.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (int m = 0; m < M; m += 3)
          for (int d = 0; d < D; d += 7) {
            double v = 0;
            for (int b = 0; b < nb; ++b)
              for (int s2 = 0; s2 < S; ++s2)
                v += (double)from_bf16(hA[((size_t)b * M + m) * S + s2]) * from_bf16(hB[((size_t)b * D + d) * S + s2]);
            md = fmax(md, fabs(v - hW[(size_t)m * D + d])); mx = fmax(mx, fabs(v));
          }
        printf("conv_embed_wgrad_b16 D=%d nB=%d vs host double: max |diff| %.3g (max |ref| %.3g)\n", D, nb, md, mx);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dW)); CK(hipFree(slab));
      }
      for (int nh : {1, 2}) {
        const int nB = nh * B;
        uint16_t *dZ, *X; float *dWi, *slab;
        CK(hipMalloc(&dZ, (size_t)nB * M * S * 2)); CK(hipMalloc(&X, (size_t)nB * D * S * 2));
        CK(hipMemset(dZ, 0x3c, (size_t)nB * M * S * 2)); CK(hipMemset(X, 0x3b, (size_t)nB * D * S * 2));
        CK(hipMalloc(&dWi, (size_t)M * D * 4));
        size_t sl = conv_wgrad_slab_floats(nB, M, D, S); if (wgrad16_slab_floats(nB, M, D, S) > sl) sl = wgrad16_slab_floats(nB, M, D, S);
        CK(hipMalloc(&slab, sl * 4)); rau::split_ws_register(slab, (sl * 4) / 4);
        const double bytes = ((double)nB * M * S + (double)nB * D * S) * 2;
        for (int rep = 0; rep < 2; ++rep) {
          const double us = timeit(st, 10, [&] { return conv_embed_wgrad_b16(st, nB, D, S, M, dZ, X, dWi, slab); });
          printf("  conv_embed_wgrad_b16 D=%d nB=%d: %.1f us  %.2f TB/s algorithmic  %.0f TFLOP/s\n", D, nB, us,
                 bytes / us * 1e-6, 2.0 * M * D * (double)nB * S / us * 1e-6);
        }
        CK(hipFree(dZ)); CK(hipFree(X)); CK(hipFree(dWi)); CK(hipFree(slab));
      }
    }
  }
  if (want("fwd16")) {
    // bf16-operand i_embed forward (configs[2]): the EXPERIMENT tools/exp_fwd16.hip vs host double, then
    // its timing next to the library's kernel (conv_embed_fwd_b16)
    auto to_bf16 = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); };
    auto from_bf16 = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int D : {512, 2048}) {
      {
        const int nb = 6;
        std::vector<uint16_t> hX((size_t)nb * D * S);
        std::vector<float> hW((size_t)M * D), hb(M);
        for (auto& v : hX) v = to_bf16((rand() % 2001) / 1000.f - 1.f);
        for (auto& v : hW) v = 0.05f * ((rand() % 2001) / 1000.f - 1.f);
        for (auto& v : hb) v = 0.1f * ((rand() % 2001) / 1000.f - 1.f);
        uint16_t *dX, *dWD; float *dWf, *dbias, *dI;
        CK(hipMalloc(&dX, hX.size() * 2)); CK(hipMalloc(&dWD, (size_t)M * D * 2)); CK(hipMalloc(&dWf, (size_t)M * D * 4));
        CK(hipMalloc(&dbias, M * 4)); CK(hipMalloc(&dI, (size_t)nb * M * S * 4));
        CK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dWf, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dI, 0xff, (size_t)nb * M * S * 4));
        CK(weights_kblocked(st, M, D, dWf, dWD));
        CK(fwd16(st, nb, D, S, M, dX, dWD, dbias, dI));
        CK(hipStreamSynchronize(st));
        std::vector<float> oI((size_t)nb * M * S);
        CK(hipMemcpy(oI.data(), dI, oI.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; size_t nanc = 0;
        for (size_t i = 0; i < oI.size(); ++i) if (oI[i] != oI[i]) ++nanc;
        for (int b = 0; b < nb; ++b)
          for (int m = 0; m < M; m += 5)
            for (int s2 = 0; s2 < S; s2 += 3) {
              double v = hb[m];
              for (int d = 0; d < D; ++d) v += (double)from_bf16(to_bf16(hW[(size_t)m * D + d])) * from_bf16(hX[((size_t)b * D + d) * S + s2]);
              v = tanh(v);
              md = fmax(md, fabs(v - oI[((size_t)b * M + m) * S + s2])); mx = fmax(mx, fabs(v));
            }
        printf("fwd16 D=%d nB=%d vs host double: max |diff| %.3g (max |ref| %.3g), unwritten/NaN words %zu\n", D, nb, md, mx, nanc);
        CK(hipFree(dX)); CK(hipFree(dWD)); CK(hipFree(dWf)); CK(hipFree(dbias)); CK(hipFree(dI));
      }
      for (int nh : {1, 2}) {
        const int nB = nh * B;
        uint16_t *X, *WD, *WT; float *bias, *I;
        CK(hipMalloc(&X, (size_t)nB * D * S * 2)); CK(hipMalloc(&WD, (size_t)M * D * 2)); CK(hipMalloc(&WT, (size_t)M * D * 2));
        CK(hipMemset(X, 0x3b, (size_t)nB * D * S * 2)); CK(hipMemset(WD, 0x3a, (size_t)M * D * 2)); CK(hipMemset(WT, 0x3a, (size_t)M * D * 2));
        CK(hipMalloc(&bias, M * 4)); CK(hipMemset(bias, 0, M * 4));
        CK(hipMalloc(&I, (size_t)nB * M * S * 4));
        const double fb = (double)nB * D * S * 2 + (double)nB * M * S * 4, fl = 2.0 * M * D * (double)nB * S;
        for (int rep = 0; rep < 2; ++rep) {
          double us = timeit(st, 10, [&] { return fwd16(st, nB, D, S, M, X, WD, bias, I); });
          printf("  fwd16            D=%d nB=%d: %.1f us  %.2f TB/s algorithmic  %.0f TFLOP/s\n", D, nB, us, fb / us * 1e-6, fl / us * 1e-6);
          us = timeit(st, 10, [&] { return conv_embed_fwd_b16(st, nB, D, S, M, X, WT, bias, I); });
          printf("  round-2 128x128  D=%d nB=%d: %.1f us  %.2f TB/s algorithmic  %.0f TFLOP/s\n", D, nB, us, fb / us * 1e-6, fl / us * 1e-6);
        }
        CK(hipFree(X)); CK(hipFree(WD)); CK(hipFree(WT)); CK(hipFree(bias)); CK(hipFree(I));
      }
    }
  }
  printf("done\n");
  return 0;
}
