// kbench.hip -- kernel micro-benchmark (development tool, not part of librau.so).
// Times individual launchers of rau_vqa_amd/csrc on random data with HIP events.
//   ./kbench [B]      prints avg us + TFLOP/s per kernel class
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include "../rau_vqa_amd/csrc/gemm_core.h"
#include "../rau_vqa_amd/csrc/kernels.h"
#include <cstring>
using namespace rau;

// pure-MFMA calibration: 4 waves per block, NACC independent accumulators
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_peak(int iters, float* out) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static float* dev_rand(size_t n, float scale = 1.f) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() % 2001) / 1000.f - 1.f);
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

// LDS-fragment-read + MFMA loop of the 128x128 tile kernel in isolation (same fragment mapping
// and LDS layout as gemm_kernel::compute): WRITES = re-store the stage each K-step, BAR = barrier
template <int PINNED, int WRITES, int BAR>
__global__ __launch_bounds__(256, 2) void k_ldsmfma(int iters, float* out) {
  constexpr int BK = 32, LD = 132;
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BK * LD];
  float* As = smem;
  float* Bs = smem + 2 * BK * LD;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
  for (int i = tid; i < 2 * 2 * BK * LD; i += 256) smem[i] = (float)((i * 7 + tid) & 15) * 0.125f;
  __syncthreads();
  rau::f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fa = (l >> 5) * LD + wm * 64 + (l & 31);
  const int fb = (l >> 5) * LD + wn * 64 + (l & 31);
  float4 wv = make_float4(0.5f, 0.25f, 0.125f, 1.f);
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    const float* as = As + cur * BK * LD + fa;
    const float* bs = Bs + cur * BK * LD + fb;
    float a[2][2], b[2][2];
    for (int i = 0; i < 2; ++i) a[0][i] = as[i * 32];
    for (int j = 0; j < 2; ++j) b[0][j] = bs[j * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < BK / 2) {
        for (int i = 0; i < 2; ++i) a[nx][i] = as[(kk + 1) * 2 * LD + i * 32];
        for (int j = 0; j < 2; ++j) b[nx][j] = bs[(kk + 1) * 2 * LD + j * 32];
        if (PINNED) __builtin_amdgcn_sched_barrier(0);
      }
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
    }
    if (WRITES) {   // the 8 ds_write_b128 per thread of one K-step, into the other stage
      float* d = smem + (cur ^ 1) * BK * LD + (tid >> 5) * LD + (tid & 31) * 4;
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(d + i * 8 * LD) = wv;
      float* e = Bs + (cur ^ 1) * BK * LD + (tid >> 5) * LD + (tid & 31) * 4;
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(e + i * 8 * LD) = wv;
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
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
  printf("%-28s %9.1f us  %7.1f TFLOP/s (%.0f%% of 157.3)\n", name, us, flops / us / 1e6,
         flops / us / 1e6 / 157.3 * 100);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256;
  const int D = 512, S = 196, M = 512, A = 256;
  hipStream_t st; CK(hipStreamCreate(&st));
  float* X = dev_rand((size_t)B * D * S);
  float* Wi = dev_rand((size_t)M * D, 0.08f), *bi = dev_rand(M, 0.08f);
  float* I; CK(hipMalloc(&I, (size_t)B * M * S * 4));
  uint32_t* mask; CK(hipMalloc(&mask, (size_t)B * D * S / 8 + 64));
  CK(fill_masks(st, 1, 3, 0, 0.5f, (size_t)B * D * S, mask));
  float* Wp = dev_rand((size_t)A * M, 0.08f), *bp = dev_rand(A, 0.08f), *u = dev_rand((size_t)B * A, 0.1f);
  float* ws = dev_rand(A, 0.08f);
  float* T; CK(hipMalloc(&T, (size_t)B * A * S * 4));
  float* epart; CK(hipMalloc(&epart, (size_t)2 * B * S * 4));
  float* dz = dev_rand((size_t)B * S, 0.01f), *dj = dev_rand((size_t)B * M, 0.01f), *a = dev_rand((size_t)B * S, 0.01f);
  float* dZ; CK(hipMalloc(&dZ, (size_t)B * M * S * 4));
  float* dWp = dev_rand((size_t)A * M), *dWi = dev_rand((size_t)M * D);
  size_t sl = conv_wgrad_slab_floats(8 * B, M, D, S); if (conv_wgrad_slab_floats(B, A, M, S) > sl) sl = conv_wgrad_slab_floats(B, A, M, S);
  float* slab; CK(hipMalloc(&slab, sl * 4 + (size_t)64 * 2048 * 512 * 4)); rau::split_ws_register(slab, (sl * 4 + (size_t)64 * 2048 * 512 * 4) / 4);
  const double NS = (double)B * S;
  const char* only = argc > 2 ? argv[2] : "";
  if (!strcmp(only, "peak") || !only[0]) {
    for (int wgs : {256, 512, 1024}) {
      char nm[64]; snprintf(nm, 64, "mfma_peak 4acc grid %d", wgs);
      report(nm, timeit(st, 5, [&] { hipLaunchKernelGGL(k_mfma_peak<4>, dim3(wgs), dim3(256), 0, st, 4000, I); return hipGetLastError(); }), 4096.0 * 4 * 4000 * 4 * wgs);
    }
  }
  if (!strcmp(only, "var") || !only[0]) {
    // main-loop-only variants of the conv_embed_fwd GEMM (raw accumulator stores)
    GemmParams P{};
    P.M = M; P.N = B * S; P.K = D; P.nk = D / 32;
    P.A = Wi; P.a_rs = D; P.B = X; P.b_rs = S; P.b_bs = (long)D * S; P.S = S;
    P.C = I; P.c_rs = P.N; P.slab_stride = 0;
    report("v: KC x RC_FLAT slab-store", timeit(st, 20, [&] { return launch_gemm<128, 128, 32, SRC_KC, SRC_RC_FLAT, EPI_SLAB>(st, P, 1); }), 2.0 * M * NS * D);
    {
      GemmParams P8 = P; P8.N = 8 * B * S; P8.A = Wi; P8.a_rs = M; P8.C = nullptr;
      float* xd8; CK(hipMalloc(&xd8, (size_t)8 * B * D * S * 4)); CK(hipMemset(xd8, 0, (size_t)8 * B * D * S * 4));
      float* I8; CK(hipMalloc(&I8, (size_t)8 * B * M * S * 4));
      P8.B = xd8; P8.C = I8; P8.c_bs = (long)M * S; P8.bias = bi; P8.act = 1;
      for (int dbg : {0, 1, 2, 3, 0}) {
        P8.dbg = dbg;
        char nm[64]; snprintf(nm, 64, "v: embed_fwd x8 dbg=%d", dbg);
        report(nm, timeit(st, 5, [&] { return launch_gemm<128, 128, 32, SRC_RC, SRC_RC_FLAT, EPI_CONV>(st, P8, 1); }), 2.0 * M * NS * D * 8);
      }
      CK(hipFree(xd8)); CK(hipFree(I8));
    }
    for (int dbg : {1, 2, 3}) {
      GemmParams D1 = P; D1.dbg = dbg;
      char nm[64]; snprintf(nm, 64, "v: KC x RC_FLAT dbg=%d", dbg);
      report(nm, timeit(st, 20, [&] { return launch_gemm<128, 128, 32, SRC_KC, SRC_RC_FLAT, EPI_SLAB>(st, D1, 1); }), 2.0 * M * NS * D);
    }
    GemmParams Q = P; Q.B = X; Q.b_rs = P.N; // treat X as plain [K][N] row-major
    report("v: KC x RC slab-store", timeit(st, 20, [&] { return launch_gemm<128, 128, 32, SRC_KC, SRC_RC, EPI_SLAB>(st, Q, 1); }), 2.0 * M * NS * D);
    GemmParams R2 = P; R2.A = X; R2.a_rs = M;  // A as [K][M] row-contig
    report("v: RC x RC slab-store", timeit(st, 20, [&] { return launch_gemm<128, 128, 32, SRC_RC, SRC_RC, EPI_SLAB>(st, R2, 1); }), 2.0 * M * NS * D);
  }
  if (!strcmp(only, "mb")) {
    for (int wgs : {256, 512}) {
      const double fl = (double)wgs * 3000 * 2.0 * 128 * 128 * 32;
      char nm[64];
#define MB(P_, W_, B_)                                                                              \
      snprintf(nm, 64, "lds+mfma pin%d wr%d bar%d g%d", P_, W_, B_, wgs);                           \
      report(nm, timeit(st, 3, [&] { hipLaunchKernelGGL((k_ldsmfma<P_, W_, B_>), dim3(wgs), dim3(256), 0, st, 3000, I); return hipGetLastError(); }), fl)
      MB(0, 0, 0); MB(1, 0, 0); MB(0, 1, 0); MB(0, 1, 1); MB(1, 1, 1);
#undef MB
    }
    return 0;
  }
  if (only[0] && strcmp(only, "conv") && strcmp(only, "all")) return 0;
  report("conv_embed_fwd", timeit(st, 20, [&] { return conv_embed_fwd(st, B, D, S, M, X, Wi, bi, I); }), 2.0 * M * NS * D);
  {
    const int H = 8;
    float* xd; CK(hipMalloc(&xd, (size_t)H * B * D * S * 4));
    float* I8; CK(hipMalloc(&I8, (size_t)H * B * M * S * 4));
    uint32_t* m8; CK(hipMalloc(&m8, (size_t)H * B * D * S / 8 + 64));
    CK(fill_masks(st, 1, 3, 0, 0.5f, (size_t)H * B * D * S, m8));
    report("dropout_features x8", timeit(st, 10, [&] { return dropout_features(st, H, (size_t)B * D * S, X, m8, 2.f, xd); }), 0);
    report("conv_embed_fwd x8 hops", timeit(st, 5, [&] { return conv_embed_fwd(st, H * B, D, S, M, xd, Wi, bi, I8); }), 2.0 * M * NS * D * H);
    report("conv_embed_wgrad x8 hops", timeit(st, 5, [&] { return conv_embed_wgrad(st, H * B, D, S, M, I8, I8, xd, dWi, slab); }), 2.0 * M * NS * D * H);
    CK(hipFree(xd)); CK(hipFree(I8)); CK(hipFree(m8));
  }
  report("conv_att_pre", timeit(st, 20, [&] { return conv_att_pre(st, B, M, S, A, I, Wp, bp, T); }), 2.0 * A * NS * M);
  { float* jv; CK(hipMalloc(&jv, (size_t)B * M * 4));
    report("att_fwd_fused", timeit(st, 20, [&] { return att_fwd_fused(st, B, M, A, S, T, u, ws, bp, dz, I, dj, T, a, jv); }), 0);
    float* du; CK(hipMalloc(&du, (size_t)B * A * 8));
    report("att_bwd_fused", timeit(st, 20, [&] { return att_bwd_fused(st, B, M, A, S, I, dj, a, dz, ws, T, epart, du, du + B * A); }), 0); }
  report("conv_att_dgrad", timeit(st, 20, [&] { return conv_att_dgrad(st, B, M, S, A, T, Wp, dj, a, dZ); }), 2.0 * A * NS * M);
  report("conv_att_wgrad", timeit(st, 20, [&] { return conv_att_wgrad(st, B, M, S, A, T, I, dWp, slab); }), 2.0 * A * NS * M);
  report("conv_embed_wgrad", timeit(st, 20, [&] { return conv_embed_wgrad(st, B, D, S, M, dZ, I, X, dWi, slab); }), 2.0 * M * NS * D);
  // small GEMMs
  float* h = dev_rand((size_t)B * 2048, 0.5f), *W = dev_rand((size_t)2048 * 2048, 0.08f);
  float* C; CK(hipMalloc(&C, (size_t)8 * B * 2048 * 4));
  LinOpts o; o.slab = slab; o.slab_floats = sl;
  struct Sh { const char* n; int N, Kd; };
  const Sh nts[] = {{"nt N2048 K512", 2048, 512}, {"nt N512 K512", 512, 512}, {"nt N1000 K512", 1000, 512},
                    {"nt N196 K512", 196, 512}, {"nt N512 K196", 512, 196}, {"nt N512 K2048", 512, 2048}};
  for (auto& s : nts)
    report(s.n, timeit(st, 50, [&] { return gemm_nt(st, B, s.N, s.Kd, h, s.Kd, W, s.Kd, C, s.N, o); }), 2.0 * B * s.N * s.Kd);
  const Sh nns[] = {{"nn N512 K2048", 512, 2048}, {"nn N512 K1000", 512, 1000}, {"nn N512 K512", 512, 512},
                    {"nn N196 K512", 196, 512}, {"nn N512 K256", 512, 256}, {"nn N2048 K512", 2048, 512}};
  for (auto& s : nns)
    report(s.n, timeit(st, 50, [&] { return gemm_nn(st, B, s.N, s.Kd, h, s.Kd, W, s.N, C, s.N, o); }), 2.0 * B * s.N * s.Kd);
  const int rows = 8 * B;
  float* dY = dev_rand((size_t)rows * 2048, 0.1f), *Xr = dev_rand((size_t)rows * 2048, 0.1f);
  report("tn 2048x512 rows8B", timeit(st, 20, [&] { return gemm_tn_acc(st, 2048, 512, rows, dY, 2048, Xr, 512, W, 512, slab); }), 2.0 * 2048 * 512 * rows);
  report("tn 1000x512 rows8B", timeit(st, 20, [&] { return gemm_tn_acc(st, 1000, 512, rows, dY, 1000, Xr, 512, W, 512, slab); }), 2.0 * 1000 * 512 * rows);
  report("tn 512x2048 rows8B", timeit(st, 20, [&] { return gemm_tn_acc(st, 512, 2048, rows, dY, 512, Xr, 2048, W, 2048, slab); }), 2.0 * 2048 * 512 * rows);
  CK(hipStreamSynchronize(st));
  return 0;
}
