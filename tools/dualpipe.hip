// dualpipe: can a CU run the f32 MFMA pipe and the packed-f32 VALU pipe at full rate at the same time?
// 2 waves per SIMD: mode 0 = both MFMA, 1 = both VALU (v_pk_fma_f32), 2 = one of each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, float* out) {
  const int w = threadIdx.x >> 6;
  const bool mfma = MODE == 0 || (MODE == 2 && w < 4);
  float s = 0.f;
  if (mfma) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    // 16 independent packed accumulators; per iteration 16 v_pk_fma_f32 = 16 x 64 lanes x 2 x 2 flops
    f32x2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 0.f};
    f32x2 a = {threadIdx.x * 1e-3f, 0.25f}, b = {blockIdx.x * 1e-3f + 0.5f, 1.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(a, b, acc[i]);
      // the four MFMAs of the other kind take 4 x 64 cycles; 16 pk_fma take 16 x 4 = 64 cycles:
      // run 4 rounds so that one iteration is the same 256 cycles of pipe time
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(a, b, acc[i]);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
  }
  if (s == 12345.678f) out[0] = s;
}
template <int MODE>
static void run(const char* name, int iters, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int wgs = 256 * 4;
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), 0, 0, iters / 10, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), 0, 0, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)wgs * 8;
  const double mf = MODE == 0 ? waves : MODE == 2 ? waves / 2 : 0, vf = MODE == 1 ? waves : MODE == 2 ? waves / 2 : 0;
  const double fl_m = mf * iters * 4.0 * 4096, fl_v = vf * iters * 64.0 * 64 * 2 * 2;
  printf("%-18s %.3f ms: MFMA %.1f TF + VALU %.1f TF = %.1f TF\n", name, ms, fl_m / ms / 1e9, fl_v / ms / 1e9,
         (fl_m + fl_v) / ms / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 4);
  run<0>("mfma only", 20000, out);
  run<1>("pk_fma only", 20000, out);
  run<2>("mfma + pk_fma", 20000, out);
  return 0;
}
