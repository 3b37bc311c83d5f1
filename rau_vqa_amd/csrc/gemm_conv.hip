// gemm_conv.hip -- the two per-position 1x1 convolutions of the RAU and their
// gradients as flattened-(sample,position) GEMMs on the f32-MFMA engine.
// These carry ~94% of the path's FLOPs.
#include "gemm_core.h"
#include "kernels.h"

namespace rau {

// I[b,m,s] = tanh(sum_d Wi[m,d] X'[b,d,s] + bi[m])  -- reference SS:238-242.
// GEMM: M rows = multfeat, N cols = nB*S, K = D.  A = Wi [M][D] (k contiguous),
// B = X [b][d][s] (position contiguous), dropout applied while staging B.
hipError_t conv_embed_fwd(hipStream_t st, int nB, int D, int S, int M, const float* X,
                          const uint32_t* mask, size_t mask_e0, float mscale, const float* Wi,
                          const float* bi, float* I) {
  GemmParams P{};
  P.M = M; P.N = nB * S; P.K = D; P.nk = (D + BK - 1) / BK;
  P.A = Wi; P.a_rs = D;
  P.B = X; P.b_rs = S; P.b_bs = (long)D * S;
  P.S = S;
  P.mask = mask; P.mscale = mscale; P.mask_e0 = mask_e0;
  P.C = I; P.c_bs = (long)M * S;
  P.bias = bi;
  if (mask) return launch_gemm<128, 128, SRC_KC, SRC_RC_FLAT_MASK, EPI_CONV_TANH>(st, P, 1);
  return launch_gemm<128, 128, SRC_KC, SRC_RC_FLAT, EPI_CONV_TANH>(st, P, 1);
}

int conv_att_tiles(int A) { return (A + 127) / 128; }

// attbycontent, reference SS:244-252.
hipError_t conv_att_fwd(hipStream_t st, int nB, int M, int S, int A, const float* I,
                        const float* Wp, const float* bp, const float* u, const float* ws,
                        float* T, float* e_part) {
  GemmParams P{};
  P.M = A; P.N = nB * S; P.K = M; P.nk = (M + BK - 1) / BK;
  P.A = Wp; P.a_rs = M;
  P.B = I; P.b_rs = S; P.b_bs = (long)M * S;
  P.S = S;
  P.C = T; P.c_bs = (long)A * S;
  P.bias = bp; P.u = u; P.v1 = ws; P.out2 = e_part;
  return launch_gemm<128, 128, SRC_KC, SRC_RC_FLAT, EPI_ATT_SCORE>(st, P, 1);
}

// backward of attbycontent + attselect into the i_embed pre-activation gradient.
hipError_t conv_att_dgrad(hipStream_t st, int nB, int M, int S, int A, const float* T,
                          const float* dz, const float* ws, const float* Wp, const float* dj,
                          const float* a, const float* I, float* dZ) {
  GemmParams P{};
  P.M = M; P.N = nB * S; P.K = A; P.nk = (A + BK - 1) / BK;
  P.A = Wp; P.a_rs = M;                 // Wp stored [A][M]: reduction-major, m contiguous
  P.B = T; P.b_rs = S; P.b_bs = (long)A * S;
  P.S = S;
  P.dz = dz; P.ws = ws;
  P.C = dZ; P.c_bs = (long)M * S;
  P.v1 = dj; P.v2 = a; P.I = I;
  return launch_gemm<128, 128, SRC_RC, SRC_RC_FLAT_DS, EPI_DI>(st, P, 1);
}

static int conv_wgrad_splits(int nB, int rowsA, int rowsB, int S) {
  const int tiles = ((rowsA + 127) / 128) * ((rowsB + 127) / 128);
  int s = 1024 / tiles;
  if (s > nB) s = nB;
  if (s < 1) s = 1;
  return s;
}
size_t conv_wgrad_slab_floats(int nB, int rowsA, int rowsB, int S) {
  return (size_t)conv_wgrad_splits(nB, rowsA, rowsB, S) * rowsA * rowsB;
}

template <int ASRC, int BSRC>
static hipError_t conv_wgrad(hipStream_t st, GemmParams P, int nB, int S, float* dW,
                             float* slab) {
  P.S = S;
  P.cps = (S + BK - 1) / BK;
  P.K = S;
  P.nk = nB * P.cps;
  int s = conv_wgrad_splits(nB, P.M, P.N, S);
  // whole samples per split
  const int spb = (nB + s - 1) / s;
  P.C = slab; P.c_rs = P.N; P.slab_stride = (long)P.M * P.N;
  P.tiles_m = (P.M + 127) / 128;
  P.tiles_n = (P.N + 127) / 128;
  P.nk_per_split = spb * P.cps;
  const int splits = (P.nk + P.nk_per_split - 1) / P.nk_per_split;
  dim3 grid(P.tiles_m * P.tiles_n, 1, splits);
  hipLaunchKernelGGL((gemm_kernel<128, 128, ASRC, BSRC, EPI_SLAB>), grid, dim3(256), 0, st, P);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return splitk_reduce_acc(st, (size_t)P.M * P.N, splits, slab, (size_t)P.M * P.N, dW);
}

hipError_t conv_att_wgrad(hipStream_t st, int nB, int M, int S, int A, const float* T,
                          const float* dz, const float* ws, const float* I, float* dWp,
                          float* slab) {
  GemmParams P{};
  P.M = A; P.N = M;
  P.A = T; P.a_bs = (long)A * S;
  P.B = I; P.b_bs = (long)M * S;
  P.dz = dz; P.ws = ws;
  return conv_wgrad<SRC_SC_DS, SRC_SC>(st, P, nB, S, dWp, slab);
}

hipError_t conv_embed_wgrad(hipStream_t st, int nB, int D, int S, int M, const float* dZ,
                            const float* X, const uint32_t* mask, size_t mask_e0, float mscale, float* dWi,
                            float* slab) {
  GemmParams P{};
  P.M = M; P.N = D;
  P.A = dZ; P.a_bs = (long)M * S;
  P.B = X; P.b_bs = (long)D * S;
  P.mask = mask; P.mscale = mscale; P.mask_e0 = mask_e0;
  if (mask) return conv_wgrad<SRC_SC, SRC_SC_MASK>(st, P, nB, S, dWi, slab);
  return conv_wgrad<SRC_SC, SRC_SC>(st, P, nB, S, dWi, slab);
}

}  // namespace rau
