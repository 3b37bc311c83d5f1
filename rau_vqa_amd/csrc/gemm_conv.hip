// gemm_conv.hip -- the two per-position 1x1 convolutions of the RAU and their
// gradients as flattened-(sample,position) GEMMs on the f32-MFMA engine.
// These carry ~94% of the path's FLOPs.
#include <cstdlib>

#include "gemm_core.h"
#include "kernels.h"

namespace rau {

constexpr int BK = 32;

// I[b,m,s] = tanh(sum_d Wi[m,d] X'[b,d,s] + bi[m])  -- reference SS:238-242.
// GEMM: M rows = multfeat, N cols = nB*S, K = D.  A = Wi^T [D][M] (the weight
// transposed once per step by transpose2d: a row-contiguous A operand stages with
// 16-byte LDS writes and full-line global reads, ~10% faster than the k-contiguous
// loader), B = X' [b][d][s] (position contiguous; dropout already applied by
// dropout_features).  nB may be a group of hops.


// 14 x 14 maps in exact f32: the wide tiling (conv_wide.hip) takes the samples in groups of four; a
// remainder of 1-3 samples goes to the tiling `rest` falls back to.  RAU_CONV_WIDE_PER_CU=1|2 (A/B
// knob): workgroups of the wide kernel per CU.
static bool wide_on(int which) {   // RAU_CONV_WIDE=<mask>: 1 i_embed forward, 2 ifeatproj forward
  static const int mask = [] { const char* e = std::getenv("RAU_CONV_WIDE"); return e ? std::atoi(e) : 3; }();
  return (mask & which) != 0;
}
static int wide_per_cu() {   // RAU_CONV_WIDE_PER_CU=1|2: workgroups per CU of the forward convs
  static const int v = [] { const char* e = std::getenv("RAU_CONV_WIDE_PER_CU"); return e ? std::atoi(e) : 1; }();
  return v == 2 ? 2 : 1;
}

hipError_t conv_embed_fwd(hipStream_t st, int nB, int D, int S, int M, const float* X,
                          const float* WiT, const float* bi, float* I, int bf16, int one_per_cu) {
  GemmParams P{};
  P.M = M; P.N = nB * S; P.K = D; P.nk = (D + BK - 1) / BK;
  P.A = WiT; P.a_rs = M;
  P.B = X; P.b_rs = S; P.b_bs = (long)D * S;
  P.S = S;
  P.C = I; P.c_bs = (long)M * S;
  P.bias = bi;
  P.act = 1;
  if (!bf16 && wide_on(1) && conv_wide_ok(M, D, S, M) && nB >= 4) {
    const int n4 = nB & ~3;
    hipError_t e = conv_wide(st, 0, n4, M, D, S, WiT, M, X, (long)D * S, I, (long)M * S, bi, 1, nullptr,
                             nullptr, nullptr, nullptr, 0, one_per_cu ? 1 : wide_per_cu());
    if (e != hipSuccess || n4 == nB) return e;
    return conv_embed_fwd(st, nB - n4, D, S, M, X + (size_t)n4 * D * S, WiT, bi, I + (size_t)n4 * M * S,
                          bf16, one_per_cu);
  }
  if (!bf16 && conv_sample_ok(S, 1))   // 14 x 14 maps: one sample per tile column block (gemm_sample.hip)
    return conv_sample(st, 0, nB, M, D, S, WiT, M, X, (long)D * S, I, (long)M * S, bi, 1, nullptr, nullptr);
  if (bf16 == 2) return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_CONV, 2>(st, P, 1);
  if (bf16) return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_CONV, 1>(st, P, 1);
  return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_CONV>(st, P, 1, one_per_cu ? 30 * 1024 : 0);
}

// The forward bf16 tiles are compiled for 128 registers (four could share a CU) but take a little unused
// LDS so that only THREE do: with four the register file is full and every kernel of the recurrence waits
// for a tile to retire before it can start -- in this mode the forward phase is bound by the encoder
// (8.43 -> 8.22 ms per step at D = 2048; with 149-register tiles three per CU: 8.75).  Round 4: the
// quad-planar images (gemm_core.h ImgQuads) make a tile's two stages exactly 40 KB = a quarter of the CU, so
// 1 KB of padding holds the count at three; the round-3 padding of 8 KB on top of that left the recurrence's
// kernels 13 KB of LDS per CU and cost the step 0.3 ms (7.78-7.81 vs 7.47-7.52).
static int b16_fwd_pad() { return 1024; }
hipError_t conv_embed_fwd_b16(hipStream_t st, int nB, int D, int S, int M, const void* X16,
                              const void* WiT16, const float* bi, float* I) {
  GemmParams P{};
  P.M = M; P.N = nB * S; P.K = D; P.nk = (D + BK - 1) / BK;
  P.A = reinterpret_cast<const float*>(WiT16); P.a_rs = M;
  P.B = reinterpret_cast<const float*>(X16); P.b_rs = S; P.b_bs = (long)D * S;   // in bf16 elements
  P.S = S;
  P.C = I; P.c_bs = (long)M * S;
  P.bias = bi;
  P.act = 1;
  return launch_gemm<128, 128, BK, SRC_RC_B16, SRC_RC_FLAT_B16, EPI_CONV, 1>(st, P, 1, b16_fwd_pad());
}
hipError_t conv_att_pre_b16(hipStream_t st, int nB, int M, int S, int A, const float* I,
                            const void* WpT16, const float* bp, float* Pout) {
  GemmParams P{};
  P.M = A; P.N = nB * S; P.K = M; P.nk = (M + BK - 1) / BK;
  P.A = reinterpret_cast<const float*>(WpT16); P.a_rs = A;
  P.B = I; P.b_rs = S; P.b_bs = (long)M * S;
  P.S = S;
  P.C = Pout; P.c_bs = (long)A * S;
  P.bias = bp;
  P.act = 0;
  return launch_gemm<128, 128, BK, SRC_RC_B16, SRC_RC_FLAT, EPI_CONV, 1>(st, P, 1, b16_fwd_pad());
}

// P[b,k,s] = sum_m Wp[k,m] I[b,m,s] + bp[k]: the hop-invariant part of attbycontent's
// pre-activation (reference SS:247-249); nB may be H*B.  The per-hop part
// (+ u[b,k], tanh, score, softmax) is att_fwd_fused in kernels.hip.
hipError_t conv_att_pre(hipStream_t st, int nB, int M, int S, int A, const float* I,
                        const float* WpT, const float* bp, float* Pout, int bf16, int one_per_cu) {
  GemmParams P{};
  P.M = A; P.N = nB * S; P.K = M; P.nk = (M + BK - 1) / BK;
  P.A = WpT; P.a_rs = A;                // Wp^T [M][A]: reduction-major
  P.B = I; P.b_rs = S; P.b_bs = (long)M * S;
  P.S = S;
  P.C = Pout; P.c_bs = (long)A * S;
  P.bias = bp;
  P.act = 0;
  if (!bf16 && wide_on(2) && conv_wide_ok(A, M, S, A) && nB >= 4) {
    const int n4 = nB & ~3;
    hipError_t e = conv_wide(st, 0, n4, A, M, S, WpT, A, I, (long)M * S, Pout, (long)A * S, bp, 0, nullptr,
                             nullptr, nullptr, nullptr, 0, one_per_cu ? 1 : wide_per_cu());
    if (e != hipSuccess || n4 == nB) return e;
    return conv_att_pre(st, nB - n4, M, S, A, I + (size_t)n4 * M * S, WpT, bp, Pout + (size_t)n4 * A * S,
                        bf16, one_per_cu);
  }
  if (!bf16 && conv_sample_ok(S, 2))
    return conv_sample(st, 0, nB, A, M, S, WpT, A, I, (long)M * S, Pout, (long)A * S, bp, 0, nullptr, nullptr);
  if (bf16 == 2) return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_CONV, 2>(st, P, 1);
  if (bf16) return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_CONV, 1>(st, P, 1);
  return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_CONV>(st, P, 1, one_per_cu ? 30 * 1024 : 0);
}

// backward of attbycontent + attselect into the gradient w.r.t. i_embed's OUTPUT:
// dI[b,m,s] = sum_k Wp[k,m] dS[b,k,s] + dj[b,m] a[b,s].  The tanh derivative
// (1 - I^2) is applied where dI is consumed (conv_embed_wgrad's operand loader, which
// also yields the bias gradient), not here: reading I in this short-K GEMM's epilogue cost more
// than the GEMM itself.
hipError_t conv_att_dgrad(hipStream_t st, int nB, int M, int S, int A, const float* dS,
                          const float* Wp, const float* dj, const float* a, float* dI, int bf16) {
  GemmParams P{};
  P.M = M; P.N = nB * S; P.K = A; P.nk = (A + BK - 1) / BK;
  P.A = Wp; P.a_rs = M;                 // Wp stored [A][M]: reduction-major, m contiguous
  P.B = dS; P.b_rs = S; P.b_bs = (long)A * S;
  P.S = S;
  P.C = dI; P.c_bs = (long)M * S;
  P.v1 = dj; P.v2 = a;
  if (bf16 != 2 && M % 4 == 0 && conv_sample_ok(S, 4))   // exact f32 also in RAU_BF16 mode (see conv_dz_fused_ok)
    return conv_sample(st, 1, nB, M, A, S, Wp, M, dS, (long)A * S, dI, (long)M * S, nullptr, 0, dj, a);
  if (bf16 == 2) return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_OUTER, 2>(st, P, 1);
  if (bf16) return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_OUTER, 1>(st, P, 1);
  return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_OUTER>(st, P, 1);
}

bool conv_dz_fused_ok(int S, int M, int bf16) {
  // bf16 == 1 too: the dgrad then runs on the exact-f32 per-sample kernel (K = A = 256: cheap) and
  // the i_embed weight gradient re-reads one A operand (dZ) instead of two (dI, I) per tile column
  return bf16 != 2 && M % 4 == 0 && conv_sample_ok(S, 4);
}
hipError_t conv_att_dgrad_dz(hipStream_t st, int nB, int M, int S, int A, const float* dS,
                             const float* Wp, const float* dj, const float* a, const float* I,
                             float* dZ, float* rs, int dz16, int bf16, int ds16, int light) {
  if (bf16 == 1 && dgrad16_ok(M, A, S, M))
    return dgrad16(st, nB, M, A, S, Wp, M, dS, (long)A * S, dZ, (long)M * S, dj, a, I, rs, dz16, ds16);
  if (ds16) return hipErrorInvalidValue;
  // f32 output on the shapes dgrad_dma.hip takes: per-sample tiles fed by LDS-DMA (round 4).  Two of its
  // workgroups hold 480 of a SIMD's 512 registers: where the recurrence is the longer path (contexts of up to
  // 64 samples) its kernels then wait for a bulk tile to retire -- 4.23 / 4.18 vs 4.13 / 4.09 ms on the
  // 64-sample shard -- so such callers keep the register-staged tile
  if (!dz16 && !light && dgrad_dma_ok(M, A, S, M))
    return dgrad_dma(st, nB, M, A, S, Wp, M, dS, (long)A * S, dZ, (long)M * S, dj, a, I, rs);
  // one sample per tile (the wide tiling with this epilogue is a concluded negative in the step:
  // DESIGN.md section 8; tools/convbench keeps it for the stand-alone comparison)
  return conv_sample(st, 2, nB, M, A, S, Wp, M, dS, (long)A * S, dZ, (long)M * S, nullptr, 0, dj, a, I, rs,
                     dz16);
}

// dX'[b,d,s] = sum_m Wi[m,d] dZ[b,m,s]: gradient w.r.t. i_embed's (dropped-out) input.  The
// reference computes it and discards it (SS:579); only the module-level
// rau_multimodal_backward forms it, and only on request.
hipError_t conv_embed_dgrad(hipStream_t st, int nB, int D, int S, int M, const float* dZ,
                            const float* Wi, float* dX) {
  GemmParams P{};
  P.M = D; P.N = nB * S; P.K = M; P.nk = (M + BK - 1) / BK;
  P.A = Wi; P.a_rs = D;                 // Wi stored [M][D]: reduction-major, d contiguous
  P.B = dZ; P.b_rs = S; P.b_bs = (long)M * S;
  P.S = S;
  P.C = dX; P.c_bs = (long)D * S;
  P.bias = nullptr;
  P.act = 0;
  if (conv_sample_ok(S, 8))
    return conv_sample(st, 0, nB, D, M, S, Wi, D, dZ, (long)M * S, dX, (long)D * S, nullptr, 0, nullptr, nullptr);
  return launch_gemm<128, 128, BK, SRC_RC, SRC_RC_FLAT, EPI_CONV>(st, P, 1);
}

static int conv_wgrad_splits(int nB, int rowsA, int rowsB) {
  const int tiles = ((rowsA + 127) / 128) * ((rowsB + 127) / 128);
  constexpr int target = 512;   // = one resident wave of workgroups (2 per CU x 256 CUs)
  int s = target / tiles;
  if (s > nB) s = nB;
  if (s < 1) s = 1;
  return s;
}
size_t conv_wgrad_slab_floats(int nB, int rowsA, int rowsB, int S) {
  // split partials of the product + of the A operand's row sums (i_embed bias gradient)
  return (size_t)conv_wgrad_splits(nB, rowsA, rowsB) * ((size_t)rowsA * rowsB + rowsA);
}

// dW[ra, rb] += sum_{b,s} Aop[b,ra,s] * Bop[b,rb,s]; whole samples per split,
// partial tiles to slabs, fixed-order reduction into dW.
template <int BKT, int ASRC, int DT = 0, int BSRC = SRC_SC>
static hipError_t conv_wgrad(hipStream_t st, GemmParams P, int nB, int S, float* dW,
                             float* slab, float* drow = nullptr) {
  P.S = S;
  P.cps = (S + BKT - 1) / BKT;
  P.K = S;
  P.nk = nB * P.cps;
  const int s = conv_wgrad_splits(nB, P.M, P.N);
  const int spb = (nB + s - 1) / s;   // samples per split
  P.C = slab; P.c_rs = P.N; P.slab_stride = (long)P.M * P.N;
  P.tiles_m = (P.M + 127) / 128;
  P.tiles_n = (P.N + 127) / 128;
  P.nk_per_split = spb * P.cps;
  const int splits = (P.nk + P.nk_per_split - 1) / P.nk_per_split;
  P.rs_out = drow ? slab + (size_t)splits * P.M * P.N : nullptr;
  {   // a split's tiles share an XCD (and its L2 copy of the split's samples)
    dim3 grid(8 * ((splits + 7) / 8) * P.tiles_m * P.tiles_n);
    hipLaunchKernelGGL((gemm_split_xcd_kernel<128, 128, BKT, ASRC, BSRC, EPI_SLAB, DT>), grid,
                       dim3(256), 0, st, P, splits);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = splitk_reduce_acc(st, (size_t)P.M * P.N, splits, slab, (size_t)P.M * P.N, dW);
  if (e != hipSuccess || !drow) return e;
  return splitk_reduce_acc(st, (size_t)P.M, splits, P.rs_out, (size_t)P.M, drow);
}
template <int ASRC>
static hipError_t conv_wgrad_any(hipStream_t st, const GemmParams& P, int nB, int S, float* dW,
                                 float* slab, int bf16, float* drow = nullptr) {
  if (!bf16 && ASRC == SRC_SC && !drow && wgrad_dma_ok(P.M, P.N, S))   // exact f32, plain operands
    return wgrad_dma(st, nB, P.M, P.N, S, P.A, P.a_bs, P.B, P.b_bs, dW, slab, conv_wgrad_splits(nB, P.M, P.N));
  // bf16 MFMA steps are 16 deep: 32-wide chunks, the last one of a 196-position map zero-filled
  if (bf16 == 2) return conv_wgrad<32, ASRC, 2>(st, P, nB, S, dW, slab, drow);
  if (bf16) return conv_wgrad<32, ASRC, 1>(st, P, nB, S, dW, slab, drow);
  // 14x14 maps: 196 = 7 * 28, so a 28-deep K-step wastes no MFMA work on padding
  if (S % 28 == 0) return conv_wgrad<28, ASRC>(st, P, nB, S, dW, slab, drow);
  return conv_wgrad<32, ASRC>(st, P, nB, S, dW, slab, drow);
}

// dWp[k,m] += sum_{b,s} dS[b,k,s] I[b,m,s]
hipError_t conv_att_wgrad(hipStream_t st, int nB, int M, int S, int A, const float* dS,
                          const float* I, float* dWp, float* slab, int bf16) {
  GemmParams P{};
  P.M = A; P.N = M;
  P.A = dS; P.a_bs = (long)A * S;
  P.B = I; P.b_bs = (long)M * S;
  return conv_wgrad_any<SRC_SC>(st, P, nB, S, dWp, slab, bf16);
}

hipError_t conv_att_wgrad_ds16(hipStream_t st, int nB, int M, int S, int A, const void* dS16,
                               const float* I, float* dWp, float* slab) {
  GemmParams P{};
  P.M = A; P.N = M;
  P.A = reinterpret_cast<const float*>(dS16); P.a_bs = (long)A * S;   // in bf16 elements
  P.B = I; P.b_bs = (long)M * S;
  return conv_wgrad<32, SRC_SC_B16, 1, SRC_SC>(st, P, nB, S, dWp, slab);
}

// dWi[m,d] += sum_{b,s} dZ[b,m,s] X'[b,d,s] with dZ = dI * (1 - I^2) formed while
// staging the operand (nB may be a group of hops)
hipError_t conv_embed_wgrad(hipStream_t st, int nB, int D, int S, int M, const float* dI,
                            const float* I, const float* X, float* dWi, float* slab, int bf16,
                            float* dbi, int dz_final) {
  GemmParams P{};
  P.M = M; P.N = D;
  P.A = dI; P.A2 = I; P.a_bs = (long)M * S;
  P.B = X; P.b_bs = (long)D * S;
  if (dz_final) return conv_wgrad_any<SRC_SC>(st, P, nB, S, dWi, slab, bf16);
  return conv_wgrad_any<SRC_SC_DTANH>(st, P, nB, S, dWi, slab, bf16, dbi);
}

hipError_t conv_embed_wgrad_b16(hipStream_t st, int nB, int D, int S, int M, const void* dZ16,
                                const void* X16, float* dWi, float* slab) {
  if (wgrad16_ok(M, D, S))
    return wgrad16(st, nB, M, D, S, dZ16, (long)M * S, X16, (long)D * S, dWi, slab);
  GemmParams P{};
  P.M = M; P.N = D;
  P.A = reinterpret_cast<const float*>(dZ16); P.a_bs = (long)M * S;  // both in bf16 elements
  P.B = reinterpret_cast<const float*>(X16); P.b_bs = (long)D * S;
  return conv_wgrad<32, SRC_SC_B16, 1, SRC_SC_B16>(st, P, nB, S, dWi, slab);
}

}  // namespace rau
