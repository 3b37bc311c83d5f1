// gemm_lin.hip -- Linear-layer GEMMs on the f32-MFMA engine (gemm_core.h):
// forward y = x W^T (gemm_nt), input gradient dx = dy W (gemm_nn) and the
// deterministic split-K weight gradient dW += dy^T x (gemm_tn_acc).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gemm_core.h"
#include "kernels.h"

namespace rau {

constexpr int BK = 32;   // K-step per LDS stage, 128x128 tiles
// Skinny (64x64-tile) GEMMs run on the chain stream next to two resident bulk-GEMM
// workgroups (2 x 67.6 KB of the CU's 160 KB LDS): a 16-deep stage keeps them at
// 17.4 KB so they can co-reside instead of waiting for a bulk workgroup to retire.
constexpr int BKS = 16;

static GemmParams lin_params(int M, int N, int K, const float* A, long lda, const float* W,
                             long ldw, float* C, long ldc, const LinOpts& o) {
  GemmParams P{};
  P.M = M; P.N = N; P.K = K;
  P.nk = (K + BK - 1) / BK;   // callers using another K-step overwrite nk
  P.A = A; P.a_rs = lda;
  P.B = W; P.b_rs = ldw;
  P.C = C; P.c_rs = ldc;
  P.bias = o.bias; P.bias2 = o.bias2;
  P.addend = o.addend; P.add_rs = o.add_rs;
  P.accumulate = o.accumulate; P.act = o.act;
  P.ymul = o.ymul; P.y_rs = o.y_rs;
  P.emask = o.emask; P.emask_e0 = o.emask_e0; P.emscale = o.emscale;
  P.alpha = o.alpha;
  P.round16 = lin_bf16();
  return P;
}

// Skinny problems (few output tiles, long K) are latency-bound per workgroup, so
// they are split over K across workgroups -- but only up to ~160 workgroups: they
// run next to the bulk GEMMs, and more (shorter) workgroups disturb those more than
// they shorten the chain (sweep: 128-192 best, 384 +2%, no split +16%).  The
// partials are combined by lin_reduce_epilogue in fixed order.
static int skinny_splits(int M, int N, int K, const LinOpts& o) {
  if (!o.slab) return 1;
  const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  const int nk = (K + BKS - 1) / BKS;
  constexpr int target = 160;   // workgroups a skinny launch aims for (64 .. 200 measured equal)
  int s = (target + tiles - 1) / tiles;
  if (s > nk / 2) s = nk / 2;
  if (s < 1) s = 1;
  while (s > 1 && (size_t)s * M * N > o.slab_floats) --s;
  const int per = (nk + s - 1) / s;
  return (nk + per - 1) / per;
}

template <int ASRC, int BSRC>
static hipError_t lin_gemm(hipStream_t st, int M, int N, int K, const float* A, long lda,
                           const float* W, long ldw, float* C, long ldc, const LinOpts& o) {
  GemmParams P = lin_params(M, N, K, A, lda, W, ldw, C, ldc, o);
  if ((long)M * N >= 128L * 128 * 256 && !o.defer_splits)
    return launch_gemm<128, 128, BK, ASRC, BSRC, EPI_LIN>(st, P, 1);
  if (o.slab && skinny_dma_ok(M, K, lda, ldw, BSRC == SRC_RC, 1, &N, &A, &W)) {
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    const int s = skinny_dma_splits(M, K, tiles, (size_t)N, o.slab_floats);
    if ((size_t)s * M * N <= o.slab_floats && (s > 1 || o.defer_splits)) {
      const long off = 0;
      hipError_t e = skinny_dma(st, BSRC == SRC_RC, 1, M, K, &A, lda, &W, ldw, &N, o.slab, &off, s);
      if (e != hipSuccess) return e;
      if (o.defer_splits) {
        *o.defer_splits = s;
        return hipSuccess;
      }
      return lin_reduce_epilogue(st, M, N, s, o.slab, C, ldc, o);
    }
  }
#ifdef RAU_DEV_HOOKS
  if (std::getenv("RAU_LIN_TRACE"))
    std::fprintf(stderr, "lin_gemm fallback %s M=%d N=%d K=%d lda=%ld ldw=%ld slab=%d\n",
                 BSRC == SRC_RC ? "nn" : "nt", M, N, K, lda, ldw, o.slab != nullptr);
#endif
  P.nk = (K + BKS - 1) / BKS;
  const int s = skinny_splits(M, N, K, o);
  if (o.defer_splits) {  // partials stay in the slab; the consumer kernel reduces them
    if (!o.slab || (size_t)s * M * N > o.slab_floats) return hipErrorInvalidValue;
    P.C = o.slab;
    P.c_rs = N;
    P.slab_stride = (long)M * N;
    *o.defer_splits = s;
    return launch_gemm<64, 64, BKS, ASRC, BSRC, EPI_SLAB>(st, P, s);
  }
  if (s <= 1) return launch_gemm<64, 64, BKS, ASRC, BSRC, EPI_LIN>(st, P, 1);
  P.C = o.slab;
  P.c_rs = N;
  P.slab_stride = (long)M * N;
  hipError_t e = launch_gemm<64, 64, BKS, ASRC, BSRC, EPI_SLAB>(st, P, s);
  if (e != hipSuccess) return e;
  return lin_reduce_epilogue(st, M, N, s, o.slab, C, ldc, o);
}

hipError_t gemm_nt(hipStream_t st, int M, int N, int K, const float* A, long lda,
                   const float* W, long ldw, float* C, long ldc, const LinOpts& o) {
  return lin_gemm<SRC_KC, SRC_KC>(st, M, N, K, A, lda, W, ldw, C, ldc, o);
}

hipError_t gemm_nn(hipStream_t st, int M, int N, int K, const float* A, long lda,
                   const float* W, long ldw, float* C, long ldc, const LinOpts& o) {
  return lin_gemm<SRC_KC, SRC_RC>(st, M, N, K, A, lda, W, ldw, C, ldc, o);
}

// nb same-shape skinny problems in ONE launch, partials left in the slab laid out
// [problem][split][M*N]; the consumer kernel reduces them (deferred, fixed order).
template <int ASRC, int BSRC>
static hipError_t batched_deferred(hipStream_t st, int nb, int M, int N, int K,
                                   const float* const* A, long lda, const float* const* W,
                                   long ldw, float* slab, size_t slab_floats, int* splits) {
  if (nb < 1 || nb > 3) return hipErrorInvalidValue;
  {
    const int Ns[3] = {N, N, N};
    if (skinny_dma_ok(M, K, lda, ldw, BSRC == SRC_RC, nb, Ns, A, W)) {
      const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
      const int s = skinny_dma_splits(M, K, nb * tiles, (size_t)nb * N, slab_floats);
      if ((size_t)nb * s * M * N <= slab_floats) {
        long off[3];
        for (int i = 0; i < nb; ++i) off[i] = (long)i * s * M * N;
        *splits = s;
        return skinny_dma(st, BSRC == SRC_RC, nb, M, K, A, lda, W, ldw, Ns, slab, off, s);
      }
    }
  }
#ifdef RAU_DEV_HOOKS
  if (std::getenv("RAU_LIN_TRACE"))
    std::fprintf(stderr, "batched fallback %s nb=%d M=%d N=%d K=%d slab_floats=%zu\n", BSRC == SRC_RC ? "nn" : "nt",
                 nb, M, N, K, slab_floats);
#endif
  LinOpts o;
  o.slab = slab;
  o.slab_floats = slab_floats / nb;
  GemmParams P = lin_params(M, N, K, A[0], lda, W[0], ldw, slab, N, o);
  P.nk = (K + BKS - 1) / BKS;
  // same K-split as a single problem: these launches are latency-bound, more (shorter)
  // workgroups beat fewer long ones
  const int s = skinny_splits(M, N, K, o);
  if ((size_t)nb * s * M * N > slab_floats) return hipErrorInvalidValue;
  P.nbatch = nb;
  for (int i = 0; i < nb; ++i) { P.Ab[i] = A[i]; P.Bb[i] = W[i]; }
  P.C = slab;
  P.c_rs = N;
  P.slab_stride = (long)M * N;
  P.slab_batch_stride = (long)s * M * N;
  *splits = s;
  return launch_gemm<64, 64, BKS, ASRC, BSRC, EPI_SLAB>(st, P, s);
}
hipError_t gemm_nt_batched_deferred(hipStream_t st, int nb, int M, int N, int K,
                                    const float* const* A, long lda, const float* const* W,
                                    long ldw, float* slab, size_t slab_floats, int* splits) {
  return batched_deferred<SRC_KC, SRC_KC>(st, nb, M, N, K, A, lda, W, ldw, slab, slab_floats, splits);
}
hipError_t gemm_nn_batched_deferred(hipStream_t st, int nb, int M, int N, int K,
                                    const float* const* A, long lda, const float* const* W,
                                    long ldw, float* slab, size_t slab_floats, int* splits) {
  return batched_deferred<SRC_KC, SRC_RC>(st, nb, M, N, K, A, lda, W, ldw, slab, slab_floats, splits);
}

// nb (<= 3) skinny problems that share A and K but have different widths / weights, in ONE
// launch; partials stay in the slab, problem i at slab + off[i] laid out [split][M][N[i]].
hipError_t gemm_nt_hetero_deferred(hipStream_t st, int nb, int M, int K, const float* A, long lda,
                                   const float* const* W, long ldw, const int* N, float* slab,
                                   size_t slab_floats, int* splits, size_t* off) {
  if (nb < 1 || nb > 3) return hipErrorInvalidValue;
  int nmax = 0, tiles = 0;
  size_t nsum = 0;
  for (int i = 0; i < nb; ++i) {
    nmax = N[i] > nmax ? N[i] : nmax;
    tiles += ((M + 63) / 64) * ((N[i] + 63) / 64);
    nsum += (size_t)N[i];
  }
  {
    const float* As[3] = {A, A, A};
    if (skinny_dma_ok(M, K, lda, ldw, false, nb, N, As, W)) {
      const int s = skinny_dma_splits(M, K, tiles, nsum, slab_floats);
      if ((size_t)s * M * nsum <= slab_floats) {
        long o2[3];
        size_t acc = 0;
        for (int i = 0; i < nb; ++i) {
          o2[i] = (long)acc;
          off[i] = acc;
          acc += (size_t)s * M * N[i];
        }
        *splits = s;
        return skinny_dma(st, false, nb, M, K, As, lda, W, ldw, N, slab, o2, s);
      }
    }
  }
#ifdef RAU_DEV_HOOKS
  if (std::getenv("RAU_LIN_TRACE"))
    std::fprintf(stderr, "hetero fallback nb=%d M=%d K=%d N=%d,%d,%d\n", nb, M, K, N[0], nb > 1 ? N[1] : 0, nb > 2 ? N[2] : 0);
#endif
  const int nk = (K + BKS - 1) / BKS;
  int s = (160 * nb + tiles - 1) / tiles;     // the merged launch stands for nb launches
  if (s > nk / 2) s = nk / 2;
  if (s < 1) s = 1;
  while (s > 1 && (size_t)s * M * nsum > slab_floats) --s;
  if ((size_t)s * M * nsum > slab_floats) return hipErrorInvalidValue;
  {
    const int per = (nk + s - 1) / s;
    s = (nk + per - 1) / per;
  }
  LinOpts o;
  GemmParams P = lin_params(M, nmax, K, A, lda, W[0], ldw, slab, nmax, o);
  P.nk = nk;
  P.nbatch = nb;
  size_t acc = 0;
  for (int i = 0; i < nb; ++i) {
    P.Ab[i] = A;
    P.Bb[i] = W[i];
    P.Nb[i] = N[i];
    P.slab_off[i] = (long)acc;
    off[i] = acc;
    acc += (size_t)s * M * N[i];
  }
  *splits = s;
  return launch_gemm<64, 64, BKS, SRC_KC, SRC_KC, EPI_SLAB>(st, P, s);
}

static int tn_splits(int M, int N, int K) {
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const int nk = (K + BK - 1) / BK;
  constexpr int target = 512;   // workgroups a Linear weight-gradient launch aims for
  int s = target / tiles;
  if (s > nk / 8) s = nk / 8;
  if (s < 1) s = 1;
  const int per = (nk + s - 1) / s;
  return (nk + per - 1) / per;
}
size_t gemm_tn_slab_floats(int M, int N, int K) {
  const int s = tn_splits(M, N, K);
  // product partials (when split) + per-split column sums of A (the bias gradient)
  return (s > 1 ? (size_t)s * M * N : 0) + (size_t)s * M;
}
hipError_t gemm_tn_acc(hipStream_t st, int M, int N, int K, const float* A, long lda,
                       const float* B, long ldb, float* C, long ldc, float* slab, float* dbias, int bf16) {
  LinOpts o;
  o.accumulate = 1;
  const int s = tn_splits(M, N, K);
  float* rs = dbias ? slab + (s > 1 ? (size_t)s * M * N : 0) : nullptr;
  hipError_t e;
  if (s <= 1) {
    GemmParams P = lin_params(M, N, K, A, lda, B, ldb, C, ldc, o);
    P.rs_out = rs;
    if (bf16)
      e = dbias ? launch_gemm<128, 128, BK, SRC_RC_SUM, SRC_RC, EPI_LIN, 1>(st, P, 1)
                : launch_gemm<128, 128, BK, SRC_RC, SRC_RC, EPI_LIN, 1>(st, P, 1);
    else
    e = dbias ? launch_gemm<128, 128, BK, SRC_RC_SUM, SRC_RC, EPI_LIN>(st, P, 1)
              : launch_gemm<128, 128, BK, SRC_RC, SRC_RC, EPI_LIN>(st, P, 1);
  } else {
    if (ldc != N) return hipErrorInvalidValue;
    GemmParams P = lin_params(M, N, K, A, lda, B, ldb, slab, N, o);
    P.slab_stride = (long)M * N;
    P.rs_out = rs;
    if (bf16)
      e = dbias ? launch_gemm<128, 128, BK, SRC_RC_SUM, SRC_RC, EPI_SLAB, 1>(st, P, s)
                : launch_gemm<128, 128, BK, SRC_RC, SRC_RC, EPI_SLAB, 1>(st, P, s);
    else
    e = dbias ? launch_gemm<128, 128, BK, SRC_RC_SUM, SRC_RC, EPI_SLAB>(st, P, s)
              : launch_gemm<128, 128, BK, SRC_RC, SRC_RC, EPI_SLAB>(st, P, s);
    if (e != hipSuccess) return e;
    e = splitk_reduce_acc(st, (size_t)M * N, s, slab, (size_t)M * N, C);
  }
  if (e != hipSuccess || !dbias) return e;
  return splitk_reduce_acc(st, (size_t)M, s, rs, (size_t)M, dbias);
}

// ---- grouped Linear weight gradients: dW_p += dY_p^T X_p (and db_p += column sums of dY_p) for
// all the Linears of a parameter group in ONE GEMM launch + ONE reduction launch.  Every problem
// shares the reduction length K (rows = hops x batch, or tokens x batch).  The K split is chosen
// for long K loops (>= 16 steps per workgroup) rather than for workgroup count: these launches
// run underneath the bulk conv GEMMs and the recurrence, so what counts is how little of the
// machine they take, not their own latency.
static int group_splits(const TnProblem* pr, int np, int K) {
  int tiles = 0;
  for (int i = 0; i < np; ++i) tiles += ((pr[i].M + 127) / 128) * ((pr[i].N + 127) / 128);
  const int nk = (K + BK - 1) / BK;
  constexpr int target = 0;   // 0: K split fitted to whole rounds of the resident workgroups (below)
  const int smax = std::max(1, nk / 16);
  int s = 1;
  if (target > 0) {
    s = (target + tiles / 2) / (tiles > 0 ? tiles : 1);
    s = std::min(std::max(s, 1), smax);
  } else {
    // the split that fills whole rounds of the 512 resident workgroups best (the encoder's group
    // runs alone at the end of the step: a 1.3-round launch wastes a third of it); ties -> fewer
    // splits (less slab traffic)
    double best = -1.0;
    for (int c = 1; c <= smax; ++c) {
      const int wgs = tiles * c;
      const double eff = (double)wgs / (512.0 * ((wgs + 511) / 512));
      if (eff > best + 0.02) { best = eff; s = c; }
    }
  }
  const int per = (nk + s - 1) / s;
  return (nk + per - 1) / per;
}
size_t gemm_tn_group_slab_floats(const TnProblem* pr, int np, int K) {
  const int s = group_splits(pr, np, K);
  size_t n = 0;
  for (int i = 0; i < np; ++i) n += (size_t)s * ((size_t)pr[i].M * pr[i].N + pr[i].M);
  return n;
}

struct GroupReduce {
  int np, splits;
  long e0[kGroupMax + 1];        // prefix sums of M*N
  long b0[kGroupMax + 1];        // prefix sums of M (bias gradients)
  const float* slab[kGroupMax]; const float* rs[kGroupMax];
  float* dst[kGroupMax]; float* db[kGroupMax];
  long mn[kGroupMax]; int m[kGroupMax];
};
__global__ void k_group_reduce_acc(const GroupReduce R) {
  const long nW = R.e0[R.np], nB = R.b0[R.np];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nW + nB;
       i += (long)gridDim.x * blockDim.x) {
    const bool bias = i >= nW;
    const long j = bias ? i - nW : i;
    const long* pre = bias ? R.b0 : R.e0;
    int p = 0;
    while (p + 1 < R.np && j >= pre[p + 1]) ++p;
    const long k = j - pre[p];
    if (bias) {
      if (!R.db[p]) continue;
      float v = R.db[p][k];
      for (int s = 0; s < R.splits; ++s) v += R.rs[p][(long)s * R.m[p] + k];
      R.db[p][k] = v;
    } else {
      float v = R.dst[p][k];
      for (int s = 0; s < R.splits; ++s) v += R.slab[p][(long)s * R.mn[p] + k];
      R.dst[p][k] = v;
    }
  }
}

hipError_t gemm_tn_group_acc(hipStream_t st, const TnProblem* pr, int np, int K, float* slab,
                             size_t slab_floats, int bf16) {
  if (np < 1 || np > kGroupMax) return hipErrorInvalidValue;
  if (gemm_tn_group_slab_floats(pr, np, K) > slab_floats) return hipErrorInvalidValue;
  const int s = group_splits(pr, np, K);
  GroupParams G{};
  GroupReduce R{};
  G.np = np; G.K = K;
  G.nk = (K + BK - 1) / BK;
  G.nk_per_split = (G.nk + s - 1) / s;
  R.np = np; R.splits = s;
  float* w = slab;
  int wg = 0;
  for (int i = 0; i < np; ++i) {
    GroupProb& q = G.p[i];
    q.A = pr[i].A; q.lda = pr[i].lda; q.B = pr[i].B; q.ldb = pr[i].ldb;
    q.M = pr[i].M; q.N = pr[i].N;
    q.tiles_m = (q.M + 127) / 128; q.tiles_n = (q.N + 127) / 128;
    q.slab = w;
    w += (size_t)s * q.M * q.N;
    q.rs_out = w;
    w += (size_t)s * q.M;
    G.wg0[i] = wg;
    wg += q.tiles_m * q.tiles_n * s;
    R.slab[i] = q.slab; R.rs[i] = q.rs_out; R.dst[i] = pr[i].C; R.db[i] = pr[i].dbias;
    R.mn[i] = (long)q.M * q.N; R.m[i] = q.M;
    R.e0[i + 1] = R.e0[i] + R.mn[i];
    R.b0[i + 1] = R.b0[i] + q.M;
  }
  G.wg0[np] = wg;
  if (bf16)   // RAU_BF16 mode: both operands rounded to bf16 while staged, f32 accumulate; bias sums stay f32
    hipLaunchKernelGGL((gemm_group_kernel<128, 128, BK, SRC_RC_SUM, SRC_RC, 1>), dim3(wg), dim3(256), 0,
                       st, G);
  else
  hipLaunchKernelGGL((gemm_group_kernel<128, 128, BK, SRC_RC_SUM, SRC_RC>), dim3(wg), dim3(256), 0,
                     st, G);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const long n = R.e0[np] + R.b0[np];
  const int blocks = (int)std::min<long>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(k_group_reduce_acc, dim3(blocks), dim3(256), 0, st, R);
  return hipGetLastError();
}

}  // namespace rau
