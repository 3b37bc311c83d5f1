// Split-K partial products of the recurrence's Linear GEMMs (LSTM gates, hop projections, their
// input gradients): C[M,N] = A[M,K] W^T (W stored [N][K]) or A[M,K] W (W stored [K][N]), M = batch
// (<= 256), N 512..2048, K 512..2048.  Reference ops: model/DeepLSTM.lua:29-65 (i2h / h2h Linear),
// train_vqa_RAU_SS.lua:448-462, 581-596 (encoder forward / backward through time).
//
// These launches sit on the step's critical path, ~100 per step, 14 us alone / 28 us in
// the step with the register-staged 64x64x16 single-stage tile of gemm_core.h (one global round
// trip and two barriers per 16-deep K-step; every XCD fetching all of W through the fabric).  Here:
//  * operands go HBM/L2 -> LDS by DMA (global_load_lds_dwordx4) into a ring of KNST stages, one
//    barrier per stage, fragments of stage s+1 read while stage s's MFMAs run.  KNST = 2 (16 KB):
//    in the step a 4-slot ring (32 KB) measures the same and an 8-slot ring (64 KB) +0.5 ms -- the
//    workgroups must fit next to the resident bulk tiles, and the request queue of the bulk
//    kernels' own DMA, not this kernel's prefetch depth, sets the latency a stage sees;
//  * work item g = (problem, K split, tile column, tile row) in that order, dealt to the XCDs in
//    contiguous runs (workgroup L -> XCD L & 7 -> items [ (L & 7) * per, ... )), so that one XCD
//    works on one K slice (or a column range of it): each L2 fetches its slice of W once;
//  * the LDS image is DMA's lane-linear one; the XOR swizzle that makes the fragment reads
//    conflict-free is applied on the GLOBAL side (which 16 bytes a lane fetches), not the LDS side.
// Output: raw partial sums to the slab [split][M][N]; lin_reduce_epilogue / the LSTM cell kernels
// add them in split order (deterministic), exactly as for the tile this replaces.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace rau {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int KT = 64;            // tile rows and columns
constexpr int KBK = 16;           // K-step per stage
constexpr int KPART = KT * KBK;   // floats per operand per stage (4 KB)
constexpr int KSTAGE = 2 * KPART; // 8 KB
constexpr int KNST = 2;           // ring slots (power of two)

struct SkinnyParams {
  int M, K, nprob, splits, nst;   // nst = K-steps per split (even)
  int tiles_m, tiles_n;           // tiles_n of the widest problem
  const float* A[3]; const float* B[3];
  int N[3]; long off[3];          // problem p: N[p] columns, partials at slab + off[p] ([split][M][N[p]])
  long lda, ldb;
  float* slab;
};

template <int OFF>
__device__ __forceinline__ void lds_read128(f32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read32(float& dst, uint32_t addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// 16-byte column swizzle of the [row][16 k] image (64-byte rows): the four rows a ds_read_b128
// lane group takes on one 64-byte bank quarter get four different columns
__device__ __forceinline__ int kc_swz(int row) { return (-((row & 15) >> 2)) & 3; }

// BRC = false: W stored [N][K] (k contiguous);  true: W stored [K][N] (n contiguous)
template <bool BRC>
__global__ __launch_bounds__(256) void k_skinny_dma(const SkinnyParams P) {
  RAU_CHAIN_PRIO();
  __shared__ __attribute__((aligned(16))) float smem[KNST * KSTAGE];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 1, wn = w >> 1;

  // ---- work item
  const int tiles = P.tiles_m * P.tiles_n;
  const int total = P.nprob * P.splits * tiles;
  int g = blockIdx.x;
  if ((total & 7) == 0) g = (g & 7) * (total >> 3) + (g >> 3);
  const int tm = g % P.tiles_m;
  g /= P.tiles_m;
  const int tn = g % P.tiles_n;
  g /= P.tiles_n;
  const int split = g % P.splits, prob = g / P.splits;
  const int PN = P.N[prob];
  const int m0 = tm * KT, n0 = tn * KT;
  if (n0 >= PN) return;                      // narrower problem of a merged launch (whole workgroup)
  const int nk = P.K / KBK;
  const int s0 = split * P.nst;
  int nst = nk - s0;
  if (nst > P.nst) nst = P.nst;
  if (nst <= 0) return;

  // ---- DMA sources: one 16-byte piece of each operand per lane and stage
  const char *ga, *gb;
  long stepb;
  {
    const int p = tid, row = p >> 2, c = (p & 3) ^ kc_swz(row);
    int r = m0 + row;
    if (r > P.M - 1) r = P.M - 1;            // rows past M: a duplicate, never stored
    ga = reinterpret_cast<const char*>(P.A[prob] + (long)r * P.lda + (long)s0 * KBK + c * 4);
    if (!BRC) {
      int n = n0 + row;
      if (n > PN - 1) n = PN - 1;
      gb = reinterpret_cast<const char*>(P.B[prob] + (long)n * P.ldb + (long)s0 * KBK + c * 4);
      stepb = KBK * 4;
    } else {
      const int k = p >> 4, cc = (p & 15) ^ (((k >> 2) & 1) << 2);
      gb = reinterpret_cast<const char*>(P.B[prob] + ((long)s0 * KBK + k) * P.ldb + n0 + cc * 4);
      stepb = (long)KBK * P.ldb * 4;
    }
  }
  int issued = 0;
  auto issue = [&]() {   // next stage, into ring slot issued % 4
    float* dst = smem + (issued & (KNST - 1)) * KSTAGE + w * 256;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)ga, (lds_ptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)gb, (lds_ptr_t)(dst + KPART), 16, 0, 0);
    ga += KBK * 4;
    gb += stepb;
    ++issued;
  };

  // ---- fragment addresses (bytes, stage 0).  Lane (r = l & 15, kk = l >> 4) of MFMA e (0..3) of
  // a stage holds k = 4 kk + e: one ds_read_b128 per 16-row block covers the stage for [row][k]
  // operands; [k][n] operands take a ds_read_b32 per MFMA and block.
  const int fr = l & 15, kk = l >> 4;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  uint32_t fa[2], fb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 32 + i * 16 + fr;
    fa[i] = lds0 + (uint32_t)(ra * 64 + ((kk ^ kc_swz(ra)) << 4));
    if (!BRC) {
      const int rb = wn * 32 + i * 16 + fr;
      fb[i] = lds0 + (uint32_t)(KPART * 4 + rb * 64 + ((kk ^ kc_swz(rb)) << 4));
    } else {
      const int nb = (wn * 32 + i * 16 + fr) ^ ((kk & 1) << 4);
      fb[i] = lds0 + (uint32_t)(KPART * 4 + (4 * kk * 64 + nb) * 4);
    }
  }

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // [set][block]; element e = MFMA e's value.  [k][n] operands: scalars, so that each ds_read_b32
  // lands in the register the MFMA reads (no compiler-made copy ahead of the wait)
  f32x4 af[2][2], bq[2][2];
  float bs[2][2][4];

  auto read_frags = [&](int set, int slot) {
    const uint32_t so = (uint32_t)slot * (KSTAGE * 4);
    lds_read128<0>(af[set][0], fa[0] + so);
    lds_read128<0>(af[set][1], fa[1] + so);
    if constexpr (!BRC) {
      lds_read128<0>(bq[set][0], fb[0] + so);
      lds_read128<0>(bq[set][1], fb[1] + so);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        lds_read32<0>(bs[set][j][0], fb[j] + so);
        lds_read32<256>(bs[set][j][1], fb[j] + so);
        lds_read32<512>(bs[set][j][2], fb[j] + so);
        lds_read32<768>(bs[set][j][3], fb[j] + so);
      }
    }
  };
  // wait until the stage with `later` stages issued after it has landed (this wave's pieces)
  auto wait_landed = [&](int later) {
    if (later > KNST - 2) later = KNST - 2;
    switch (later) {
      case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };

  // ---- prologue: three stages in flight, stage 0's fragments in set 0
  for (int s = 0; s < KNST - 1 && s < nst; ++s) issue();
  wait_landed(issued - 1);
  __builtin_amdgcn_s_barrier();
  if (issued < nst) issue();
  read_frags(0, 0);

  auto body = [&](auto set_tag, int s) {
    constexpr int set = decltype(set_tag)::value;
    const bool more = s + 1 < nst;
    if (more) wait_landed(issued - 1 - (s + 1));
    // fragments of stage s are in registers (every consumer below depends on this wait)
    if constexpr (!BRC)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(af[set][0]), "+v"(af[set][1]), "+v"(bq[set][0]), "+v"(bq[set][1])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(af[set][0]), "+v"(af[set][1]), "+v"(bs[set][0][0]), "+v"(bs[set][0][1]),
                     "+v"(bs[set][0][2]), "+v"(bs[set][0][3]), "+v"(bs[set][1][0]),
                     "+v"(bs[set][1][1]), "+v"(bs[set][1][2]), "+v"(bs[set][1][3])
                   :
                   : "memory");
    __builtin_amdgcn_sched_barrier(0);
    // stage s+1 is visible to all; every wave is done with stage s's slot
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (issued < nst) issue();
    if (more) read_frags(set ^ 1, (s + 1) & (KNST - 1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
              af[set][i][e], BRC ? bs[set][j][e] : bq[set][j][e], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll 1
  for (int s = 0; s < nst; s += 2) {
    body(std::integral_constant<int, 0>{}, s);
    body(std::integral_constant<int, 1>{}, s + 1);
  }

  // ---- partial sums to the slab: D[row 4 (l >> 4) + r][col l & 15]
  float* C = P.slab + P.off[prob] + (long)split * P.M * PN;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 32 + i * 16 + 4 * kk + r;
      if (m < P.M) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn * 32 + j * 16 + fr;
          if (n < PN) C[(long)m * PN + n] = acc[i][j][r];
        }
      }
    }
}

}  // namespace

bool skinny_dma_ok(int M, int K, long lda, long ldb, bool brc, int nprob, const int* N,
                   const float* const* A, const float* const* B) {
  static const bool off = std::getenv("RAU_SKINNY_DMA_OFF") != nullptr;
  if (off || M < 1 || nprob < 1 || nprob > 3) return false;
  if (K % (2 * KBK) != 0 || (lda & 3) || (ldb & 3)) return false;
  for (int p = 0; p < nprob; ++p) {
    if (N[p] < 1) return false;
    if (brc && (N[p] % KT) != 0) return false;     // [K][N] rows are read 64 columns at a time
    if ((reinterpret_cast<uintptr_t>(A[p]) & 15) || (reinterpret_cast<uintptr_t>(B[p]) & 15)) return false;
  }
  return true;
}

// K splits: about one workgroup per CU (256), an even number of K-steps per split
int skinny_dma_splits(int M, int K, int tiles_all, size_t cols_all, size_t slab_floats) {
  const int nk = K / KBK;
  int s = (256 + tiles_all / 2) / tiles_all;   // 160 .. 512 measured equal in the step
  if (s < 1) s = 1;
  if (s > nk / 2) s = nk / 2;
  while (s > 1 && (size_t)s * M * cols_all > slab_floats) --s;
  int per = (nk + s - 1) / s;
  per += per & 1;
  return (nk + per - 1) / per;
}

hipError_t skinny_dma(hipStream_t st, bool brc, int nprob, int M, int K, const float* const* A,
                      long lda, const float* const* B, long ldb, const int* N, float* slab,
                      const long* off, int splits) {
  SkinnyParams P{};
  P.M = M; P.K = K; P.nprob = nprob; P.splits = splits;
  const int nk = K / KBK;
  int per = (nk + splits - 1) / splits;
  per += per & 1;
  P.nst = per;
  if ((long)per * splits < nk) return hipErrorInvalidValue;
  int nmax = 0;
  for (int p = 0; p < nprob; ++p) {
    P.A[p] = A[p]; P.B[p] = B[p]; P.N[p] = N[p]; P.off[p] = off[p];
    nmax = N[p] > nmax ? N[p] : nmax;
  }
  P.tiles_m = (M + KT - 1) / KT;
  P.tiles_n = (nmax + KT - 1) / KT;
  P.lda = lda; P.ldb = ldb; P.slab = slab;
  const int grid = nprob * splits * P.tiles_m * P.tiles_n;
  if (brc) hipLaunchKernelGGL(k_skinny_dma<true>, dim3(grid), dim3(256), 0, st, P);
  else hipLaunchKernelGGL(k_skinny_dma<false>, dim3(grid), dim3(256), 0, st, P);
  return hipGetLastError();
}

}  // namespace rau
