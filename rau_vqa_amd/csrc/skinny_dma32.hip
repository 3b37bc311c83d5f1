// skinny_dma32.hip -- skinny_dma.hip's split-K partial products with 32-DEEP stages (round 3, second
// session): a stage is 128 bytes of every [row][k] operand row (whole lines, 8 pieces with an 8-column
// XOR swizzle) and two MFMA k-groups, so a workgroup passes half as many waits and barriers per K; 32 KB
// of LDS and 90 registers instead of 16 KB and 36.  Stand-alone 10-18 % faster on the recurrence's
// shapes (tools/linbench).  In the step it pays where the recurrence is the longer path (bf16 mode
// -1.1 %, 64-sample contexts -1.9 %, evaluate-mode forward +1.5-3.5 %) and costs the f32 256-sample
// step 0.9 % (its footprint beside the bulk tiles), so gemm_lin.hip picks it by the caller's policy
// (skinny_dma_set_deep, set by the step-level entry points from side_split()'s predicate).
// Reference ops as in skinny_dma.hip.
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
constexpr int KBK = 32;           // K-step per stage: 128 bytes of every [row][k] operand row = whole lines
constexpr int KPART = KT * KBK;   // floats per operand per stage (8 KB)
constexpr int KSTAGE = 2 * KPART; // 16 KB
constexpr int KNST = 2;           // ring slots (power of two)
constexpr int KDMA = 4;           // DMA instructions per wave and stage (two per operand)

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

// 16-byte column swizzle of the [row][32 k] image (128-byte rows, 8 pieces): piece p of row r sits in
// column p ^ ((r >> 1) & 7), so the sixteen rows a ds_read_b128 lane group takes at one k-group hit
// sixteen different 16-byte bank groups
__device__ __forceinline__ int kc_swz(int row) { return (row >> 1) & 7; }

// BRC = false: W stored [N][K] (k contiguous);  true: W stored [K][N] (n contiguous)
template <bool BRC>
__global__ __launch_bounds__(256) void k_skinny_dma32(const SkinnyParams P) {
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

  // ---- DMA sources: two 16-byte pieces of each operand per lane and stage (pieces tid and tid + 256)
  const char *ga[2], *gb[2];
  long stepb;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = tid + 256 * j, row = p >> 3, c = (p & 7) ^ kc_swz(row);
    int r = m0 + row;
    if (r > P.M - 1) r = P.M - 1;            // rows past M: a duplicate, never stored
    ga[j] = reinterpret_cast<const char*>(P.A[prob] + (long)r * P.lda + (long)s0 * KBK + c * 4);
    if (!BRC) {
      int n = n0 + row;
      if (n > PN - 1) n = PN - 1;
      gb[j] = reinterpret_cast<const char*>(P.B[prob] + (long)n * P.ldb + (long)s0 * KBK + c * 4);
      stepb = KBK * 4;
    } else {
      const int k = p >> 4, cc = (p & 15) ^ (((k >> 2) & 1) << 2);
      gb[j] = reinterpret_cast<const char*>(P.B[prob] + ((long)s0 * KBK + k) * P.ldb + n0 + cc * 4);
      stepb = (long)KBK * P.ldb * 4;
    }
  }
  int issued = 0;
  auto issue = [&]() {   // next stage, into ring slot issued % KNST
    float* dst = smem + (issued & (KNST - 1)) * KSTAGE + w * 256;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)ga[j], (lds_ptr_t)(dst + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)gb[j], (lds_ptr_t)(dst + KPART + j * 1024), 16, 0, 0);
      ga[j] += KBK * 4;
      gb[j] += stepb;
    }
    ++issued;
  };

  // ---- fragment addresses (bytes, stage 0).  Lane (r = l & 15, kk = l >> 4): MFMA e (0..3) of the
  // stage's half h (0, 1) holds k = 16 h + 4 kk + e: two ds_read_b128 per 16-row block cover the stage
  // for [row][k] operands (pieces kk and kk + 4 of the row); [k][n] operands take a ds_read_b32 per MFMA
  // and block.
  const int fr = l & 15, kk = l >> 4;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  uint32_t fa[2][2], fb[2][2];   // [block][half]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wm * 32 + i * 16 + fr;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      fa[i][h] = lds0 + (uint32_t)(ra * 128 + (((kk + 4 * h) ^ kc_swz(ra)) << 4));
    if (!BRC) {
      const int rb = wn * 32 + i * 16 + fr;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        fb[i][h] = lds0 + (uint32_t)(KPART * 4 + rb * 128 + (((kk + 4 * h) ^ kc_swz(rb)) << 4));
    } else {
      const int nb = (wn * 32 + i * 16 + fr) ^ ((kk & 1) << 4);
      fb[i][0] = lds0 + (uint32_t)(KPART * 4 + (4 * kk * 64 + nb) * 4);
      fb[i][1] = fb[i][0] + 16 * 64 * 4;
    }
  }

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // [set][block]; element e = MFMA e's value.  [k][n] operands: scalars, so that each ds_read_b32
  // lands in the register the MFMA reads (no compiler-made copy ahead of the wait)
  f32x4 af[2][2][2], bq[2][2][2];   // [set][block][half]
  float bs[2][2][8];

  auto read_frags = [&](int set, int slot) {
    const uint32_t so = (uint32_t)slot * (KSTAGE * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      lds_read128<0>(af[set][0][h], fa[0][h] + so);
      lds_read128<0>(af[set][1][h], fa[1][h] + so);
      if constexpr (!BRC) {
        lds_read128<0>(bq[set][0][h], fb[0][h] + so);
        lds_read128<0>(bq[set][1][h], fb[1][h] + so);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          lds_read32<0>(bs[set][j][4 * h + 0], fb[j][h] + so);
          lds_read32<256>(bs[set][j][4 * h + 1], fb[j][h] + so);
          lds_read32<512>(bs[set][j][4 * h + 2], fb[j][h] + so);
          lds_read32<768>(bs[set][j][4 * h + 3], fb[j][h] + so);
        }
      }
    }
  };
  // wait until the stage with `later` stages issued after it has landed (this wave's pieces)
  auto wait_landed = [&](int later) {
    if (later > KNST - 2) later = KNST - 2;
    switch (later) {   // KDMA instructions per stage
      case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
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
                   : "+v"(af[set][0][0]), "+v"(af[set][0][1]), "+v"(af[set][1][0]), "+v"(af[set][1][1]),
                     "+v"(bq[set][0][0]), "+v"(bq[set][0][1]), "+v"(bq[set][1][0]), "+v"(bq[set][1][1])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(af[set][0][0]), "+v"(af[set][0][1]), "+v"(af[set][1][0]), "+v"(af[set][1][1]),
                     "+v"(bs[set][0][0]), "+v"(bs[set][0][1]), "+v"(bs[set][0][2]), "+v"(bs[set][0][3]),
                     "+v"(bs[set][0][4]), "+v"(bs[set][0][5]), "+v"(bs[set][0][6]), "+v"(bs[set][0][7]),
                     "+v"(bs[set][1][0]), "+v"(bs[set][1][1]), "+v"(bs[set][1][2]), "+v"(bs[set][1][3]),
                     "+v"(bs[set][1][4]), "+v"(bs[set][1][5]), "+v"(bs[set][1][6]), "+v"(bs[set][1][7])
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
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                af[set][i][h][e], BRC ? bs[set][j][4 * h + e] : bq[set][j][h][e], acc[i][j], 0, 0, 0);
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

bool skinny_dma32_ok(int M, int K, long lda, long ldb, bool brc, int nprob, const int* N,
                   const float* const* A, const float* const* B) {
  if (M < 1 || nprob < 1 || nprob > 3) return false;
  if (K % (2 * KBK) != 0 || (lda & 3) || (ldb & 3)) return false;
  for (int p = 0; p < nprob; ++p) {
    if (N[p] < 1) return false;
    if (brc && (N[p] % KT) != 0) return false;     // [K][N] rows are read 64 columns at a time
    if ((reinterpret_cast<uintptr_t>(A[p]) & 15) || (reinterpret_cast<uintptr_t>(B[p]) & 15)) return false;
  }
  return true;
}

// K splits: about one workgroup per CU (256), an even number of K-steps per split
int skinny_dma32_splits(int M, int K, int tiles_all, size_t cols_all, size_t slab_floats) {
  const int nk = K / KBK;
  int s = (256 + tiles_all / 2) / tiles_all;   // 160 .. 512 measured equal in the step
  if (s < 1) s = 1;
  if (s > nk / 2) s = nk / 2;
  while (s > 1 && (size_t)s * M * cols_all > slab_floats) --s;
  int per = (nk + s - 1) / s;
  per += per & 1;
  return (nk + per - 1) / per;
}

hipError_t skinny_dma32(hipStream_t st, bool brc, int nprob, int M, int K, const float* const* A,
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
  if (brc) hipLaunchKernelGGL(k_skinny_dma32<true>, dim3(grid), dim3(256), 0, st, P);
  else hipLaunchKernelGGL(k_skinny_dma32<false>, dim3(grid), dim3(256), 0, st, P);
  return hipGetLastError();
}

}  // namespace rau
