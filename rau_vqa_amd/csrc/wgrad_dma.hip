// wgrad_dma.hip -- weight gradients of the two 1x1 convolutions on 14 x 14 maps, operands by LDS-DMA.
//
//   dW[ra, rb] += sum_{b, s} Aop[b, ra, s] * Bop[b, rb, s]
//
// (dWp[k, m] += sum dS[b,k,s] I[b,m,s], reference SS:247 backward;  dWi[m, d] += sum dZ[b,m,s] X'[b,d,s],
//  SS:240 backward, with dZ = dI (1 - I^2) already formed by the dgrad's epilogue.)
//
// Both operands are stored [sample][row][position] with the REDUCTION index contiguous, so a tile's
// rows can go to LDS exactly as they lie in memory -- no transposing ds_write pass (round 1-2:
// scalar stores of every float4, one pass per K-step) and no staging registers: a stage is the
// 28-position chunk (196 = 7 x 28: no padded MFMA work) of 128 rows of each operand, 896 pieces of
// 16 bytes per operand = 14 + 14 `global_load_lds_dwordx4` per stage, seven per wave, next stage in
// flight under the current one's 112 MFMAs per wave (two stages, one barrier each).
// LDS image [row][28] (pitch 28 floats = 7 pieces): an MFMA fragment is lane (row = l & 15, k group
// g = l >> 4); read as 8 bytes -- k = 8 kk + 2 g + {0, 1} -- the 32 lanes of a half-wave touch 16
// whole pieces whose indices 7 row + 2 kk are distinct mod 16: conflict-free ds_read_b64, three of
// them cover 24 positions with two MFMAs each, the last 4 positions take a ds_read_b32 (2-way
// conflict, one read in seven).  The k ORDER inside a dot product therefore differs from the
// register-staged kernel's (pairs interleaved): same values to f32 rounding, not bitwise.
// 4 waves x (64 x 64): 16 accumulator blocks of v_mfma_f32_16x16x4_f32.  Whole samples per K split,
// a split's tiles on one XCD (as gemm_split_xcd_kernel), partial tiles to slabs, fixed-order
// reduction by the caller: deterministic.
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace rau {

namespace {

constexpr int GS = 196;               // positions per sample
constexpr int GCH = 28;               // positions per stage
constexpr int GNCH = GS / GCH;        // 7 chunks per sample
constexpr int GT = 128;               // tile rows of each operand
constexpr int GPART = GT * GCH;       // floats of one operand's part of a stage (3584)
constexpr int GSTAGE = 2 * GPART;     // 7168 floats = 28672 bytes
constexpr int GNST = 2;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct WgradParams {
  int ra, rb, nB, tiles_a, tiles_b, splits, spb;   // spb = samples per split
  const float* A; long a_bs;                      // [b][ra][S]
  const float* B; long b_bs;                      // [b][rb][S]
  float* slab;                                     // [split][ra][rb]
};

template <class F, int... I>
__device__ __forceinline__ void gfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void gfor(F&& f) { gfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <int OFF>
__device__ __forceinline__ void lds_read64(float2& dst, uint32_t addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read32(float& dst, uint32_t addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// NG = wave groups per workgroup.  NG = 1: four waves, two workgroups per CU (round 3).  NG = 2 (round 4):
// EIGHT waves = two groups of four that split the workgroup's K range between them (group g takes the
// stages g, g + 2, ..: the same stage count up to one, one barrier per stage for all eight waves), each with
// its own two-stage ring and its own 128 x 128 accumulators; at the end group 1 hands its accumulators to
// group 0 through its ring's LDS and group 0 writes ONE partial tile.  Same waves per SIMD, registers and
// LDS per CU as two 4-wave workgroups -- and half the split-K partials: 64 splits x 8 tiles of the
// attention weight gradient were 34 MB of slab writes + reads per launch for a 0.5 MB result (HBM bytes
// 1.26x the algorithmic ones in the round-3/4 PMC passes), and the reduction kernel behind every launch
// sums half as many.  The partials are still summed in a fixed order: deterministic.
template <int NG>
__global__ __launch_bounds__(256 * NG, 2) void k_wgrad_dma(const WgradParams P) {
  __shared__ __attribute__((aligned(16))) float smem_all[NG * GNST * GSTAGE];
  const int tid = threadIdx.x, l = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = NG == 1 ? 0 : wv >> 2;       // wave group
  const int w = wv & 3;                        // wave within its group
  float* smem = smem_all + grp * (GNST * GSTAGE);
  const int wm = w & 1, wn = w >> 1;
  // a K split's tiles sit on one XCD (they stream the same samples): workgroup L -> XCD r = L % 8,
  // q = L / 8, split = r + 8 (q / tiles), tile = q % tiles
  const int tiles = P.tiles_a * P.tiles_b;
  const int L = blockIdx.x, r8 = L & 7, q8 = L >> 3;
  const int split = r8 + 8 * (q8 / tiles), tile = q8 % tiles;
  if (split >= P.splits) return;
  const int ta = tile % P.tiles_a, tb = tile / P.tiles_a;
  const int b_lo = split * P.spb;
  int b_hi = b_lo + P.spb;
  if (b_hi > P.nB) b_hi = P.nB;
  const int nst_all = (b_hi - b_lo) * GNCH;            // stages of this workgroup's K range
  const int nst = (nst_all - grp + NG - 1) / NG;       // ... of this wave group: grp, grp + NG, ..
  const int nloop = (nst_all + NG - 1) / NG;           // barrier rounds (the same for every wave)

  // ---- DMA slots: instruction i = w + 4 n (n = 0..6) of the stage's 28; i < 14: operand A pieces
  // [64 i, 64 i + 64), else operand B.  Piece p of a part = (row p / 7, 16-byte column p % 7).
  uint32_t voff[7];
#pragma unroll
  for (int n = 0; n < 7; ++n) {
    const int i = w + 4 * n;
    const int p = 64 * (i < 14 ? i : i - 14) + l;
    voff[n] = (uint32_t)((p / 7) * GS * 4 + (p % 7) * 16);
  }
  const char* a0 = reinterpret_cast<const char*>(P.A + (size_t)b_lo * P.a_bs + (size_t)ta * GT * GS);
  const char* b0 = reinterpret_cast<const char*>(P.B + (size_t)b_lo * P.b_bs + (size_t)tb * GT * GS);
  // stage s = (sample s / 7, chunk s % 7); its seven loads per wave go out one at a time, between the
  // MFMA groups of the stage before (a burst of seven costs the issuing wave 400-700 cycles)
  const char *ap = nullptr, *bp = nullptr;
  float* dst = nullptr;
  auto stage_ptrs = [&](int sg) {            // sg: the group's own stage counter
    const int s = NG * sg + grp;
    const int sb = s / GNCH, ch = s - sb * GNCH;
    ap = a0 + ((size_t)sb * P.a_bs + ch * GCH) * 4;
    bp = b0 + ((size_t)sb * P.b_bs + ch * GCH) * 4;
    dst = smem + (sg & 1) * GSTAGE;
  };
  auto issue_slot = [&](int n) {
    const int i = w + 4 * n;
    uint32_t vo = voff[n];
    asm volatile("" : "+v"(vo));
    __builtin_amdgcn_global_load_lds((glb_ptr_t)((i < 14 ? ap : bp) + vo),
                                     (lds_ptr_t)(dst + (i < 14 ? 0 : GPART) + 256 * (i < 14 ? i : i - 14)),
                                     16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int row = l & 15, g = l >> 4;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  // fragment addresses (bytes) of block 0 of this wave: 8-byte reads at k = 8 kk + 2 g, 4-byte at 24 + g
  const uint32_t fa64 = lds0 + (uint32_t)((wm * 64 + row) * GCH + 2 * g) * 4;
  const uint32_t fb64 = lds0 + (uint32_t)(GPART + (wn * 64 + row) * GCH + 2 * g) * 4;
  const uint32_t fa32 = lds0 + (uint32_t)((wm * 64 + row) * GCH + 24 + g) * 4;
  const uint32_t fb32 = lds0 + (uint32_t)(GPART + (wn * 64 + row) * GCH + 24 + g) * 4;
  constexpr int BLK = 16 * GCH * 4;   // bytes between consecutive 16-row blocks

  if (nst > 0) {
    stage_ptrs(0);
#pragma unroll
    for (int n = 0; n < 7; ++n) issue_slot(n);
  }
#pragma unroll 1
  for (int s = 0; s < nloop; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of stage s has landed
    __builtin_amdgcn_s_barrier();                      // everyone's has; everyone is done with stage s-1
    if (NG > 1 && s >= nst) continue;                  // the shorter group's last round: barrier only
    const bool more = s + 1 < nst;
    if (more) stage_ptrs(s + 1);
    const uint32_t so = (uint32_t)((s & 1) * GSTAGE) * 4;
    float2 a2[2][4], b2[2][4];
    float a1[4], b1[4];
    // round kk: fragments of k = 8 kk + 2 g + {0, 1}; rounds 0..2, then the 4-k tail
    auto rd64 = [&](auto kk_tag, int set) {
      constexpr int KK = decltype(kk_tag)::value;
      gfor<4>([&](auto it) {
        constexpr int i = decltype(it)::value;
        lds_read64<i * BLK + KK * 32>(a2[set][i], fa64 + so);
        lds_read64<i * BLK + KK * 32>(b2[set][i], fb64 + so);
      });
    };
    // 16 MFMAs on component .x, one DMA slot, 16 on .y, one DMA slot
    auto mma64 = [&](int set, int slot) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[set][i].x, b2[set][j].x, acc[i][j], 0, 0, 0);
      if (more) issue_slot(slot);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[set][i].y, b2[set][j].y, acc[i][j], 0, 0, 0);
      if (more) issue_slot(slot + 1);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    rd64(K0{}, 0);
    rd64(K1{}, 1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // round 0 is in
    __builtin_amdgcn_sched_barrier(0);
    mma64(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    rd64(K2{}, 0);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // round 1 is in
    __builtin_amdgcn_sched_barrier(0);
    mma64(1, 2);
    __builtin_amdgcn_sched_barrier(0);
    gfor<4>([&](auto it) {
      constexpr int i = decltype(it)::value;
      lds_read32<i * BLK>(a1[i], fa32 + so);
      lds_read32<i * BLK>(b1[i], fb32 + so);
    });
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // round 2 is in
    __builtin_amdgcn_sched_barrier(0);
    mma64(0, 4);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
    if (more) issue_slot(6);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- NG = 2: group 1's accumulators to group 0 through LDS, lane-linear: 4 waves x 16 blocks x 64 lanes x
  // 16 bytes = 64 KB, more than one ring (56 KB), so the exchange area starts at the workgroup's LDS base and
  // runs into the second ring (nobody reads a ring after the barrier; no DMA is in flight)
  if (NG > 1) {
    static_assert(NG == 1 || (size_t)4 * 16 * 64 * 16 <= (size_t)NG * GNST * GSTAGE * 4, "exchange area fits");
    __builtin_amdgcn_s_barrier();
    f32x4* xch = reinterpret_cast<f32x4*>(smem_all);
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) xch[((w * 16 + i * 4 + j) << 6) + l] = acc[i][j];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 o = xch[((w * 16 + i * 4 + j) << 6) + l];
        acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
      }
  }
  // ---- partial tile to this split's slab: block (i, j), register r = C[ra0 + 4 g + r][rb0 + row]
  float* C = P.slab + (size_t)split * P.ra * P.rb;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cb = tb * GT + wn * 64 + j * 16 + row;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ca = ta * GT + wm * 64 + i * 16 + 4 * g + r;
        C[(size_t)ca * P.rb + cb] = acc[i][j][r];
      }
    }
}

}  // namespace

// Shapes it takes: 14 x 14 maps, both row counts multiples of 128.
bool wgrad_dma_ok(int ra, int rb, int S) {
  static const bool off = std::getenv("RAU_WGRAD_DMA_OFF") != nullptr;   // A/B knob
  return !off && S == GS && ra % GT == 0 && rb % GT == 0;
}

// dW[ra, rb] += sum_{b,s} A[b,ra,s] B[b,rb,s]; slab: >= splits * ra * rb floats (conv_wgrad_slab_floats)
hipError_t wgrad_dma(hipStream_t st, int nB, int ra, int rb, int S, const float* A, long a_bs,
                     const float* B, long b_bs, float* dW, float* slab, int splits) {
  if (!wgrad_dma_ok(ra, rb, S) || nB < 1 || splits < 1) return hipErrorInvalidValue;
  WgradParams P{};
  P.ra = ra; P.rb = rb; P.nB = nB;
  P.tiles_a = ra / GT; P.tiles_b = rb / GT;
  P.spb = (nB + splits - 1) / splits;
  P.splits = (nB + P.spb - 1) / P.spb;
  P.A = A; P.a_bs = a_bs; P.B = B; P.b_bs = b_bs; P.slab = slab;
  // Wave groups per workgroup.  Two (eight waves, half the K splits, one workgroup per CU by LDS) where the
  // output is FEW tiles and the split-K partials dominate the kernel's HBM bytes: the attention weight
  // gradient (8 tiles: 1.26x -> 1.13x of the algorithmic bytes, time unchanged: 0.607 vs 0.601 of peak in the
  // step).  One for the i_embed weight gradient (16 tiles, 1.12x): there the eight waves on one barrier
  // measured slower (0.69-0.75 vs 0.73-0.79 of peak alone, 0.67 vs 0.72 in the step, step 9.45-9.51 vs
  // 9.37-9.39 ms).  RAU_WGRAD_GROUPS=1|2 forces either form for both (A/B).
  static const int genv = [] { const char* e = std::getenv("RAU_WGRAD_GROUPS"); return e ? std::atoi(e) : 0; }();
  const int groups = genv == 1 || genv == 2 ? genv : (P.tiles_a * P.tiles_b <= 8 ? 2 : 1);
  if (groups == 2 && splits >= 2) {
    const int s2 = (splits + 1) / 2;
    P.spb = (nB + s2 - 1) / s2;
    P.splits = (nB + P.spb - 1) / P.spb;
    const dim3 grid(8 * ((P.splits + 7) / 8) * P.tiles_a * P.tiles_b);
    hipLaunchKernelGGL(k_wgrad_dma<2>, grid, dim3(512), 0, st, P);
  } else {
    const dim3 grid(8 * ((P.splits + 7) / 8) * P.tiles_a * P.tiles_b);
    hipLaunchKernelGGL(k_wgrad_dma<1>, grid, dim3(256), 0, st, P);   // two per CU (one per CU measured 10.23 vs 9.70 ms/step)
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return splitk_reduce_acc(st, (size_t)ra * rb, P.splits, slab, (size_t)ra * rb, dW);
}

}  // namespace rau
