// i_embed weight gradient in bf16-operand mode (BASELINE.json configs[2]; reference op: the
// SpatialConvolution(2048 -> 512, 1x1) backward of train_vqa_RAU_SS.lua:579 -> accGradParameters):
//   dW[m][d] += sum_{b,s} dZ16[b][m][s] * X16[b][d][s],   14 x 14 maps (S = 196), both operands bf16
// in HBM (rau_ctx: dZ is written as bf16 by the attention dgrad's epilogue, X16 = the hop's dropout
// copy of the feature map), f32 accumulation with v_mfma_f32_16x16x32_bf16.
//
// Round 2 ran this on the 128x128x32 register-staged tile of gemm_core.h: 442 us per hop = 0.93 TB/s
// algorithmic, matrix pipe 20 % busy, 0.25 LDS bank conflicts per access -- each K-step one loaded
// round trip, operands transposed while staging.  The contraction index s is the CONTIGUOUS one of
// both operands, so nothing needs transposing:
//  * tile = 256 rows of X x 128 rows of dZ per workgroup (4 waves x (128 x 64) = 32 accumulator
//    blocks each; a 256 x 256 tile with 64 blocks per wave spilled): 1.2 GB of L2 -> LDS traffic
//    per hop at D = 2048 instead of 1.6 GB with 128 x 128 tiles;
//  * operands go L2 -> LDS by DMA (global_load_lds_dwordx4) in 32-position chunks: a stage is
//    [256 + 128 rows][64 B] = 24 KB, ring of three, two stages in flight
//    behind the one being multiplied; fragments of stage s+1 are read while stage s's MFMAs run; the
//    image is lane-linear ([row][4 pieces of 16 B]) and the XOR swizzle that makes the ds_read_b128
//    fragment reads conflict-free is applied to WHICH piece a lane fetches;
//  * 196 = 6 x 32 + 4: the last four positions of every row of a sample arrive by three dword DMAs
//    per wave ([384 rows][8 B]) and take one v_mfma_f32_16x16x16_bf16 step whose lanes k >= 4 hold
//    zeros;
//  * a K split's tiles sit on one XCD (they stream the same samples), partials go to the slab
//    [split][m][d] and splitk_reduce_acc adds them in split order (deterministic).
// Measured (tools/convbench wgrad16, D = 2048, 256 samples): 210 us per hop = 1.22 TB/s algorithmic
// (round-2 tile on the same box: 272 us), LDS bank conflicts 0, HBM bytes = algorithmic; in the step
// conv_embed_wgrad 2.67 -> 1.85 ms, step 9.62 -> 9.25 ms.  What bounds it (ablations on the same box):
// without its MFMAs 210 us, without its DMA 127 us, loop skeleton + epilogue + reduce alone 88 us; the
// same DMA count over CONTIGUOUS memory 135 us.  A 32-position chunk of a 392-byte row is 64 bytes at
// an odd offset: every 16-byte piece pulls a mostly unused 128-byte line through the L2 -> CU path.
// Ring depth (3 vs 6 stages) and spreading the DMA issues between the MFMA rows changed nothing.
// The layout fix was tried and dropped: operands written k-blocked by their producers
// ([b][7][rows][32 s] for this product, [D / 32][(b, s)][32 d] for the forward) made every stage one
// contiguous block -- weight gradient 160-200 us, forward 240 us per hop (round 2: 230) -- but the
// dropout pass writing two layouts cost 1.4 ms per step more than it saved (step 9.68 vs 9.26 ms).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace rau {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int HS = 196;                 // positions per sample
constexpr int HROW = HS * 2;            // bytes per operand row
constexpr int HTX = 256;                // tile rows of X (4 waves = 2 x 128 rows of X by 2 x 64 of dZ)
constexpr int HTZ = 128;                // tile rows of dZ
constexpr int HNCH = 6;                 // 32-position chunks per sample (+ a 4-position tail)
constexpr int HPX = HTX * 64;           // bytes of X per stage
constexpr int HSTAGE = (HTX + HTZ) * 64;   // 24 KB
constexpr int HNST = 3;                 // ring slots
constexpr int HTAIL = (HTX + HTZ) * 8;  // bytes of one sample's tails (X rows, then dZ rows)
constexpr int HLDS = HNST * HSTAGE + 2 * HTAIL;   // 78 KB

struct Wgrad16Params {
  int ra, rb, nB, tiles_a, tiles_b, splits, spb;
  const uint16_t* A; long a_bs;   // dZ16 [b][ra][S]  (a_bs in elements)
  const uint16_t* B; long b_bs;   // X16  [b][rb][S]
  float* slab;                    // [split][ra][rb]
};

template <int OFF>
__device__ __forceinline__ void lds_read128(f32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read64(float2& dst, uint32_t addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ int swz(int row) { return (-((row & 15) >> 2)) & 3; }

template <class F, int... I>
__device__ __forceinline__ void gfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void gfor(F&& f) { gfor_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int NI = 8, NJ = 4;   // 16-row blocks of X / of dZ per wave

__global__ __launch_bounds__(256, 1) void k_wgrad16(const Wgrad16Params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = w & 1, wz = w >> 1;   // 128-row half of the X tile / 64-row half of the dZ tile
  const int tiles = P.tiles_a * P.tiles_b;
  const int L = blockIdx.x, r8 = L & 7, q8 = L >> 3;
  const int split = r8 + 8 * (q8 / tiles), tile = q8 % tiles;
  if (split >= P.splits) return;
  const int ta = tile % P.tiles_a, tb = tile / P.tiles_a;
  const int b_lo = split * P.spb;
  int b_hi = b_lo + P.spb;
  if (b_hi > P.nB) b_hi = P.nB;
  const int nsm = b_hi - b_lo;
  const int total = nsm * HNCH;
  if (nsm <= 0) return;

  // ---- DMA slots.  A stage's pieces: X instruction n (0..3) covers pieces 256 n + tid, dZ instruction
  // n (0..1) likewise; piece p = (row p >> 2, LDS column p & 3) fetches the row's 16-byte piece
  // (p & 3) ^ swz(row).  Byte offsets from the sample's first tile row.
  uint32_t voff[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int p = 256 * n + tid, row = p >> 2;
    voff[n] = (uint32_t)(row * HROW + (((p & 3) ^ swz(row)) << 4));
  }
  // tails (positions 192..195 = bytes 384..391 of a row) as [384 rows][2 dwords]: instructions 0, 1
  // cover the X rows, instruction 2 the dZ rows
  const uint32_t toff_x = (uint32_t)((tid >> 1) * HROW + 6 * 64 + (tid & 1) * 4);
  const char* x0 = reinterpret_cast<const char*>(P.B + (size_t)b_lo * P.b_bs + (size_t)tb * HTX * HS);
  const char* z0 = reinterpret_cast<const char*>(P.A + (size_t)b_lo * P.a_bs + (size_t)ta * HTZ * HS);
  const uint32_t xbs = (uint32_t)P.b_bs * 2, zbs = (uint32_t)P.a_bs * 2;   // bytes per sample

  // stage (sample sb, chunk ch) into ring slot ch % 3 (6 chunks per sample: the slot is static);
  // with it one of the three tail instructions of the same sample (chunks 3..5 repeat 0..2): seven
  // DMA instructions per wave and stage, whatever the stage
  auto issue = [&](int sb, auto ch_tag) {
    constexpr int ch = decltype(ch_tag)::value;
    // SGPR base (x0 / z0) + one 32-bit VGPR offset per piece: a split's samples span < 2 GB
    const uint32_t xs = (uint32_t)sb * xbs, zs = (uint32_t)sb * zbs;
    char* dst = smem + (ch % HNST) * HSTAGE + w * 1024;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      uint32_t vo = voff[n] + xs + ch * 64;
      asm volatile("" : "+v"(vo));
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(x0 + vo), (lds_ptr_t)(dst + n * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      uint32_t vo = voff[n] + zs + ch * 64;
      asm volatile("" : "+v"(vo));
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(z0 + vo), (lds_ptr_t)(dst + HPX + n * 4096), 16, 0, 0);
    }
    {
      constexpr int n = ch % 3;
      char* tdst = smem + HNST * HSTAGE + (sb & 1) * HTAIL + n * 1024 + w * 256;
      uint32_t vo = toff_x + (n == 1 ? 128 * HROW : 0) + (n < 2 ? xs : zs);
      asm volatile("" : "+v"(vo));
      __builtin_amdgcn_global_load_lds((glb_ptr_t)((n < 2 ? x0 : z0) + vo), (lds_ptr_t)tdst, 4, 0, 0);
    }
  };

  // ---- fragments.  Lane (fr = l & 15, kk = l >> 4) of a 16-row block holds positions 8 kk .. 8 kk + 7
  // of the stage's 32: one ds_read_b128.  X rows feed the MFMA's A operand (accumulator rows), dZ
  // rows its B operand, so a lane's four accumulator registers are four consecutive d of one m.
  const int fr = l & 15, kk = l >> 4;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  const uint32_t fx = lds0 + (uint32_t)((wx * 128 + fr) * 64 + ((kk ^ swz(fr)) << 4));
  const uint32_t fz = lds0 + (uint32_t)(HPX + (wz * 64 + fr) * 64 + ((kk ^ swz(fr)) << 4));
  const uint32_t tx = lds0 + (uint32_t)(HNST * HSTAGE + (wx * 128 + fr) * 8);
  const uint32_t tz = lds0 + (uint32_t)(HNST * HSTAGE + (HTX + wz * 64 + fr) * 8);

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 xf[2][NI], zf[2][NJ];
  float2 xt[NI], zt[NJ];

  auto read_frags = [&](auto set_tag, auto slot_tag) {
    constexpr int set = decltype(set_tag)::value;
    constexpr int so = decltype(slot_tag)::value * HSTAGE;
    const uint32_t ax = fx + so, az = fz + so;
    gfor<NI>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      lds_read128<i * 1024>(xf[set][i], ax);
    });
    gfor<NJ>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      lds_read128<j * 1024>(zf[set][j], az);
    });
  };
  auto read_tail = [&](int buf) {
    const uint32_t ax = tx + buf * HTAIL, az = tz + buf * HTAIL;
    gfor<NI>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      lds_read64<i * 128>(xt[i], ax);
    });
    gfor<NJ>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      lds_read64<j * 128>(zt[j], az);
    });
  };

  // ---- prologue
  int s = 0;   // stage being multiplied
  issue(0, std::integral_constant<int, 0>{});
  issue(0, std::integral_constant<int, 1>{});
  asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  issue(0, std::integral_constant<int, 2>{});
  read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

  auto body = [&](int sb, auto ch_tag) {
    constexpr int ch = decltype(ch_tag)::value;
    constexpr int set = ch & 1;
    const bool more = s + 1 < total;
    if (more) {   // stage s+1 has landed (this wave's pieces); stage s+2 may still be in flight
      if (s + 2 < total) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xf[set][0]), "+v"(xf[set][1]), "+v"(xf[set][2]), "+v"(xf[set][3]),
                   "+v"(xf[set][4]), "+v"(xf[set][5]), "+v"(xf[set][6]), "+v"(xf[set][7]),
                   "+v"(zf[set][0]), "+v"(zf[set][1]), "+v"(zf[set][2]), "+v"(zf[set][3])
                 :
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();   // stage s+1 visible; everyone holds stage s in registers
    __builtin_amdgcn_sched_barrier(0);
    if (s + 3 < total)              // stage s+3 = (sb + (ch >= 3), (ch + 3) % 6) into stage s's slot
      issue(sb + (ch >= 3 ? 1 : 0), std::integral_constant<int, (ch + 3) % HNCH>{});
    if (more) read_frags(std::integral_constant<int, set ^ 1>{}, std::integral_constant<int, (ch + 1) % HNST>{});
    if (ch == HNCH - 1) read_tail(sb & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf[set][i]),
                                                            __builtin_bit_cast(bf16x8, zf[set][j]),
                                                            acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (ch == HNCH - 1) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(xt[0]), "+v"(xt[1]), "+v"(xt[2]), "+v"(xt[3]), "+v"(xt[4]), "+v"(xt[5]),
                     "+v"(xt[6]), "+v"(xt[7]), "+v"(zt[0]), "+v"(zt[1]), "+v"(zt[2]), "+v"(zt[3])
                   :
                   : "memory");
      __builtin_amdgcn_sched_barrier(0);
      // 16x16x16: lane (fr, kk) holds positions 4 kk .. 4 kk + 3 of 16 -- only kk = 0 is real
      if (kk != 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) xt[i] = float2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) zt[j] = float2{0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, xt[i]),
                                                                __builtin_bit_cast(s16x4, zt[j]),
                                                                acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    ++s;
  };
#pragma unroll 1
  for (int sb = 0; sb < nsm; ++sb)
    gfor<HNCH>([&](auto ch_) { body(sb, ch_); });

  // ---- partial tile to the slab: accumulator block (i, j) register r = (d 16 i + 4 kk + r, m 16 j + fr)
  float* C = P.slab + (size_t)split * P.ra * P.rb;
  const int d0 = tb * HTX + wx * 128 + 4 * kk, m0 = ta * HTZ + wz * 64 + fr;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i)
      *reinterpret_cast<f32x4*>(C + (size_t)(m0 + 16 * j) * P.rb + d0 + 16 * i) = acc[i][j];
}

int wgrad16_splits(int nB, int ra, int rb) {
  const int tiles = (ra / HTZ) * (rb / HTX);
  int s = 256 / tiles;   // one workgroup per CU (160 VGPRs + 128 accumulator registers per wave)
  if (s > nB) s = nB;
  if (s < 1) s = 1;
  return s;
}

}  // namespace

bool wgrad16_ok(int ra, int rb, int S) {
  static const bool off = std::getenv("RAU_WGRAD16_OFF") != nullptr;
  return !off && S == HS && ra >= HTZ && rb >= HTX && ra % HTZ == 0 && rb % HTX == 0;
}

size_t wgrad16_slab_floats(int nB, int ra, int rb, int S) {
  if (!wgrad16_ok(ra, rb, S)) return 0;
  return (size_t)wgrad16_splits(nB, ra, rb) * ra * rb;
}

hipError_t wgrad16(hipStream_t st, int nB, int ra, int rb, int S, const void* A16, long a_bs,
                   const void* B16, long b_bs, float* dW, float* slab) {
  if (!wgrad16_ok(ra, rb, S) || nB < 1) return hipErrorInvalidValue;
  if ((double)nB * (double)(a_bs > b_bs ? a_bs : b_bs) * 2 >= 2147483648.0 * 8) return hipErrorInvalidValue;
  Wgrad16Params P{};
  P.ra = ra; P.rb = rb; P.nB = nB;
  P.tiles_a = ra / HTZ; P.tiles_b = rb / HTX;
  const int s = wgrad16_splits(nB, ra, rb);
  P.spb = (nB + s - 1) / s;
  P.splits = (nB + P.spb - 1) / P.spb;
  P.A = static_cast<const uint16_t*>(A16); P.a_bs = a_bs;
  P.B = static_cast<const uint16_t*>(B16); P.b_bs = b_bs;
  P.slab = slab;
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad16),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, HLDS);
  if (attr_err != hipSuccess) return attr_err;
  const int tiles = P.tiles_a * P.tiles_b;
  hipLaunchKernelGGL(k_wgrad16, dim3(8 * ((P.splits + 7) / 8) * tiles), dim3(256), HLDS, st, P);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return splitk_reduce_acc(st, (size_t)ra * rb, P.splits, slab, (size_t)ra * rb, dW);
}

}  // namespace rau
