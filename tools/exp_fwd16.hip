// EXPERIMENT, not part of librau.so: measured faster alone (170 vs 240 us per hop at D = 2048) and
// slower in the step (9.39 vs 9.26 ms: 98 KB of LDS and 368 registers per wave leave the recurrence's
// kernels too little of each CU while the forward phase is bound by them; a two-slot ring is
// latency-bound at 326 us).  Kept buildable through tools/convbench (`convbench 256 fwd16`) so the
// numbers in DESIGN.md section 5 can be reproduced.  What bounds it alone: LDS-DMA moves ~6.4 TB/s
// chip-wide however contiguous the source (1.09 GB per hop here).
//
// i_embed forward in bf16-operand mode (BASELINE.json configs[2]) on 14 x 14 maps:
//   I[b][m][s] = tanh(bi[m] + sum_d Wi[m][d] X16[b][d][s])
// (train_vqa_RAU_SS.lua:240, i_embed = SpatialConvolution(D -> M, 1x1) + Tanh), X16 = the hop's dropout
// copy of the feature map stored as bf16 [b][D][196], f32 accumulation (v_mfma_f32_16x16x32_bf16).
//
// Round 2 ran this on the 128x128x32 register-staged tile of gemm_core.h: 459 us per 2-hop launch =
// 1.35 TB/s algorithmic, matrix pipe 23 % busy, 0.24 LDS bank conflicts per access (the operand is
// transposed in registers while it is staged).  Here nothing is transposed on the way in:
//  * the reduction runs over channels, and 32 channels of one sample ARE one contiguous block of the
//    operand (32 rows x 392 bytes): a K-step's activations go HBM/L2 -> LDS by DMA in whole lines,
//    exactly as they lie in memory (wgrad16.hip's lesson: pieces of rows cost 3x the bytes);
//  * the MFMA wants, per lane, 8 consecutive CHANNELS of one position -- a column of that image:
//    `ds_read_b64_tr_b16` delivers 4 rows x 16 columns transposed, two of them make a fragment
//    (verified on the hardware against the guide's lane map before anything else was built);
//  * weights are kept k-blocked, WD [D / 32][M][32 d] bf16 (weights_kblocked, once per step), so a
//    K-step of a row tile is one contiguous 8 KB block and its fragments plain ds_read_b128 of a
//    [row][4 pieces] image, swizzled on the global side as in skinny_dma.hip;
//  * tile = TWO samples x 128 rows of M, 4 waves = (sample, 64-row half): 13 position blocks (196 +
//    12 never-stored columns that read on into the next rows) x 4 row blocks = 52 accumulator
//    blocks per wave; positions are the accumulator's rows, so a lane's four registers are four
//    consecutive positions of one m: 16-byte stores.  512 tiles per hop = two whole rounds of one
//    workgroup per CU (208 accumulator + ~150 other registers per wave, 98 KB of LDS);
//  * ring of three 32.5 KB stages, two in flight behind the one being multiplied, one barrier per
//    stage, fragments of stage s+1 read under stage s's MFMAs.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "../rau_vqa_amd/csrc/common.h"
#include "exp_fwd16.h"

namespace rau {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int FS = 196;                  // positions per sample
constexpr int FROW = FS * 2;             // bytes per channel row
constexpr int FNS = 2;                   // samples per tile
constexpr int FTM = 128;                 // rows of M per tile
constexpr int FXS = 32 * FROW;           // bytes of one sample's K-step (12544)
constexpr int FXA = FNS * FXS;           // 25088
constexpr int FWB = FTM * 64;            // bytes of the weights' K-step (8192)
constexpr int FSTAGE = FXA + FWB;        // 33280
constexpr int FNST = 3;
constexpr int FLDS = FNST * FSTAGE + 64; // + slack for the never-stored columns of the last row
constexpr int FNPB = 13;                 // position blocks per sample
constexpr int FNMB = 4;                  // row blocks per wave
constexpr int FSLOTS = 9;                // DMA instructions per wave and stage

struct Fwd16Params {
  int nB, D, M, tiles_m;
  const uint16_t* X;   // [b][D][196]
  const uint16_t* WD;  // [D / 32][M][32]
  const float* bias;
  float* C;            // [b][M][196]
};

template <int OFF>
__device__ __forceinline__ void lds_read128(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr(u32x2& dst, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ int swz(int row) { return (-((row & 15) >> 2)) & 3; }

template <class F, int... I>
__device__ __forceinline__ void gfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void gfor(F&& f) { gfor_impl(f, std::make_integer_sequence<int, N>{}); }

__global__ __launch_bounds__(256, 1) void k_fwd16(const Fwd16Params P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wj = w & 1, wh = w >> 1;   // sample of the pair / 64-row half of the M tile
  // the row tiles of a sample pair are consecutive work items; an XCD takes a contiguous run
  const int total_wg = (P.nB / FNS) * P.tiles_m;
  int g = blockIdx.x;
  if ((total_wg & 7) == 0) g = (g & 7) * (total_wg >> 3) + (g >> 3);
  const int tm = g % P.tiles_m, pair = g / P.tiles_m;
  const int b0 = pair * FNS, m0 = tm * FTM;
  const int total = P.D / 32;

  // ---- DMA.  A stage = 1568 pieces of 16 bytes of X (sample jj, then the 784 pieces of its 32
  // rows, contiguous in memory and in LDS) + 512 of WD.  Wave-instructions: X 0..23 cover pieces
  // [64 i, +64), X 24 covers [1504, 1568) (its first 32 pieces repeat instruction 23's), W 25..32.
  // Wave w issues instructions w, w + 4, ..: nine slots each, the surplus three repeat W 32.
  uint32_t voff[FSLOTS];
  int loff[FSLOTS];
  bool isw[FSLOTS];
  const size_t xbs = (size_t)P.D * FROW;   // bytes per sample
#pragma unroll
  for (int n = 0; n < FSLOTS; ++n) {
    int i = w + 4 * n;
    if (i > 32) i = 32;
    if (i < 25) {
      const int p0 = i < 24 ? 64 * i : 1504;
      const int p = p0 + l;
      const int jj = p / 784, r = p - jj * 784;
      voff[n] = (uint32_t)((size_t)jj * xbs + (size_t)r * 16);
      loff[n] = p0 * 16;
      isw[n] = false;
    } else {
      const int q = (i - 25) * 64 + l, row = q >> 2;
      voff[n] = (uint32_t)(row * 64 + (((q & 3) ^ swz(row)) << 4));
      loff[n] = FXA + (i - 25) * 1024;
      isw[n] = true;
    }
  }
  const char* x0 = reinterpret_cast<const char*>(P.X) + (size_t)b0 * xbs;
  const char* w0 = reinterpret_cast<const char*>(P.WD) + (size_t)m0 * 64;
  const uint32_t wcs = (uint32_t)P.M * 64;   // bytes per K-step of WD
  uint32_t xnext = 0, wnext = 0;
  int nissued = 0;
  auto issue_slot = [&](auto n_tag) {
    constexpr int n = decltype(n_tag)::value;
    uint32_t vo = voff[n] + (isw[n] ? wnext : xnext);
    asm volatile("" : "+v"(vo));
    __builtin_amdgcn_global_load_lds((glb_ptr_t)((isw[n] ? w0 : x0) + vo),
                                     (lds_ptr_t)(smem + (nissued % FNST) * FSTAGE + loff[n]), 16, 0, 0);
  };
  auto issued = [&]() { xnext += 32 * FROW; wnext += wcs; ++nissued; };

  // ---- fragments.  MFMA lane (fr = l & 15, kk = l >> 4) holds channels 8 kk .. 8 kk + 7.
  // Positions (A operand): transposing read, lane 4 q + p of a 16-lane group supplies the address of
  // row q, columns 4 p .. 4 p + 3 of the block; two reads (rows 8 kk + 0..3, + 4..7) per fragment.
  // Weights (B operand): one ds_read_b128 of the swizzled [row][4 pieces] image.
  const int fr = l & 15, kk = l >> 4;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  const uint32_t fx = lds0 + (uint32_t)(wj * FXS + (8 * kk + (fr >> 2)) * FROW + (fr & 3) * 8);
  const uint32_t fw = lds0 + (uint32_t)(FXA + (wh * 64 + fr) * 64 + ((kk ^ swz(fr)) << 4));

  f32x4 acc[FNPB][FNMB];
#pragma unroll
  for (int i = 0; i < FNPB; ++i)
#pragma unroll
    for (int j = 0; j < FNMB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x2 xa[2][FNPB][2];
  u32x4 wf[2][FNMB];

  auto read_frags = [&](auto set_tag, int slot) {
    constexpr int set = decltype(set_tag)::value;
    const uint32_t ax = fx + slot * FSTAGE, aw = fw + slot * FSTAGE;
    gfor<FNPB>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      lds_read_tr<i * 32>(xa[set][i][0], ax);
      lds_read_tr<i * 32 + 4 * FROW>(xa[set][i][1], ax);
    });
    gfor<FNMB>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      lds_read128<j * 1024>(wf[set][j], aw);
    });
  };

  // ---- prologue: two stages in flight, stage 0's fragments in set 0 (total >= 3)
  gfor<FSLOTS>([&](auto n_) { issue_slot(n_); });
  issued();
  if (FNST == 3) {
    gfor<FSLOTS>([&](auto n_) { issue_slot(n_); });
    issued();
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  gfor<FSLOTS>([&](auto n_) { issue_slot(n_); });
  issued();
  read_frags(std::integral_constant<int, 0>{}, 0);

  auto body = [&](auto set_tag, int s) {
    constexpr int set = decltype(set_tag)::value;
    const bool more = s + 1 < total;
    if (more) {   // stage s+1 has landed (this wave's pieces); stage s+2 may still be in flight
      if (FNST == 3 && s + 2 < total) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // fragments of stage s are in registers (every consumer below depends on these waits)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xa[set][0][0]), "+v"(xa[set][0][1]), "+v"(xa[set][1][0]), "+v"(xa[set][1][1]),
                   "+v"(xa[set][2][0]), "+v"(xa[set][2][1]), "+v"(xa[set][3][0]), "+v"(xa[set][3][1]),
                   "+v"(xa[set][4][0]), "+v"(xa[set][4][1]), "+v"(xa[set][5][0]), "+v"(xa[set][5][1]),
                   "+v"(xa[set][6][0]), "+v"(xa[set][6][1])
                 :
                 : "memory");
    asm volatile(""
                 : "+v"(xa[set][7][0]), "+v"(xa[set][7][1]), "+v"(xa[set][8][0]), "+v"(xa[set][8][1]),
                   "+v"(xa[set][9][0]), "+v"(xa[set][9][1]), "+v"(xa[set][10][0]), "+v"(xa[set][10][1]),
                   "+v"(xa[set][11][0]), "+v"(xa[set][11][1]), "+v"(xa[set][12][0]), "+v"(xa[set][12][1]),
                   "+v"(wf[set][0]), "+v"(wf[set][1]), "+v"(wf[set][2]), "+v"(wf[set][3])
                 :
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();   // stage s+1 visible; everyone holds stage s in registers
    __builtin_amdgcn_sched_barrier(0);
    if (more) read_frags(std::integral_constant<int, set ^ 1>{}, (s + 1) % FNST);
    __builtin_amdgcn_sched_barrier(0);
    const bool feed = s + FNST < total;   // stage s+3 goes into stage s's slot, one instruction per MFMA row
    gfor<FNPB>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      const bf16x8 a = __builtin_bit_cast(bf16x8, u32x4{xa[set][i][0][0], xa[set][i][0][1],
                                                        xa[set][i][1][0], xa[set][i][1][1]});
#pragma unroll
      for (int j = 0; j < FNMB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, wf[set][j]),
                                                            acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i < FSLOTS) {
        if (feed) issue_slot(i_);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    if (feed) issued();
  };
  int s = 0;
#pragma unroll 1
  for (; s + 1 < total; s += 2) {
    body(std::integral_constant<int, 0>{}, s);
    body(std::integral_constant<int, 1>{}, s + 1);
  }
  if (s < total) body(std::integral_constant<int, 0>{}, s);

  // ---- epilogue: block (i, j) register r = (position 16 i + 4 kk + r, row m 16 j + fr)
  float* Cb = P.C + (size_t)(b0 + wj) * P.M * FS;
#pragma unroll
  for (int j = 0; j < FNMB; ++j) {
    const int m = m0 + wh * 64 + 16 * j + fr;
    const float bv = P.bias ? P.bias[m] : 0.f;
#pragma unroll
    for (int i = 0; i < FNPB; ++i) {
      const int sp = 16 * i + 4 * kk;
      if (sp < FS) {   // 196 % 4 == 0: a lane's four positions are all valid or all beyond the map
        f32x4 v = acc[i][j];
        v[0] = tanh_fast(v[0] + bv); v[1] = tanh_fast(v[1] + bv);
        v[2] = tanh_fast(v[2] + bv); v[3] = tanh_fast(v[3] + bv);
        *reinterpret_cast<f32x4*>(Cb + (size_t)m * FS + sp) = v;
      }
    }
  }
}

// WD [D / 32][M][32 d] = bf16(W[m][d])
__global__ void k_weights_kblocked(int M, int D, const float* __restrict__ W, uint16_t* __restrict__ WD) {
  const size_t n4 = (size_t)M * D / 4;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
    const size_t e = q * 4;
    const int m = (int)(e / D), d = (int)(e % D);
    const float4 x = *reinterpret_cast<const float4*>(W + e);
    typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
    b16x4 o;
    o[0] = (__bf16)x.x; o[1] = (__bf16)x.y; o[2] = (__bf16)x.z; o[3] = (__bf16)x.w;
    *reinterpret_cast<uint2*>(WD + ((size_t)(d >> 5) * M + m) * 32 + (d & 31)) = __builtin_bit_cast(uint2, o);
  }
}

}  // namespace

hipError_t weights_kblocked(hipStream_t st, int M, int D, const float* W, void* WD) {
  if (D % 32 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_weights_kblocked, dim3(1024), dim3(256), 0, st, M, D, W, reinterpret_cast<uint16_t*>(WD));
  return hipGetLastError();
}

bool fwd16_ok(int nB, int D, int S, int M) {
  return S == FS && nB >= FNS && nB % FNS == 0 && M % FTM == 0 && D % 32 == 0 && D / 32 >= 3 &&
         (double)D * FROW * FNS < 4294967296.0;
}

// I[b][m][s] = tanh(bi[m] + sum_d WD[d / 32][m][d % 32] X16[b][d][s])
hipError_t fwd16(hipStream_t st, int nB, int D, int S, int M, const void* X16, const void* WD,
                 const float* bi, float* I) {
  if (!fwd16_ok(nB, D, S, M)) return hipErrorInvalidValue;
  Fwd16Params P{};
  P.nB = nB; P.D = D; P.M = M; P.tiles_m = M / FTM;
  P.X = static_cast<const uint16_t*>(X16);
  P.WD = static_cast<const uint16_t*>(WD);
  P.bias = bi; P.C = I;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd16),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, FLDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(k_fwd16, dim3((nB / FNS) * P.tiles_m), dim3(256), FLDS, st, P);
  return hipGetLastError();
}

}  // namespace rau
