// conv_wide.hip -- the position-flattened 1x1-conv GEMMs of 14 x 14 maps on WIDE tiles fed by LDS-DMA.
//
//   C[b, m, s] = epi( sum_k Wt[k, m] * X[b, k, s] )        m < M, s < 196
//
// (i_embed, reference SS:240: Wt = Wi^T, X = dropped-out feature map, epi = tanh(. + bi);
//  ifeatproj, SS:247: Wt = Wp^T, X = I, epi = . + bp;  attbycontent's input gradient + the gradient
//  through i_embed's tanh, SS:565-579: Wt = Wp, X = dS, epi = (. + dj[b,m] a[b,s]) (1 - I^2).)
//
// Why a third tiling (round 3) next to gemm_core.h's 128 x 128 flattened-column tiles and
// gemm_sample.hip's one-sample tiles: measured on the round-2 kernels, these products lose a third of
// the f32 matrix pipe to (a) a ragged last round (3136 tiles on 512 resident workgroups), (b) an
// un-overlapped prologue / epilogue per 27-us tile and (c) the register-staged operand pipeline
// (global -> VGPR -> ds_write -> barrier each K-step, both workgroups of a CU in lock-step).  Here
//   * a tile is 64 rows x FOUR WHOLE SAMPLES = 784 = 49 * 16 flattened positions: no padded MFMA
//     work at all (one sample is 12.25 blocks of 16), 1024 tiles per 2-hop launch of i_embed =
//     whole rounds at one or two workgroups per CU, and a tile lasts ~3x longer than a 128 x 128 one,
//     so its fixed costs weigh a third;
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no
//     ds_write pass, loads of K-step t+2 in flight across the barrier of K-step t (counted vmcnt +
//     raw s_barrier, cdna_hip_programming.md section 5 "Pipelining across barriers").  The LDS image
//     of the X part is [8 k][784] floats, pitch 784 = 16 mod 32 banks: a 16x16x4 fragment read (lanes
//     0-15 row k, 16-31 row k+1) is conflict-free, and the image is lane-linear for the DMA
//     (49 pieces of 16 bytes per sample row, 196 per k-row);
//   * each of the 4 waves owns 16 rows x all 784 positions: 49 accumulator blocks of
//     v_mfma_f32_16x16x4_f32 (196 registers), ONE W fragment per 49 MFMAs; positions feed the MFMA's
//     A operand, so a lane's four accumulator registers are four consecutive positions (16-byte
//     stores straight from registers: the epilogue needs no LDS and no barrier).
// Exact f32: every output element is the same k-ordered fmaf chain as in the other two tilings
// (bitwise equal results).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace rau {

namespace {

constexpr int WS = 196;               // positions per sample (14 x 14)
constexpr int WNS = 4;                // samples per tile
constexpr int WNP = WNS * WS;         // 784 flattened positions per tile
constexpr int WNB = WNP / 16;         // 49 position blocks
constexpr int WBM = 64;               // rows per tile (4 waves x 16)
constexpr int WBK = 8;                // K-step
constexpr int WXST = WBK * WNP;       // floats of the X part of a stage (6272)
constexpr int WWST = WBK * WBM;       // floats of the W part (512)
constexpr int WSTAGE = WXST + WWST;   // 6784 floats = 27136 bytes
constexpr int WNST = 3;               // ring of stages: loads run two K-steps ahead
constexpr int WNSLOT = 7;             // LDS-DMA instructions per wave and K-step
static_assert(WNP % 16 == 0 && (WNP % 32) == 16, "tile row = whole 16-blocks, pitch 16 mod 32");

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// LDS fragment read the compiler does not track: the caller counts lgkmcnt itself
template <int OFF>
__device__ __forceinline__ void ds_read_f32(float& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

struct WideParams {
  int M, K, nG, tiles_m;               // nG = groups of 4 samples
  const float* Wt; long w_rs;          // [K][M]
  const float* X; long x_bs;           // [b][K][S]
  float* C; long c_bs;                 // [b][M][S]
  const float* bias; int act;          // EPI 0
  const float* dj; const float* av;    // EPI 2: + dj[b,m] * a[b,s]
  const float* Y; float* rs;           // EPI 2: * (1 - Y[b,m,s]^2); rs[b,m] = sum_s of the result
  int c16;                             // EPI 2: C stored as bf16
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void k_conv_wide(const WideParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // WNST stages (+ residency padding)
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = l & 15, lq = l >> 4;
  const int nwg = P.tiles_m * P.nG;
  const int id = xcd_remap(blockIdx.x, nwg);        // the row tiles of one sample group share an XCD's L2
  const int tm = id % P.tiles_m, g = id / P.tiles_m;
  const int m0 = tm * WBM, b0 = g * WNS;

  // ---- LDS-DMA slots.  A stage is 1696 pieces of 16 bytes: 1568 of X (k-row kk, sample jj, piece
  // off: 196 per k-row) then 128 of W (k-row kk, 16 per row).  Wave-instructions: X 0..23 cover
  // pieces [64 i, +64), X 24 covers [1504, 1568) (its first 32 pieces repeat instruction 23's: the
  // same bytes to the same place), W 25, 26.  Wave w issues instructions w, w + 4, ..: seven each
  // (wave 3's seventh repeats instruction 24).
  uint32_t voff[WNSLOT];
  int loff[WNSLOT];
  const bool slot6_w = (w == 1 || w == 2);
#pragma unroll
  for (int n = 0; n < WNSLOT; ++n) {
    int i = w + 4 * n;
    if (i == 27) i = 24;
    if (i < 25) {
      const int p0 = i < 24 ? 64 * i : WXST / 4 - 64;
      const int p = p0 + l;
      const int kk = p / (WNP / 4), r = p - kk * (WNP / 4);
      const int jj = r / (WS / 4), off = r - jj * (WS / 4);
      voff[n] = (uint32_t)(((long)jj * P.x_bs + (long)kk * WS) * 4 + off * 16);
      loff[n] = p0 * 4;
    } else {
      const int q0 = (i - 25) * 64;
      const int q = q0 + l;
      const int kk = q >> 4, c = q & 15;
      voff[n] = (uint32_t)((long)kk * P.w_rs * 4 + c * 16);
      loff[n] = WXST + q0 * 4;
    }
  }
  const char* xk = reinterpret_cast<const char*>(P.X + (size_t)b0 * P.x_bs);   // advances 8 k-rows per step
  const char* wk = reinterpret_cast<const char*>(P.Wt + m0);
  const long xstep = (long)WBK * WS * 4, wstep = (long)WBK * P.w_rs * 4;
  auto issue = [&](int n, int stage, const char* xb, const char* wb) {
    const char* base = (n == 6 && slot6_w) ? wb : xb;
    uint32_t vo = voff[n];
    asm volatile("" : "+v"(vo));   // keep the per-lane offset 32 bits wide (hipcc otherwise hoists 7 zero-extended pairs)
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + vo),
                                     (lds_ptr_t)(smem + stage * WSTAGE + loff[n]), 16, 0, 0);
  };

  f32x4 acc[WNB];
#pragma unroll
  for (int j = 0; j < WNB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = P.K / WBK;
  // prologue: K-steps 0 and 1 in flight
#pragma unroll
  for (int n = 0; n < WNSLOT; ++n) issue(n, 0, xk, wk);
#pragma unroll
  for (int n = 0; n < WNSLOT; ++n) issue(n, 1, xk + xstep, wk + wstep);
  const int xfrag = lq * WNP + lr;                 // fragment of (k = 4q + lq, position 16 j + lr)
  const int wfrag = WXST + lq * WBM + 16 * w + lr; // fragment of (k = 4q + lq, row 16 w + lr)

  // Fragment pipeline: the 49 position fragments of a half K-step (4 k) go through two register
  // sets of 7; while the 7 MFMAs of chunk c issue, the reads of chunk c + 1 are in flight -- across
  // half-steps and K-steps too, because the synchronisation point sits in the MIDDLE of a K-step:
  //   first half of K-step t  : stage t % 3, k 0..3
  //   s_waitcnt vmcnt(0) + s_barrier: every wave's loads of K-step t+1 have landed (issued one
  //     K-step ago), and every wave is done with K-step t-1, whose stage (t+2) % 3 is free again
  //   second half             : k 4..7, and the 7 LDS-DMA instructions of K-step t+2 go out
  //     between its chunks
  // so the first fragments of K-step t+1 are read (from a stage known to be complete) under the last
  // MFMAs of K-step t.  The fragment reads are inline asm with hand-counted lgkmcnt waits: left to
  // itself hipcc either sinks every read to just in front of its MFMA (one register pair for all 49
  // fragments, LDS latency exposed 25 times per half-step) or, with the order pinned, still waits
  // lgkmcnt(0) -- i.e. for the chunk it has just issued -- in front of every other MFMA group.
  float fa[7], fb[7], wcur, wnext;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  const uint32_t xfrag_b = lds0 + (uint32_t)xfrag * 4, wfrag_b = lds0 + (uint32_t)wfrag * 4;
  // one half-step on (xa, q); its last chunk prefetches chunk 0 of (xa_n, wa_n, q_n).
  // A0: the half-step's chunk 0 sits in fa (else fb).  DMA: issue the loads of K-step t+2.
  // LAST: the tile's final half-step requests nothing in its last chunk.  (An asm read whose result
  // nobody uses is not harmless: hipcc sees a dead value, hands its register to the next accumulator
  // write, and the LDS data lands in it whenever it arrives.)
  auto half_step = [&](auto start_a, auto dma, auto last, auto q_tag, auto qn_tag, uint32_t xa,
                       uint32_t xa_n, uint32_t wa_n, int st2, const char* xn, const char* wn) {
    constexpr bool A0 = decltype(start_a)::value, DMA = decltype(dma)::value;
    constexpr bool LAST = decltype(last)::value;
    constexpr int Q = decltype(q_tag)::value, QN = decltype(qn_tag)::value;
    static_for<7>([&](auto c_tag) {
      constexpr int c = decltype(c_tag)::value;
      constexpr bool cur_a = A0 ? (c % 2 == 0) : (c % 2 == 1);
      float (&fn)[7] = cur_a ? fb : fa;          // set being filled (chunk c + 1)
      const float (&fc)[7] = cur_a ? fa : fb;    // set being consumed (chunk c)
      // read i of the next chunk: chunk c + 1 of this half-step, or chunk 0 of the next one
      auto rd = [&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        if constexpr (LAST && c == 6) return;
        else if constexpr (c < 6) ds_read_f32<(Q * 4 * WNP + 16 * (7 * (c + 1) + i)) * 4>(fn[i], xa);
        else ds_read_f32<(QN * 4 * WNP + 16 * i) * 4>(fn[i], xa_n);
      };
      auto mma = [&](int i) {
        acc[7 * c + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fc[i], wcur, acc[7 * c + i], 0, 0, 0);
      };
      using std::integral_constant;
      // this chunk's fragments were requested under the first MFMAs of the previous chunk
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      rd(integral_constant<int, 0>{}); rd(integral_constant<int, 1>{}); rd(integral_constant<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(1);
      __builtin_amdgcn_sched_barrier(0);
      rd(integral_constant<int, 3>{}); rd(integral_constant<int, 4>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(2);
      __builtin_amdgcn_sched_barrier(0);
      rd(integral_constant<int, 5>{}); rd(integral_constant<int, 6>{});
      if constexpr (c == 6 && !LAST) ds_read_f32<QN * 4 * WBM * 4>(wnext, wa_n);
      __builtin_amdgcn_sched_barrier(0);
      mma(3);
      if constexpr (DMA) issue(c, st2, xn, wn);
      mma(4); mma(5); mma(6);
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (!LAST) wcur = wnext;
  };
  auto kstep = [&](auto dma, auto last, int stage) {
    int st1 = stage + 1, st2 = stage + 2;
    if (st1 >= WNST) st1 -= WNST;
    if (st2 >= WNST) st2 -= WNST;
    const uint32_t xa = xfrag_b + stage * (WSTAGE * 4), wa = wfrag_b + stage * (WSTAGE * 4);
    const uint32_t xa1 = xfrag_b + st1 * (WSTAGE * 4), wa1 = wfrag_b + st1 * (WSTAGE * 4);
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    half_step(std::true_type{}, std::false_type{}, std::false_type{}, I0{}, I1{}, xa, xa, wa, 0, nullptr,
              nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    half_step(std::false_type{}, dma, last, I1{}, I0{}, xa, xa1, wa1, st2, xk + 2 * xstep, wk + 2 * wstep);
    xk += xstep;
    wk += wstep;
  };

  asm volatile("s_waitcnt vmcnt(7)" ::: "memory");   // K-step 0 has landed (this wave's share)
  __builtin_amdgcn_s_barrier();
  static_for<7>([&](auto i_tag) {
    constexpr int i = decltype(i_tag)::value;
    ds_read_f32<16 * i * 4>(fa[i], xfrag_b);
  });
  ds_read_f32<0>(wcur, wfrag_b);
  int stage = 0;
  for (int t = 0; t + 2 < nk; ++t) {
    kstep(std::true_type{}, std::false_type{}, stage);
    stage = stage + 1 == WNST ? 0 : stage + 1;
  }
  kstep(std::false_type{}, std::false_type{}, stage);   // K-step nk-2: nothing left to load
  stage = stage + 1 == WNST ? 0 : stage + 1;
  kstep(std::false_type{}, std::true_type{}, stage);    // K-step nk-1: nothing left to read ahead either

  // ---- epilogue, straight from registers: block j, register r = C[m][position 16 j + 4 lq + r]
  const int m = m0 + 16 * w + lr;
  if (EPI == 0) {
    const float bv = P.bias ? P.bias[m] : 0.f;
#pragma unroll
    for (int j = 0; j < WNB; ++j) {
      const int p = 16 * j + 4 * lq;
      const int jj = p / WS, s = p - jj * WS;
      float4 v = make_float4(acc[j][0] + bv, acc[j][1] + bv, acc[j][2] + bv, acc[j][3] + bv);
      if (P.act) { v.x = tanh_fast(v.x); v.y = tanh_fast(v.y); v.z = tanh_fast(v.z); v.w = tanh_fast(v.w); }
      *reinterpret_cast<float4*>(P.C + (size_t)(b0 + jj) * P.c_bs + (size_t)m * WS + s) = v;
    }
  } else {
    float djv[WNS], rsum[WNS];
#pragma unroll
    for (int jj = 0; jj < WNS; ++jj) {
      djv[jj] = P.dj[(size_t)(b0 + jj) * P.M + m];
      rsum[jj] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < WNB; ++j) {
      const int p = 16 * j + 4 * lq;
      const int jj = p / WS, s = p - jj * WS;
      const size_t e = (size_t)(b0 + jj) * P.c_bs + (size_t)m * WS + s;
      const float4 a4 = *reinterpret_cast<const float4*>(P.av + (size_t)(b0 + jj) * WS + s);
      const float4 y = *reinterpret_cast<const float4*>(P.Y + e);
      // a block may straddle two samples (jj depends on lq): select, no dynamic register index
      const float d = jj == 0 ? djv[0] : jj == 1 ? djv[1] : jj == 2 ? djv[2] : djv[3];
      float4 v = make_float4(acc[j][0] + d * a4.x, acc[j][1] + d * a4.y, acc[j][2] + d * a4.z,
                             acc[j][3] + d * a4.w);
      v.x *= 1.f - y.x * y.x; v.y *= 1.f - y.y * y.y;
      v.z *= 1.f - y.z * y.z; v.w *= 1.f - y.w * y.w;
      const float sv = (v.x + v.y) + (v.z + v.w);
#pragma unroll
      for (int q = 0; q < WNS; ++q) rsum[q] += jj == q ? sv : 0.f;
      if (P.c16) {
        typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
        b16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(P.C) + e) = __builtin_bit_cast(uint2, o);
      } else {
        *reinterpret_cast<float4*>(P.C + e) = v;
      }
    }
    // lanes lr, lr + 16, lr + 32, lr + 48 hold the four position quarters of a block's row
#pragma unroll
    for (int q = 0; q < WNS; ++q) {
      float v = rsum[q];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lq == 0) P.rs[(size_t)(b0 + q) * P.M + m] = v;
    }
  }
}

}  // namespace

// Shapes the wide tiling takes: 14 x 14 maps, rows a multiple of 64, reduction a multiple of 8 and
// at least two K-steps, 16-byte aligned rows.  Sample counts that are not a multiple of 4 are the
// caller's to split (conv_wide handles the multiple-of-4 part only).
bool conv_wide_ok(int M, int K, int S, long w_rs) {
  return S == WS && M % WBM == 0 && K % WBK == 0 && K >= 2 * WBK && w_rs % 4 == 0;
}

// epi 0: C = act(acc + bias[m]);  epi 2 (only in the tools' build, -DRAU_WIDE_DGRAD):
// C = (acc + dj[b,m] a[b,s]) (1 - Y[b,m,s]^2), rs[b,m] = sum_s C.
// per_cu: 1 = pad the LDS request so that only one workgroup fits a CU (leaves half of each CU's
// registers and LDS to the recurrence's kernels running beside it), 2 = two per CU.
hipError_t conv_wide(hipStream_t st, int epi, int nB, int M, int K, int S, const float* Wt, long w_rs,
                     const float* X, long x_bs, float* C, long c_bs, const float* bias, int act,
                     const float* dj, const float* av, const float* Y, float* rs, int c16, int per_cu) {
  if (!conv_wide_ok(M, K, S, w_rs) || nB % WNS != 0 || (epi != 0 && epi != 2)) return hipErrorInvalidValue;
  if (nB == 0) return hipSuccess;
  WideParams P{};
  P.M = M; P.K = K; P.nG = nB / WNS; P.tiles_m = M / WBM;
  P.Wt = Wt; P.w_rs = w_rs;
  P.X = X; P.x_bs = x_bs;
  P.C = C; P.c_bs = c_bs;
  P.bias = bias; P.act = act;
  P.dj = dj; P.av = av; P.Y = Y; P.rs = rs; P.c16 = c16;
  constexpr int kRing = WNST * WSTAGE * 4;            // 81408 bytes: two fit the 160 KB of a CU
  const int lds = per_cu == 1 ? kRing + 2048 : kRing;
  // dynamic-LDS attribute, once per process (magic static: contexts may be driven from several threads)
  static const hipError_t attr_err = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wide<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kRing + 2048);
#ifdef RAU_WIDE_DGRAD
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wide<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kRing + 2048);
#endif
    return e;
  }();
  if (attr_err != hipSuccess) return attr_err;
  const dim3 grid(P.tiles_m * P.nG), block(256);
  if (epi == 0) hipLaunchKernelGGL((k_conv_wide<0>), grid, block, lds, st, P);
#ifdef RAU_WIDE_DGRAD   // tools/convbench builds its own copy with the dgrad epilogue (a concluded negative
                        // in the step, DESIGN.md section 8: the library keeps the dgrad on per-sample tiles)
  else hipLaunchKernelGGL((k_conv_wide<2>), grid, block, lds, st, P);
#else
  else return hipErrorInvalidValue;
#endif
  return hipGetLastError();
}

}  // namespace rau
