// exp_wide2.hip -- EXPERIMENT (built into tools/convbench only, not part of librau.so): the forward 1x1-conv
// GEMMs of 14 x 14 maps on 128-row x TWO-sample tiles fed by LDS-DMA.
//
// Outcome (round 4, LOG.md): bitwise equal to the library's tilings at the first run; stand-alone 0.71 of the
// f32-MFMA peak at one workgroup per CU against 0.75 for conv_wide.hip's 64 x 784 tile (0.79-0.80 vs 0.81-0.83 at
// two); in the step 9.35-9.38 vs 9.27-9.31 ms, conv_embed_fwd 0.55-0.56 vs 0.58 of peak.  39 % less L2 -> LDS
// traffic and 46 % fewer LDS fragment reads per flop bought nothing: what the recurrence's kernels cost the
// forward convs is not a contention for those paths -- the in-step rate follows the stand-alone rate at one
// wave per SIMD.
//
//   C[b, m, s] = act( sum_k Wt[k, m] * X[b, k, s] + bias[m] )        m < M, s < 196
//
// (i_embed, reference SS:240-241: Wt = Wi^T, X = dropped-out feature map, act = tanh;
//  ifeatproj, SS:247: Wt = Wp^T, X = I, no activation.)
//
// Round 4: conv_wide.hip's kernel with the tile turned by a factor of two.  That kernel's tile is 64 rows x
// four whole samples (784 positions): per k it moves (64 + 784) floats L2 -> LDS for 2 x 64 x 784 flops and
// every wave reads all 49 position fragments + 1 weight fragment from LDS for 49 MFMAs.  What the round-4
// launch-skip runs say (profiles/r04_forward_gap.md): in the step the forward convs lose 0.48 ms to the
// recurrence's kernels running on the same CUs -- a contention for the CU's memory paths, not for its matrix
// pipe (those kernels do a tenth of the MFMA work).  This tile is 128 rows x TWO samples (392 positions):
//   * (128 + 392) floats per k for 2 x 128 x 392 flops: 39 % less L2 -> LDS traffic per flop;
//   * a wave owns 32 rows x all 392 positions = 2 x 25 accumulator blocks (24.5 real: the 25th block's upper
//     half is padding, never stored, 2 % of the MFMAs): every position fragment feeds TWO MFMAs -- 27 LDS
//     fragment reads per 50 MFMAs instead of 50 per 49;
//   * a stage is [8 k][400] of X + [8 k][128] of W = 16.9 KB, a ring of three 50.7 KB instead of 81.4.
// Tile counts stay whole rounds: (M / 128) x (samples / 2) = 1024 per 2-hop launch of i_embed.
// X rows go to LDS at a pitch of 400 floats = 16 mod 32 banks (the 16x16x4 fragment read -- lanes 0-15 row k,
// 16-31 row k + 1 -- is conflict-free); a k-row of the two samples is 98 pieces of 16 bytes = two DMA
// instructions (64 + 34 lanes); W rows are lane-linear (two per instruction) with the pieces of odd k-rows
// XOR-swizzled by 4 on the global side.  Pipeline, barrier placement (middle of a K-step) and the
// hand-counted fragment reads are conv_wide.hip's, in chunks of 5 fragments x 2 MFMAs.
// Compiled for at most 256 registers (accumulators in VGPRs: with a 512-register budget hipcc moves them to
// AGPRs and copies each block out behind the last K-step's MFMAs, with the MFMA latency exposed 100 times);
// ONE workgroup per CU in the step (half of every CU stays with the recurrence), which
// the launcher enforces by padding the LDS request as for conv_wide.hip.
// Exact f32: every output element is the same k-ordered fmaf chain as in the other tilings (bitwise equal).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "../rau_vqa_amd/csrc/common.h"
#include "../rau_vqa_amd/csrc/kernels.h"
#include "exp_wide2.h"

namespace rau {

namespace {

constexpr int CS = 196;               // positions per sample
constexpr int CNS = 2;                // samples per tile
constexpr int CNP = CNS * CS;         // 392 flattened positions
constexpr int CNB = 25;               // position blocks of 16 (the last one half padding)
constexpr int CPITCH = 16 * CNB;      // 400: LDS pitch of an X k-row, = 16 mod 32 banks
constexpr int CBM = 128;              // rows per tile (4 waves x 32)
constexpr int CBK = 8;                // K-step
constexpr int CXST = CBK * CPITCH;    // floats of the X part of a stage (3200)
constexpr int CWST = CBK * CBM;       // of the W part (1024)
constexpr int CSTAGE = CXST + CWST;   // 4224 floats = 16896 bytes
constexpr int CNST = 3;               // ring of stages: loads run two K-steps ahead
constexpr int CNSL = 5;               // LDS-DMA instructions per wave and K-step
constexpr int CCH = 5;                // fragments per chunk, chunks per half K-step
static_assert(CPITCH % 32 == 16 && CNB * 16 >= CNP, "pitch: whole blocks, 16 mod 32 banks");

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <class F, int... I>
__device__ __forceinline__ void cfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void cfor(F&& f) { cfor_impl(f, std::make_integer_sequence<int, N>{}); }

// LDS fragment read the compiler does not track: the caller counts lgkmcnt itself
template <int OFF>
__device__ __forceinline__ void lds_f32(float& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

struct Wide2Params {
  int M, K, nG, tiles_m;               // nG = pairs of samples
  const float* Wt; long w_rs;          // [K][M]
  const float* X; long x_bs;           // [b][K][S]
  float* C; long c_bs;                 // [b][M][S]
  const float* bias; int act;
};

__global__ __launch_bounds__(256, 2) void k_conv_wide2(const Wide2Params P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // CNST stages (+ residency padding)
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = l & 15, lq = l >> 4;
  const int nwg = P.tiles_m * P.nG;
  const int id = xcd_remap(blockIdx.x, nwg);        // the row tiles of one sample pair share an XCD's L2
  const int tm = id % P.tiles_m, g = id / P.tiles_m;
  const int m0 = tm * CBM, b0 = g * CNS;

  // ---- LDS-DMA slots.  Instructions 0..15 of a stage: X k-row kk = i >> 1, half h = i & 1 = pieces
  // [64 h, 64 h + 64) of the row's 98 (piece p: sample p / 49, 16-byte column p % 49; h = 1: lanes 0..33);
  // 16..19: W rows 2 q, 2 q + 1 (q = i - 16; lane l: row 2 q + (l >> 5), piece (l & 31) ^ 4 (l >> 5)).
  // Wave w issues X instructions w, w + 4, w + 8, w + 12 and W instruction 16 + w.
  uint32_t voff[CNSL];
  int loff[CNSL];
  const bool x_on = (w & 1) == 0 || l < CNP / 4 - 64;   // odd waves issue the 34-lane halves
#pragma unroll
  for (int n = 0; n < CNSL; ++n) {
    if (n < 4) {
      const int i = w + 4 * n, kk = i >> 1, h = i & 1;
      int p = 64 * h + l;
      if (p >= CNP / 4) p = 0;                         // masked lanes: any valid address
      const int jj = p / (CS / 4), off = p - jj * (CS / 4);
      voff[n] = (uint32_t)(((long)jj * P.x_bs + (long)kk * CS) * 4 + off * 16);
      loff[n] = kk * CPITCH + 256 * h;
    } else {
      const int kk = 2 * w + (l >> 5), piece = (l & 31) ^ ((l >> 5) << 2);
      voff[n] = (uint32_t)((long)kk * P.w_rs * 4 + piece * 16);
      loff[n] = CXST + 256 * w;
    }
  }
  const char* xk = reinterpret_cast<const char*>(P.X + (size_t)b0 * P.x_bs);   // advances 8 k-rows per step
  const char* wk = reinterpret_cast<const char*>(P.Wt + m0);
  const long xstep = (long)CBK * CS * 4, wstep = (long)CBK * P.w_rs * 4;
  auto issue = [&](auto n_tag, int stage, const char* xb, const char* wb) {
    constexpr int N = decltype(n_tag)::value;
    uint32_t vo = voff[N];
    asm volatile("" : "+v"(vo));   // keep the per-lane offset 32 bits wide
    float* dst = smem + stage * CSTAGE + loff[N];
    if constexpr (N < 4) {
      if (x_on) __builtin_amdgcn_global_load_lds((glb_ptr_t)(xb + vo), (lds_ptr_t)dst, 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + vo), (lds_ptr_t)dst, 16, 0, 0);
    }
  };
  auto issue_stage = [&](int stage, const char* xb, const char* wb) {
    cfor<CNSL>([&](auto n_tag) { issue(n_tag, stage, xb, wb); });
  };

  f32x4 acc[2][CNB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < CNB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = P.K / CBK;
  // prologue: K-steps 0 and 1 in flight
  issue_stage(0, xk, wk);
  issue_stage(1, xk + xstep, wk + wstep);

  // ---- fragment addresses (bytes, stage 0, half 0): lane (lr, lq) holds k = 4 q + lq
  //   X fragment j: position 16 j + lr                         -> + j * 64, + q * 4 * CPITCH * 4
  //   W fragment i: row 32 w + 16 i + lr, piece ((8 w + 4 i + (lr >> 2)) ^ 4 (lq & 1)) of k-row lq
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  const uint32_t xfrag_b = lds0 + (uint32_t)(lq * CPITCH + lr) * 4;
  uint32_t wfrag_b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = (8 * w + 4 * i + (lr >> 2)) ^ ((lq & 1) << 2);
    wfrag_b[i] = lds0 + (uint32_t)(CXST + lq * CBM + piece * 4 + (lr & 3)) * 4;
  }

  // Fragment pipeline (conv_wide.hip's, see there): the 25 position fragments of a half K-step (4 k) go
  // through two register sets of 5; while the 10 MFMAs of chunk c issue, the reads of chunk c + 1 are in
  // flight -- across half-steps and K-steps, because the synchronisation point sits in the MIDDLE of a K-step.
  float fa[CCH], fb[CCH], wcur[2], wnext[2];
  // one half-step on (xa, q); its last chunk prefetches chunk 0 of (xa_n, q_n) and the weight fragments of
  // (wa_n, q_n).  A0: the half-step's chunk 0 sits in fa (else fb).  DMA: issue the loads of K-step t + 2.
  // LAST: the tile's final half-step requests nothing in its last chunk (an asm read whose result nobody
  // uses is not harmless: its register gets reused and the LDS data lands in it whenever it arrives).
  auto half_step = [&](auto start_a, auto dma, auto last, auto q_tag, auto qn_tag, uint32_t xa, uint32_t xa_n,
                       uint32_t wa_n0, uint32_t wa_n1, int st2, const char* xn, const char* wn) {
    constexpr bool A0 = decltype(start_a)::value, DMA = decltype(dma)::value;
    constexpr bool LAST = decltype(last)::value;
    constexpr int Q = decltype(q_tag)::value, QN = decltype(qn_tag)::value;
    cfor<CCH>([&](auto c_tag) {
      constexpr int c = decltype(c_tag)::value;
      constexpr bool cur_a = A0 ? (c % 2 == 0) : (c % 2 == 1);
      float (&fn)[CCH] = cur_a ? fb : fa;          // set being filled (chunk c + 1)
      const float (&fc)[CCH] = cur_a ? fa : fb;    // set being consumed (chunk c)
      auto rd = [&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        if constexpr (LAST && c == CCH - 1) return;
        else if constexpr (c < CCH - 1) lds_f32<(Q * 4 * CPITCH + 16 * (CCH * (c + 1) + i)) * 4>(fn[i], xa);
        else lds_f32<(QN * 4 * CPITCH + 16 * i) * 4>(fn[i], xa_n);
      };
      auto mma = [&](int i) {   // X as the MFMA's A operand, W as its B operand: a lane's 4 registers are
                                // 4 CONSECUTIVE positions of one row m -> 16-byte stores
        acc[0][CCH * c + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fc[i], wcur[0], acc[0][CCH * c + i], 0, 0, 0);
        acc[1][CCH * c + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fc[i], wcur[1], acc[1][CCH * c + i], 0, 0, 0);
      };
      using std::integral_constant;
      // this chunk's fragments were requested under the first MFMAs of the previous chunk
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      rd(integral_constant<int, 0>{}); rd(integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(1);
      __builtin_amdgcn_sched_barrier(0);
      rd(integral_constant<int, 2>{}); rd(integral_constant<int, 3>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(2);
      __builtin_amdgcn_sched_barrier(0);
      rd(integral_constant<int, 4>{});
      if constexpr (c == CCH - 1 && !LAST) {
        lds_f32<QN * 4 * CBM * 4>(wnext[0], wa_n0);
        lds_f32<QN * 4 * CBM * 4>(wnext[1], wa_n1);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(3);
      if constexpr (DMA) issue(c_tag, st2, xn, wn);
      mma(4);
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (!LAST) { wcur[0] = wnext[0]; wcur[1] = wnext[1]; }
  };
  auto kstep = [&](auto dma, auto last, int stage) {
    int st1 = stage + 1, st2 = stage + 2;
    if (st1 >= CNST) st1 -= CNST;
    if (st2 >= CNST) st2 -= CNST;
    const uint32_t so = stage * (CSTAGE * 4), so1 = st1 * (CSTAGE * 4);
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // first half (k 0..3); its last chunk reads ahead into the SECOND half of the same stage
    half_step(std::true_type{}, std::false_type{}, std::false_type{}, I0{}, I1{}, xfrag_b + so, xfrag_b + so,
              wfrag_b[0] + so, wfrag_b[1] + so, 0, nullptr, nullptr);
    // every wave's loads of K-step t + 1 have landed (issued one K-step ago), and every wave is done with
    // K-step t - 1, whose stage (t + 2) % 3 is free again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // second half (k 4..7): reads ahead into the first half of stage t + 1, issues the loads of K-step t + 2
    half_step(std::false_type{}, dma, last, I1{}, I0{}, xfrag_b + so, xfrag_b + so1, wfrag_b[0] + so1,
              wfrag_b[1] + so1, st2, xk + 2 * xstep, wk + 2 * wstep);
    xk += xstep;
    wk += wstep;
  };

  asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // K-step 0 has landed (this wave's share)
  __builtin_amdgcn_s_barrier();
  cfor<CCH>([&](auto i_tag) {
    constexpr int i = decltype(i_tag)::value;
    lds_f32<16 * i * 4>(fa[i], xfrag_b);
  });
  lds_f32<0>(wcur[0], wfrag_b[0]);
  lds_f32<0>(wcur[1], wfrag_b[1]);
  int stage = 0;
  for (int t = 0; t + 2 < nk; ++t) {
    kstep(std::true_type{}, std::false_type{}, stage);
    stage = stage + 1 == CNST ? 0 : stage + 1;
  }
  kstep(std::false_type{}, std::false_type{}, stage);   // K-step nk-2: nothing left to load
  stage = stage + 1 == CNST ? 0 : stage + 1;
  kstep(std::false_type{}, std::true_type{}, stage);    // K-step nk-1: nothing left to read ahead either

  // ---- epilogue, straight from registers: block (i, j) register r = C[m0 + 32 w + 16 i + lr][16 j + 4 lq + r]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 32 * w + 16 * i + lr;
    const float bv = P.bias ? P.bias[m] : 0.f;
#pragma unroll
    for (int j = 0; j < CNB; ++j) {
      const int p = 16 * j + 4 * lq;
      if (p >= CNP) continue;            // the 25th block's upper half (196 % 4 == 0: a quad is all valid or all pad)
      const int jj = p / CS, s = p - jj * CS;
      float4 v = make_float4(acc[i][j][0] + bv, acc[i][j][1] + bv, acc[i][j][2] + bv, acc[i][j][3] + bv);
      if (P.act) { v.x = tanh_fast(v.x); v.y = tanh_fast(v.y); v.z = tanh_fast(v.z); v.w = tanh_fast(v.w); }
      *reinterpret_cast<float4*>(P.C + (size_t)(b0 + jj) * P.c_bs + (size_t)m * CS + s) = v;
    }
  }
}

}  // namespace

// Shapes it takes: 14 x 14 maps, rows a multiple of 128, reduction a multiple of 8 and at least two K-steps,
// 16-byte aligned rows.  Odd sample counts are the caller's to split (nB must be even).
bool conv_wide2_ok(int M, int K, int S, long w_rs) {
  return S == CS && M % CBM == 0 && K % CBK == 0 && K >= 2 * CBK && w_rs % 4 == 0;
}

// C = act(acc + bias[m]).  per_cu: 1 = pad the LDS request so that only one workgroup fits a CU, 2 = up to two.
hipError_t conv_wide2(hipStream_t st, int nB, int M, int K, int S, const float* Wt, long w_rs, const float* X,
                      long x_bs, float* C, long c_bs, const float* bias, int act, int per_cu) {
  if (!conv_wide2_ok(M, K, S, w_rs) || nB % CNS != 0 || x_bs % 4 != 0 || c_bs % 4 != 0) return hipErrorInvalidValue;
  if (nB == 0) return hipSuccess;
  Wide2Params P{};
  P.M = M; P.K = K; P.nG = nB / CNS; P.tiles_m = M / CBM;
  P.Wt = Wt; P.w_rs = w_rs;
  P.X = X; P.x_bs = x_bs;
  P.C = C; P.c_bs = c_bs;
  P.bias = bias; P.act = act;
  constexpr int kRing = CNST * CSTAGE * 4;   // 50688 bytes
  constexpr int kOne = 82 * 1024;            // more than half of a CU's 160 KB: a second workgroup does not fit
  static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wide2),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kOne);
  if (attr_err != hipSuccess) return attr_err;
  hipLaunchKernelGGL(k_conv_wide2, dim3(P.tiles_m * P.nG), dim3(256), per_cu == 2 ? kRing : kOne, st, P);
  return hipGetLastError();
}

}  // namespace rau
