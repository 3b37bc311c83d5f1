// dgrad_dma.hip -- attbycontent's input gradient on 14 x 14 maps, one sample per tile, operands by LDS-DMA.
//
//   dZ[b, m, s] = ( sum_k Wp[k, m] dS[b, k, s] + dj[b, m] a[b, s] ) (1 - I[b, m, s]^2)
//   rs[b, m]    = sum_s dZ[b, m, s]                                    (the i_embed bias gradient's rows)
//
// (reference SS:565-579: backward of SS:247-263 -- ifeatproj's gradInput, the attention-weighted sum's
//  gradient dj (x) a, and the gradient through i_embed's Tanh, SS:241.)
//
// Round 4.  The per-sample tile of gemm_sample.hip (128 rows x all 196 positions of ONE sample: whole
// rounds of workgroups, whole-row reads, one contiguous block of dZ per tile, I read with the addresses
// dZ is written with) kept, its operand pipeline replaced.  What the round-3 counters said about that
// kernel, the least efficient of the five conv classes (0.50-0.585 of the f32-MFMA peak alone, 0.47 in the
// step): matrix pipes 66 % busy, waves parked 31 % of their cycles -- a register-staged pipeline (global ->
// VGPR -> ds_write -> barrier) with ONE K-step of prefetch, four LDS round trips per K-step that hipcc
// leaves exposed (it waits lgkmcnt(0) in front of every MFMA group), and K = 256 = only 16 K-steps per
// tile.  Here, the structure that took the conv weight gradients to 0.89 busy (wgrad_dma.hip):
//   * a stage = 16 k-rows of dS[b] (16 x 196 floats, as they lie in memory) and of Wp (16 x 128), brought
//     in by global_load_lds_dwordx4 into a ring of DNST stages, six DMA instructions per wave and stage
//     issued ONE AT A TIME between MFMA groups; no staging registers, no ds_write pass;
//   * dS rows go to LDS at a pitch of 208 floats = 16 mod 32 banks (one DMA instruction per row, lanes
//     0..48 active), Wp rows lane-linear (two rows per instruction) with the 16-byte pieces of odd k-rows
//     XOR-swizzled by 4 on the GLOBAL side: both fragment reads (lanes 0-15 row k, 16-31 row k + 1) are
//     conflict-free ds_read_b32;
//   * fragment reads in inline asm, two register sets, the reads of k-group kb + 1 in flight under the 26
//     MFMAs of kb (counted lgkmcnt); the stage's barrier sits in front of its LAST k-group's MFMAs, when
//     that group's fragments are already in registers, so the first fragments of the next stage are read
//     under those MFMAs too and the loop never drains;
//   * the tile's 26 loads of I per lane (first row block) go out in front of the last stage.
// Two workgroups per CU (4 waves x (32 rows x 208 positions) = 26 accumulator blocks of
// v_mfma_f32_16x16x4_f32, 13 x 16 = 208 >= 196: 6 % padded work; pad columns are never stored): one
// tile's epilogue (100 KB of I in, 100 KB of dZ out) runs under the other's K loop.
// Exact f32: every output element is the same k-ordered fmaf chain and the same epilogue arithmetic as
// gemm_sample.hip's; results equal to f32 rounding (tools/convbench: 1 ulp in a tenth of the words -- the
// compiler contracts the epilogue's multiply-adds differently in the two kernels).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace rau {

namespace {

constexpr int DS = 196;               // positions per sample
constexpr int DSP = 208;              // LDS pitch of a dS row: 13 blocks of 16, = 16 mod 32 banks
constexpr int DNCB = 13;              // position blocks
constexpr int DBM = 128;              // tile rows
constexpr int DBK = 16;               // k-rows per stage
constexpr int DXST = DBK * DSP;       // floats of the dS part of a stage (3328)
constexpr int DWST = DBK * DBM;       // floats of the Wp part (2048)
constexpr int DSTAGE = DXST + DWST;   // 5376 floats = 21504 bytes
constexpr int DNSLOT = 6;             // DMA instructions per wave and stage
static_assert(DSP % 32 == 16 && DSP >= DS && DSP % 16 == 0, "pitch: whole blocks, 16 mod 32 banks");

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <class F, int... I>
__device__ __forceinline__ void dfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void dfor(F&& f) { dfor_impl(f, std::make_integer_sequence<int, N>{}); }

template <int OFF>
__device__ __forceinline__ void lds_f32(float& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

struct DgradParams {
  int M, K, nB, tiles_m;
  const float* Wt; long w_rs;          // Wp [K][M]
  const float* X; long x_bs;           // dS [b][K][S]
  float* C; long c_bs;                 // dZ [b][M][S]
  const float* dj; const float* av;    // [b][M], [b][S]
  const float* Y; float* rs;           // I [b][M][S]; rs [b][M]
};

template <int DNST>
__global__ __launch_bounds__(256, 2) void k_dgrad_dma(const DgradParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // DNST stages, then rowv[128], colv[208]
  float* rowv = smem + DNST * DSTAGE;
  float* colv = rowv + DBM;
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = l & 15, lq = l >> 4;
  const int nwg = P.tiles_m * P.nB;
  const int id = xcd_remap(blockIdx.x, nwg);        // the row tiles of one sample share an XCD's L2
  const int tm = id % P.tiles_m, b = id / P.tiles_m;
  const int m0 = tm * DBM;

  // (A start delay for the first round's second workgroup per CU, so that one tile's epilogue would run
  // under its partner's K loop, was measured and rejected: only the delay itself shows.  LOG.md, round 4.)

  // ---- DMA slots.  Instructions 0..15 of a stage: dS row kr = i (49 pieces, lanes 0..48, to pitch
  // 208); 16..23: Wp rows 2 q, 2 q + 1 (q = i - 16; lane l: row 2 q + (l >> 5), piece l & 31, fetched
  // from piece (l & 31) ^ 4 (l >> 5): the swizzle of odd rows).  Wave w issues i = w + 4 n, n = 0..5.
  uint32_t voff[DNSLOT];
  int loff[DNSLOT];
#pragma unroll
  for (int n = 0; n < DNSLOT; ++n) {
    const int i = w + 4 * n;
    if (n < 4) {                                   // i < 16
      voff[n] = (uint32_t)(i * DS * 4 + (l < 49 ? l : 0) * 16);
      loff[n] = i * DSP;
    } else {
      const int q = i - 16, kk = 2 * q + (l >> 5), p = (l & 31) ^ ((l >> 5) << 2);
      voff[n] = (uint32_t)((long)kk * P.w_rs * 4 + p * 16);
      loff[n] = DXST + q * 256;
    }
  }
  const char* xk = reinterpret_cast<const char*>(P.X + (size_t)b * P.x_bs);   // advances 16 k-rows per stage
  const char* wk = reinterpret_cast<const char*>(P.Wt + m0);
  const long xstep = (long)DBK * DS * 4, wstep = (long)DBK * P.w_rs * 4;
  // instruction N (compile-time: voff / loff stay in registers) of a stage of this wave, into ring slot `slot`
  auto issue = [&](auto n_tag, int slot, const char* xb, const char* wb) {
    constexpr int N = decltype(n_tag)::value;
    uint32_t vo = voff[N];
    asm volatile("" : "+v"(vo));   // keep the per-lane offset 32 bits wide
    float* dst = smem + slot * DSTAGE + loff[N];
    if constexpr (N < 4) {
      if (l < 49) __builtin_amdgcn_global_load_lds((glb_ptr_t)(xb + vo), (lds_ptr_t)dst, 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(wb + vo), (lds_ptr_t)dst, 16, 0, 0);
    }
  };

  // epilogue vectors and the pad columns [196, 208) of every dS row slot (the DMA never writes them;
  // they only feed accumulator columns that are never stored, but keep them finite)
  if (tid < DBM) rowv[tid] = P.dj[(size_t)b * P.M + m0 + tid];
  if (tid < DSP) colv[tid] = tid < DS ? P.av[(size_t)b * DS + tid] : 0.f;
  for (int e = tid; e < DNST * DBK * (DSP - DS); e += 256) {
    const int st = e / (DBK * (DSP - DS)), r = e % (DBK * (DSP - DS));
    smem[st * DSTAGE + (r / (DSP - DS)) * DSP + DS + r % (DSP - DS)] = 0.f;
  }

  using T = std::true_type;
  using F = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  using I5 = std::integral_constant<int, 5>;

  // ---- schedule (stage u lives in ring slot u % DNST; nk >= 3 stages):
  //   prologue : stages 0 .. DNST-2 issued whole, and instructions 0, 1 of stage DNST-1
  //   stage t  : k-group 0 [+ instr 2, 3 of stage t+DNST-1]   k-group 1 [+ instr 4, 5 of it]   k-group 2
  //              wait: this wave's share of stage t+1 landed; BARRIER(t): everyone's has, and everyone
  //              holds its k-group-3 fragments = is done reading stage t's slot
  //              first fragments of stage t+1 requested; k-group 3 [+ instr 0, 1 of stage t+DNST -> the
  //              slot just freed]
  // vmcnt at BARRIER(t): after stage t+1's last instruction (issued in k-group 1 of stage t-DNST+2) this
  // wave has issued the six instructions of stage t+2 when DNST = 3 and that stage exists, else none.
  const int nk = P.K / DBK;
#pragma unroll
  for (int u = 0; u < DNST - 1; ++u) {
    const char* xu = xk + u * xstep;
    const char* wu = wk + u * wstep;
    issue(I0{}, u, xu, wu); issue(I1{}, u, xu, wu); issue(I2{}, u, xu, wu);
    issue(I3{}, u, xu, wu); issue(I4{}, u, xu, wu); issue(I5{}, u, xu, wu);
  }
  issue(I0{}, DNST - 1, xk + (DNST - 1) * xstep, wk + (DNST - 1) * wstep);
  issue(I1{}, DNST - 1, xk + (DNST - 1) * xstep, wk + (DNST - 1) * wstep);

  f32x4 acc[2][DNCB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < DNCB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses (bytes, slot 0, k-group 0): lane (lr, lq) holds k = 4 kb + lq
  //   X fragment j: dS[k][16 j + lr]                        -> + j * 64, + kb * 4 * DSP * 4
  //   W fragment i: Wp[k][m0 + 32 w + 16 i + lr], piece ((8 w + 4 i + (lr >> 2)) ^ 4 (lq & 1))
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  const uint32_t xf = lds0 + (uint32_t)(lq * DSP + lr) * 4;
  uint32_t wf[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = (8 * w + 4 * i + (lr >> 2)) ^ ((lq & 1) << 2);
    wf[i] = lds0 + (uint32_t)(DXST + lq * DBM + piece * 4 + (lr & 3)) * 4;
  }

  float xa[DNCB], xb_[DNCB], wa[2], wb_[2];   // two fragment sets
  auto read_set = [&](auto set_tag, auto kb_tag, uint32_t so) {
    constexpr bool A = decltype(set_tag)::value;
    constexpr int KB = decltype(kb_tag)::value;
    float (&xs)[DNCB] = A ? xa : xb_;
    float (&ws)[2] = A ? wa : wb_;
    lds_f32<KB * 4 * DBM * 4>(ws[0], wf[0] + so);
    lds_f32<KB * 4 * DBM * 4>(ws[1], wf[1] + so);
    dfor<DNCB>([&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      lds_f32<(KB * 4 * DSP + 16 * j) * 4>(xs[j], xf + so);
    });
  };
  // the 26 MFMAs of one k-group; with `dma`, DMA instructions N0 and N0 + 1 go out between them
  auto mma_set = [&](auto set_tag, auto n0_tag, bool dma, int slot, const char* xb, const char* wb) {
    constexpr bool A = decltype(set_tag)::value;
    constexpr int N0 = decltype(n0_tag)::value;
    const float (&xs)[DNCB] = A ? xa : xb_;
    const float (&ws)[2] = A ? wa : wb_;
    dfor<DNCB>([&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      // X as the MFMA's A operand, W as its B operand: the accumulator block is C^T, i.e. a lane's 4
      // registers are 4 CONSECUTIVE positions of one row m -> 16-byte stores
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[j], ws[0], acc[0][j], 0, 0, 0);
      acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[j], ws[1], acc[1][j], 0, 0, 0);
      if constexpr (j == 3) { if (dma) issue(std::integral_constant<int, N0>{}, slot, xb, wb); }
      if constexpr (j == 9) { if (dma) issue(std::integral_constant<int, N0 + 1>{}, slot, xb, wb); }
    });
  };

  // EPI: this tile's block of I, 13 float4 per lane and row block
  float4 yv[DNCB];
  auto yload = [&](int i) {
#pragma unroll
    for (int j = 0; j < DNCB; ++j) {
      const int s = j * 16 + 4 * lq, m = m0 + w * 32 + i * 16 + lr;
      yv[j] = s < DS ? *reinterpret_cast<const float4*>(P.Y + (size_t)b * P.c_bs + (size_t)m * DS + s)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  // stage 0 has landed (this wave's share; behind it: stage 1 and two instructions of stage 2, or two
  // instructions of stage 1), then everyone's; the pad columns and rowv / colv are written
  if (DNST == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  __syncthreads();
  read_set(T{}, I0{}, 0);

  int slot = 0;
  // one stage; LAST: the tile's final stage (nothing left to load, to wait for or to read ahead) as its own
  // copy of the code, so that the I prefetch in front of it does not live in registers across the loop
  auto stage = [&](int t, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const uint32_t so = (uint32_t)slot * (DSTAGE * 4);
    const int slot_n = slot + 1 == DNST ? 0 : slot + 1;      // stage t + 1
    const int slot_p = slot == 0 ? DNST - 1 : slot - 1;      // stage t + DNST - 1 (= stage t - 1's slot)
    const bool pre_a = !LAST && t + DNST - 1 < nk;           // ... exists: its instructions 2..5 go out now
    const bool pre_b = !LAST && t + DNST < nk;               // stage t + DNST exists: instructions 0, 1 below
    const char* xa_n = xk + (long)(t + DNST - 1) * xstep;
    const char* wa_n = wk + (long)(t + DNST - 1) * wstep;
    // k-groups 0..2: the next group's fragments are requested before this group's MFMAs issue
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_set(F{}, I1{}, so);
    __builtin_amdgcn_sched_barrier(0);
    mma_set(T{}, I2{}, pre_a, slot_p, xa_n, wa_n);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_set(T{}, I2{}, so);
    __builtin_amdgcn_sched_barrier(0);
    mma_set(F{}, I4{}, pre_a, slot_p, xa_n, wa_n);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_set(F{}, I3{}, so);
    __builtin_amdgcn_sched_barrier(0);
    mma_set(T{}, I0{}, false, 0, nullptr, nullptr);
    __builtin_amdgcn_sched_barrier(0);
    // k-group 3's fragments are in registers: this wave reads stage t's slot no more
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST) {
      if (DNST == 3 && t + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_set(T{}, I0{}, (uint32_t)slot_n * (DSTAGE * 4));
      __builtin_amdgcn_sched_barrier(0);
    }
    mma_set(F{}, I0{}, pre_b, slot, xk + (long)(t + DNST) * xstep, wk + (long)(t + DNST) * wstep);
    __builtin_amdgcn_sched_barrier(0);
    slot = slot_n;
  };
#pragma unroll 1
  for (int t = 0; t + 1 < nk; ++t) stage(t, F{});
  yload(0);   // the tile's first row block of I: in flight under the last stage's 104 MFMAs
  __builtin_amdgcn_sched_barrier(0);
  stage(nk - 1, T{});

  // ---- epilogue: accumulator (i, j) register r = C[m0 + 32 w + 16 i + lr][16 j + 4 lq + r]
  float* Cb = P.C + (size_t)b * P.c_bs;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i > 0) {
      asm volatile("" ::: "memory");          // the previous block's stores are issued: its registers are free
      __builtin_amdgcn_sched_barrier(0);
      yload(i);
    }
    const int rl = w * 32 + i * 16 + lr;
    const int m = m0 + rl;
    const float rv = rowv[rl];
    float* crow = Cb + (size_t)m * DS;
    float rsum = 0.f;
#pragma unroll
    for (int j = 0; j < DNCB; ++j) {
      const int s = j * 16 + 4 * lq;
      if (s >= DS) continue;                  // S % 4 == 0: a float4 is all valid or all pad
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      const float4 c4 = *reinterpret_cast<const float4*>(colv + s);
      v.x += rv * c4.x; v.y += rv * c4.y; v.z += rv * c4.z; v.w += rv * c4.w;
      const float4 y = yv[j];                 // gradient through i_embed's tanh, and its row sums
      v.x *= 1.f - y.x * y.x; v.y *= 1.f - y.y * y.y;
      v.z *= 1.f - y.z * y.z; v.w *= 1.f - y.w * y.w;
      rsum += (v.x + v.y) + (v.z + v.w);
      *reinterpret_cast<float4*>(crow + s) = v;
    }
    // the four lanes lr, lr+16, lr+32, lr+48 hold the row's four position quarters
    rsum += __shfl_xor(rsum, 16, 64);
    rsum += __shfl_xor(rsum, 32, 64);
    if (lq == 0) P.rs[(size_t)b * P.M + m] = rsum;
  }
}

constexpr int dgrad_lds_bytes(int nst) { return (nst * DSTAGE + DBM + DSP) * 4; }

}  // namespace

// Shapes it takes: 14 x 14 maps, rows a multiple of 128, reduction a multiple of 16 (>= 3 stages), 16-byte
// aligned rows.
// RAU_DGRAD_DMA=0 keeps the register-staged per-sample kernel (A/B knob);
// RAU_DGRAD_DMA=2|3 selects the ring depth (default 3).
static int dgrad_dma_mode() {
  static const int v = [] { const char* e = std::getenv("RAU_DGRAD_DMA"); return e ? std::atoi(e) : 3; }();
  return v;
}
bool dgrad_dma_ok(int M, int K, int S, long w_rs) {
  return dgrad_dma_mode() != 0 && S == DS && M % DBM == 0 && K % DBK == 0 && K >= 3 * DBK && w_rs % 4 == 0;
}

hipError_t dgrad_dma(hipStream_t st, int nB, int M, int K, int S, const float* Wt, long w_rs,
                     const float* X, long x_bs, float* C, long c_bs, const float* dj, const float* av,
                     const float* Y, float* rs) {
  if (!dgrad_dma_ok(M, K, S, w_rs) || x_bs % 4 != 0 || c_bs % 4 != 0) return hipErrorInvalidValue;
  if (nB == 0) return hipSuccess;
  DgradParams P{};
  P.M = M; P.K = K; P.nB = nB; P.tiles_m = M / DBM;
  P.Wt = Wt; P.w_rs = w_rs;
  P.X = X; P.x_bs = x_bs;
  P.C = C; P.c_bs = c_bs;
  P.dj = dj; P.av = av; P.Y = Y; P.rs = rs;
  static const hipError_t attr_err = [] {   // once per process, thread-safe
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dgrad_dma<2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, dgrad_lds_bytes(2));
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dgrad_dma<3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, dgrad_lds_bytes(3));
    return e;
  }();
  if (attr_err != hipSuccess) return attr_err;
  const dim3 grid(P.tiles_m * nB), block(256);
  if (dgrad_dma_mode() == 2) hipLaunchKernelGGL((k_dgrad_dma<2>), grid, block, dgrad_lds_bytes(2), st, P);
  else hipLaunchKernelGGL((k_dgrad_dma<3>), grid, block, dgrad_lds_bytes(3), st, P);
  return hipGetLastError();
}

}  // namespace rau
