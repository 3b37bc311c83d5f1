// enc_ws.hip -- the question encoder's forward recurrence as ONE weight-stationary persistent launch.
//
// Reference: the 2-layer DeepLSTM unrolled over the question tokens, model/DeepLSTM.lua:29-65 driven by
// SS:448-462.  Per token t: layer-1 gates = G1[t] (x_t W_i2h1^T + biases, formed for all tokens at once
// in front of this kernel) + h1[t-1] W_h2h1^T; layer-2 gates = dropout(h1[t]) W_i2h2^T + h2[t-1] W_h2h2^T
// + biases; i, f, o = sigmoid, g = tanh (split order DeepLSTM.lua:46-54); c = f c- + i g; h = o tanh c.
//
// Why: per wavefront step the three recurrent products are 1.6 GFLOP at B = 256 (0.4 at B = 64) --
// 3.7 to 15 us of matrix-pipe time -- but as launches they cost 40-90 us per step inside the training
// step: every launch re-reads the 12.6 MB of recurrent weights through L2 and pays a kernel boundary,
// 54 dependent launches in all.  Here the weights never move:
//   * a wave keeps ONE 16-column gate tile (4 hidden units x their 4 gates) of one weight matrix, all
//     K = 512 of it, in 128 VGPRs as ready-made MFMA B operands for the whole launch;
//   * workgroups (4 waves) come in two kinds: kind A = 4 layer-1 tiles (16 units), kind B = 2 layer-2
//     tiles x {x2 half, h2 half} (8 units; the two halves of a tile are summed through LDS).  Times two
//     sample halves: (R/16 + R/8) x 2 = 192 workgroups at R = 512, one per CU, each a quarter of a
//     CU's registers -- the other half of every CU stays free for the bulk stream's conv tiles;
//   * per step a workgroup streams the h rows of ITS sample half through LDS (LDS-DMA, double
//     buffered, 32 KB stages), 128 MFMAs (v_mfma_f32_16x16x4_f32) per wave and 16-sample block, the
//     cell in registers (the four gates of a unit sit in four adjacent lanes: DPP quad broadcasts);
//   * what other workgroups need next step -- h1, x2 = dropout(h1), h2 -- leaves with agent-scope
//     (sc1, write-through) stores; one lane per workgroup then bumps a monotonic counter of its
//     (layer, sample half).  Consumers poll the counters they depend on (layer 1 only on layer 1 of
//     its own sample half: 32 arrivals; layer 2 on both), acquire once, and load.  No grid barrier:
//     the two layers and the two sample halves drift apart as far as the data allows.
//     (cdna_hip_programming.md Guideline 16, recipe R1 with the acquire kept.)
// Spins are bounded: if the grid cannot be resident the launch ends with *err = 1 instead of hanging.
// Numerics: per gate pre-activation two interleaved fmaf chains over k (even / odd 4-k groups) summed
// at the end -- not bitwise the split-K path's order, same 1e-4 bar against the oracle.
#include <type_traits>
#include <utility>

#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace rau {

namespace {

constexpr int ER = 512;               // hidden width this kernel is built for (Rq of the reference, SS:209)
constexpr int EKC = 256;              // k per LDS slot
constexpr int EPITCH = EKC + 8;       // slot row pitch in floats: 16-byte slots 2 r + g (mod 16) all distinct for ds_read_b128
constexpr int ESLOT = 16 * EPITCH;    // one slot: 16 samples x 256 k
constexpr int ESTAGE = 2 * ESLOT;     // a stage = two slots (33792 bytes)
constexpr int EOUTG = 16 * 64;        // activated gates of a 16-sample block: [sample][gate][16 units]
constexpr int EOUTV = 16 * 16;        // h | x2 | c | tanh c: [sample][16 units]
constexpr int EHAND = 2 * 4 * 64;     // kind B: partial sums of the h2-half waves, [tile][reg][lane]
constexpr int EOUT = EOUTG + 4 * EOUTV;   // one block's output tile (2048 floats)
constexpr int ESMEM = 2 * ESTAGE + EOUT + EHAND;   // 19456 floats = 77824 bytes: fits a CU beside one wide conv tile
constexpr int kWsSpinMax = 1 << 22;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int ORDER> struct GateSlots;
template <> struct GateSlots<GATES_ATT> { enum { I = 0, G = 1, F = 2, O = 3 }; };
template <> struct GateSlots<GATES_DEEP> { enum { I = 0, F = 1, O = 2, G = 3 }; };

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

// value of lane (quad base + Q) of this lane's quad
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), Q * 0x55, 0xf, 0xf, true));
}

// 8 bytes, written through to where every XCD's L2 finds them (global_store_dwordx2 sc1)
__device__ __forceinline__ void store_sc1(float* p, float a, float b) {
  const float2 v = make_float2(a, b);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS read the compiler does not track (the caller counts lgkmcnt itself)
template <int OFF>
__device__ __forceinline__ void ds_read_f128(float4& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read_b128 offset");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// one lane: wait until the monotonic counter has reached `want`
__device__ __forceinline__ void wait_counter(unsigned* c, unsigned want, int* err) {
  int spins = 0;
  while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kWsSpinMax) {   // give up: wrong numbers, but every wave still reaches the end
      *err = 1;
      break;
    }
  }
}

// BF: both operands of the gate products rounded to bf16 (rau_dtype RAU_BF16, lin_bf16() in kernels.h): the
// stationary weights once while they are loaded, the h / x2 fragments in registers in front of their MFMAs
template <int ORDER, bool KIND_B, bool BF>
__device__ __forceinline__ void enc_ws_body(const EncWsParams& Q, const int wg, float* smem) {
  using GS = GateSlots<ORDER>;
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = l & 15, g = l >> 4, u = col >> 2, q = col & 3;
  const int qslot = q == 0 ? GS::I : q == 1 ? GS::F : q == 2 ? GS::O : GS::G;
  const int P = Q.P, Bp = Q.B / P, nsb = Bp / 16;
  const int part = wg % P, grp = wg / P;
  const int U0 = KIND_B ? 8 * grp : 16 * grp;            // first unit of the workgroup
  const int uw = KIND_B ? 4 * (w >> 1) : 4 * w;          // this wave's units are U0 + uw .. + 3
  const int half = KIND_B ? (w & 1) : 0;                 // kind B: 0 = x2 W_i2h2, 1 = h2 W_h2h2
  const size_t BR = (size_t)Q.B * ER, G4 = BR * 4;
  const int b0 = part * Bp;                              // first sample of this sample half

  // ---- the wave's weights: B operand of MFMA (i, j) = W[row(col)][16 i + 4 g + j]
  float wreg[128];
  {
    const float* W = KIND_B ? (half ? Q.Wh2 : Q.Wi2) : Q.Wh1;
    const float* wrow = W + (size_t)(qslot * ER + U0 + uw + u) * ER + 4 * g;
    sfor<32>([&](auto it) {
      constexpr int i = decltype(it)::value;
      float4 v = *reinterpret_cast<const float4*>(wrow + 16 * i);
      if constexpr (BF) v = rb16(v);
      wreg[4 * i] = v.x; wreg[4 * i + 1] = v.y; wreg[4 * i + 2] = v.z; wreg[4 * i + 3] = v.w;
    });
  }
  float bias = 0.f;
  if (KIND_B) bias = Q.bi2[qslot * ER + U0 + uw + u] + Q.bh2[qslot * ER + U0 + uw + u];

  float* outg = smem + 2 * ESTAGE;      // [gates | h | x2 | c | tanh c]
  float* hand = outg + EOUT;
  unsigned* cnt1 = Q.cnt + part;        // layer-1 cells done, this sample half
  unsigned* cnt2 = Q.cnt + 8 + part;    // layer-2 cells done
  const unsigned nA = ER / 16, nB = ER / 8;   // workgroups per (layer, sample half)

  // LDS-DMA of one stage: 32 rows of 1 KB (slot, sample r), wave w takes rows 8 w .. 8 w + 7
  auto issue_stage = [&](int buf, const float* src0, const float* src1, int k0, int k1, int row0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = 8 * w + i, slot = rr >> 4, r = rr & 15;
      const float* src = (slot ? src1 + k1 : src0 + k0) + (size_t)(row0 + r) * ER + 4 * l;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src,
                                       (lds_ptr_t)(smem + buf * ESTAGE + slot * ESLOT + r * EPITCH), 16, 0, 16 /* sc1 */);
    }
  };

#pragma unroll 1
  for (int t = 1; t <= Q.TL; ++t) {
    // ---- dependencies of cell t of this layer.  No acquire behind the poll: every h / x2 row is
    // written once per launch, at an address no CU has read in this launch, with write-through
    // (sc1) stores, and is read below by sc1 loads (LDS-DMA, L1-bypassing) only.
    if (tid == 0) {
      if (!KIND_B) {
        if (t > 1) wait_counter(cnt1, nA * (unsigned)(t - 1), Q.err);       // h1[t-1]
      } else {
        wait_counter(cnt1, nA * (unsigned)t, Q.err);                         // x2[t]
        if (t > 1) wait_counter(cnt2, nB * (unsigned)(t - 1), Q.err);       // h2[t-1]
      }
    }
    __syncthreads();
    const float* h1p = Q.h1 + (size_t)(t - 1) * BR;      // h1[t-1]
    const float* x2t = Q.x2 + (size_t)(t - 1) * BR;      // x2[t] (0-based token slot t-1)
    const float* h2p = Q.h2 + (size_t)(t - 1) * BR;      // h2[t-1]
    const float* cprev = (KIND_B ? Q.c2 : Q.c1) + (size_t)(t - 1) * BR;
    float* gates = (KIND_B ? Q.G2 : Q.G1) + (size_t)(t - 1) * G4;
    float* c_out = (KIND_B ? Q.c2 : Q.c1) + (size_t)t * BR;
    float* h_out = (KIND_B ? Q.h2 : Q.h1) + (size_t)t * BR;
    float* tc_out = (KIND_B ? Q.tc2 : Q.tc1) + (size_t)(t - 1) * BR;
    float* x2_out = Q.x2 + (size_t)(t - 1) * BR;

    // stage s of kind A = sample block s (slots = k halves of h1); of kind B = (sample block s/2,
    // k half s%2) with slots = (x2, h2)
    auto issue = [&](int s) {
      if (!KIND_B) issue_stage(s & 1, h1p, h1p, 0, EKC, b0 + 16 * s);
      else issue_stage(s & 1, x2t, h2p, EKC * (s & 1), EKC * (s & 1), b0 + 16 * (s >> 1));
    };
    // a finished block's outputs leave in 16-byte pieces of 4 consecutive units, one block LATE: its
    // stores are issued in front of the next block's MFMAs, so that the wait for a stage (which is
    // a wait for everything this wave has in flight) finds them long done
    auto flush = [&](int sb) {
      const float* og = outg;
      const float* ov = og + EOUTG;
      const int bs = b0 + 16 * sb;
      constexpr int NU4 = KIND_B ? 2 : 4;                 // 16-byte pieces per sample row of the workgroup
      const int sg = tid >> 4, qq = (tid >> 2) & 3, u4 = tid & 3;
      if (u4 < NU4) {
        const int gs = qq == 0 ? GS::I : qq == 1 ? GS::F : qq == 2 ? GS::O : GS::G;
        const float4 v = *reinterpret_cast<const float4*>(og + (sg * 4 + qq) * 16 + 4 * u4);
        *reinterpret_cast<float4*>(gates + (size_t)(bs + sg) * 4 * ER + gs * ER + U0 + 4 * u4) = v;
      }
      const int arr = tid >> 6, idx = tid & 63, s2 = idx >> 2, v4 = idx & 3;
      if (v4 < NU4 && !(KIND_B && arr == 1)) {
        const float4 v = *reinterpret_cast<const float4*>(ov + arr * EOUTV + s2 * 16 + 4 * v4);
        const size_t e = (size_t)(bs + s2) * ER + U0 + 4 * v4;
        if (arr == 0) { store_sc1(h_out + e, v.x, v.y); store_sc1(h_out + e + 2, v.z, v.w); }
        else if (arr == 1) { store_sc1(x2_out + e, v.x, v.y); store_sc1(x2_out + e + 2, v.z, v.w); }
        else if (arr == 2) *reinterpret_cast<float4*>(c_out + e) = v;
        else *reinterpret_cast<float4*>(tc_out + e) = v;
      }
    };
    const int nst = KIND_B ? 2 * nsb : nsb;
    issue(0);
    f32x4 acc0, acc1;
    const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
    const uint32_t afrag = lds0 + (uint32_t)(col * EPITCH + 4 * g) * 4;   // row = sample `col` of the block
#pragma unroll 1
    for (int sb = 0; sb < nsb; ++sb) {
      const int bs = b0 + 16 * sb;                        // first sample of the block
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      // this lane's share of the epilogue's inputs, requested before the MFMAs
      float pre[4], cp[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t b = (size_t)(bs + 4 * g + r);
        pre[r] = KIND_B ? bias : gates[b * 4 * ER + qslot * ER + U0 + uw + u];
        cp[r] = cprev[b * ER + U0 + uw + u];
      }
      // one slot's MFMAs: 16 four-k groups, A fragments by ds_read_b128 one group ahead.  The reads are
      // inline asm with hand-counted waits: hipcc would wait vmcnt(0) -- for the NEXT stage's LDS-DMA,
      // just issued -- in front of every LDS read it can see, and it reads just in time.
      auto mma_slot = [&](auto i0_tag, uint32_t slot_addr) {
        constexpr int I0 = decltype(i0_tag)::value;
        float4 va, vb;
        ds_read_f128<0>(va, slot_addr);
        sfor<16>([&](auto it) {
          constexpr int ii = decltype(it)::value;
          float4& cur = (ii % 2 == 0) ? va : vb;
          float4& nxt = (ii % 2 == 0) ? vb : va;
          if constexpr (ii + 1 < 16) {
            ds_read_f128<64 * (ii + 1)>(nxt, slot_addr);
            asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (BF) cur = rb16(cur);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.x, wreg[4 * (I0 + ii)], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.y, wreg[4 * (I0 + ii) + 1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.z, wreg[4 * (I0 + ii) + 2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.w, wreg[4 * (I0 + ii) + 3], acc1, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
      };
      using I0 = std::integral_constant<int, 0>;
      using I16 = std::integral_constant<int, 16>;
      if (!KIND_B) {
        const int s = sb;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of stage s has landed
        __syncthreads();                                   // everyone's has; stage s-1 and block sb-1's out tile are done
        if (sb > 0) flush(sb - 1);
        if (s + 1 < nst) issue(s + 1);
        const uint32_t st = afrag + (uint32_t)((s & 1) * ESTAGE) * 4;
        mma_slot(I0{}, st);
        mma_slot(I16{}, st + ESLOT * 4);
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int s = 2 * sb + c;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (c == 0 && sb > 0) flush(sb - 1);
          if (s + 1 < nst) issue(s + 1);
          const uint32_t st = afrag + (uint32_t)((s & 1) * ESTAGE + half * ESLOT) * 4;
          if (c == 0) mma_slot(I0{}, st); else mma_slot(I16{}, st);
        }
      }
      f32x4 acc = acc0 + acc1;
      bool cell = true;
      if (KIND_B) {   // the h2-half waves hand their sums to the x2-half waves of the same tile
        if (half) {
#pragma unroll
          for (int r = 0; r < 4; ++r) hand[((w >> 1) * 4 + r) * 64 + l] = acc[r];
        }
        __syncthreads();
        if (!half) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += hand[((w >> 1) * 4 + r) * 64 + l];
        }
        cell = !half;
      } else {
        __syncthreads();   // every wave has long read block sb-1's tile (flush sits in front of the MFMAs)
      }
      if (cell) {
        // lane (unit u, gate q) holds that gate's pre-activation for samples 4 g + r of the block
        float* og = outg;
        float* ov = og + EOUTG;
        float act[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = acc[r] + pre[r];
          act[r] = q == 3 ? tanh_fast(x) : sigmoidf_(x);
          og[((4 * g + r) * 4 + q) * 16 + uw + u] = act[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gi = quad_bcast<0>(act[r]), gf = quad_bcast<1>(act[r]);
          const float go = quad_bcast<2>(act[r]), gg = quad_bcast<3>(act[r]);
          const float cn = gf * cp[r] + gi * gg;
          const float tc = tanh_fast(cn);
          const float hn = go * tc;
          if (q == 0) {
            const int o = (4 * g + r) * 16 + uw + u;
            ov[o] = hn;
            ov[2 * EOUTV + o] = cn;
            ov[3 * EOUTV + o] = tc;
            if (!KIND_B) {
              float xv = hn;
              if (Q.mask) {
                const size_t e = (size_t)(t - 1) * BR + (size_t)(bs + 4 * g + r) * ER + U0 + uw + u;
                xv = mask_bit(Q.mask, e) ? hn * Q.mscale : 0.f;
              }
              ov[EOUTV + o] = xv;
            }
          }
        }
      }
    }
    // ---- the last block's outputs, then publish: every storing wave drains, ONE lane counts the
    // workgroup in
    __syncthreads();
    flush(nsb - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(KIND_B ? cnt2 : cnt1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int ORDER, bool BF = false>
__global__ __launch_bounds__(256, 2) void k_enc_ws(const EncWsParams Q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  RAU_CHAIN_PRIO();
  const int nA = (ER / 16) * Q.P;
  if ((int)blockIdx.x < nA) enc_ws_body<ORDER, false, BF>(Q, blockIdx.x, smem);
  else enc_ws_body<ORDER, true, BF>(Q, blockIdx.x - nA, smem);
}

}  // namespace

// Shapes the weight-stationary encoder takes: the reference's hidden width, batches in whole
// 16-sample MFMA blocks (two sample halves when B is a multiple of 32).
bool enc_ws_ok(int B, int R) { return R == ER && B >= 16 && B % 16 == 0; }
int enc_ws_workgroups(int B) { return (ER / 16 + ER / 8) * (B % 32 == 0 ? 2 : 1); }

// The kernel's dynamic-LDS attribute, set once per process (a C++11 magic static: contexts may be
// driven from several host threads).
static hipError_t enc_ws_attr() {
  static const hipError_t err = [] {
    for (const void* f : {reinterpret_cast<const void*>(k_enc_ws<GATES_ATT>),
                          reinterpret_cast<const void*>(k_enc_ws<GATES_DEEP>),
                          reinterpret_cast<const void*>(k_enc_ws<GATES_ATT, true>),
                          reinterpret_cast<const void*>(k_enc_ws<GATES_DEEP, true>)}) {
      const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, ESMEM * 4);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }();
  return err;
}

// Workgroups of one launch wait on progress counters bumped by OTHER workgroups of the same launch
// (higher-numbered ones too), so all of them must be resident at once.  The pure predicate: with
// `blocks_per_cu` workgroups admitted per CU (occupancy query) on `n_cus` CUs.  It is deliberately
// conservative: at most ONE workgroup per CU is counted, because in the step the kernel shares every
// CU with a resident bulk tile that leaves room for exactly one (83 + 76 KB of LDS).
bool enc_ws_coresident(int B, int blocks_per_cu, int n_cus) {
  if (B < 1 || blocks_per_cu < 1 || n_cus < 1) return false;
  return (long)std::min(blocks_per_cu, 1) * n_cus >= enc_ws_workgroups(B);
}
// The same question asked of the current device (CPX partitions and other reduced-CU devices say no).
bool enc_ws_fits_device(int B) {
  if (enc_ws_attr() != hipSuccess) return false;
  int dev = 0, per_cu = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_enc_ws<GATES_DEEP>),
                                                   256, ESMEM * 4) != hipSuccess)
    return false;
  return enc_ws_coresident(B, per_cu, prop.multiProcessorCount);
}

// Q.cnt: 16 unsigned words owned by the caller, zeroed HERE (a memset node in front of the launch,
// so a captured graph replays it too).
hipError_t enc_ws_forward(hipStream_t st, int order, EncWsParams Q) {
  if (!enc_ws_ok(Q.B, Q.R) || Q.TL < 1 || !Q.cnt || !Q.err) return hipErrorInvalidValue;
  Q.P = Q.B % 32 == 0 ? 2 : 1;
  hipError_t e = enc_ws_attr();
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(Q.cnt, 0, 16 * sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  const dim3 grid(enc_ws_workgroups(Q.B)), block(256);
  if (Q.bf16) {
    if (order == GATES_ATT) hipLaunchKernelGGL((k_enc_ws<GATES_ATT, true>), grid, block, ESMEM * 4, st, Q);
    else hipLaunchKernelGGL((k_enc_ws<GATES_DEEP, true>), grid, block, ESMEM * 4, st, Q);
  } else if (order == GATES_ATT) hipLaunchKernelGGL(k_enc_ws<GATES_ATT>, grid, block, ESMEM * 4, st, Q);
  else hipLaunchKernelGGL(k_enc_ws<GATES_DEEP>, grid, block, ESMEM * 4, st, Q);
  return hipGetLastError();
}

}  // namespace rau
