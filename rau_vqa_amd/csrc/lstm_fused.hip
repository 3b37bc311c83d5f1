// lstm_fused.hip -- one LSTM cell step as ONE launch: the recurrent gate GEMM (f32 MFMA
// 16x16x4) with the cell's pointwise half in its epilogue.
//
//   pre[b, g*R + u] = input part (precomputed x W_i2h^T + biases, or the two biases)
//                     + sum_src  A_src[b, :] . W_src[g*R + u, :]
//   i, f, o = sigmoid, g = tanh (gate order of the two reference cells: model/DeepLSTM.lua:46-54
//   i f o | g, model/ATTLSTM.lua:12-19 i g f o);  c = f c_prev + i g;  h = o tanh(c)
//
// Why: on the recurrence's critical path the split-K GEMM + slab round trip + cell kernel cost
// two dependent launches per step and ~80 us next to the bulk GEMMs.  Here a workgroup owns 16
// hidden units x all four gates (64 gathered weight rows) for a block of batch rows over the full
// reduction, so the gate pre-activations never leave registers:
//   * 4 waves; MFMA 16x16x4: a wave's four accumulators are the four gates of the same
//     (16 rows x 16 units), so every lane holds i, f, o, g of its (row, unit) pairs -- no exchange;
//   * one source (K <= one h): waves = 4 row groups of 16 rows (64-row tile);
//     two sources (x2 | h2, or j | h_prev): waves = 2 row groups x 2 sources (32-row tile), the
//     source-1 waves hand their accumulators over through LDS -- halves the K loop of the
//     longer cell so both cells of the encoder wavefront finish together;
//   * W tile through LDS (shared by the row groups), k-major with the k rows permuted so that a
//     lane's float4 of A (k = 16T + 4q .. +3, loaded straight from global: the A rows are not
//     shared between waves) lines up with four conflict-free ds_read_b32 of W.
// Exact f32 (fmaf chains inside the MFMA); summation order differs from the split-K path, both
// are deterministic.
#include "common.h"
#include "kernels.h"

namespace rau {

namespace {

constexpr int FBK = 16;          // k per LDS stage and per source
constexpr int FLD = 64 + 16;     // k-row pitch (floats): 16 mod 32 -> conflict-free fragment reads

template <int ORDER> struct Slots;
template <> struct Slots<GATES_ATT> { enum { I = 0, G = 1, F = 2, O = 3 }; };
template <> struct Slots<GATES_DEEP> { enum { I = 0, F = 1, O = 2, G = 3 }; };

__device__ __forceinline__ float4 sel4(bool ok, const float4& v) {   // branch-free zeroing
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// Gate GEMM of one workgroup: acc[g] = rows [row0, row0+16) x units [u0, u0+16) of gate g.
// TWO = false: one source; a K-step is 32 deep (two 16-k blocks in the two LDS regions), waves are
//   four row groups.  TWO = true: two sources; a K-step is 16 deep per source (LDS region s =
//   source s), waves = 2 row groups x 2 sources; returns false for the source-1 waves after they
//   have handed their sums over.
// Both: every thread loads two float4 of W per K-step; A comes straight from global.  Loads are
// UNCONDITIONAL (addresses clamped into the row, values zeroed afterwards by selects): a load under
// a runtime condition makes hipcc branch around it and drain the whole load queue at every use,
// which would serialise the register ring.  The ring keeps FDEPTH K-steps of loads in flight:
// next to the bulk GEMMs a global load takes thousands of cycles while a K-step computes in
// 600-1100, so with the usual one-step prefetch every step waits for memory (measured: 75-95 us
// per launch, no faster than the two-launch path).
// NSTEPS > 0: the K loop is fully unrolled for that many K-steps (K = 512, the model's real
// widths): with a loop back-edge hipcc cannot count how many ring loads are outstanding and falls
// back to s_waitcnt vmcnt(0) at the first use in every iteration, which drains the ring; unrolled it
// emits exact counted waits.  NSTEPS = 0: generic sizes, runtime loop (correct, not pipelined).
template <bool TWO, int NSTEPS>
__device__ __forceinline__ bool gate_gemm(const LstmStepParams& P, const LstmStepSide& C,
                                          float* Ws, const int rt, const int u0, f32x4 (&acc)[4],
                                          int& row0_out) {
  constexpr int FDEPTH = TWO ? 4 : 3;   // <= 112 VGPRs: two of these workgroups fit beside two bulk tiles
  constexpr int NA = TWO ? 1 : 2;                 // A float4 per K-step and lane
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int src = TWO ? (w >> 1) : 0;
  const int rg = TWO ? (w & 1) : w;
  const int row0 = rt * (TWO ? 32 : 64) + rg * 16;
  row0_out = row0;
  const int lr = l & 15, lq = l >> 4;
  const int K = C.K[src];
  const int Kw = TWO ? (C.K[0] > C.K[1] ? C.K[0] : C.K[1]) : C.K[0];
  const int step_k = TWO ? FBK : 2 * FBK;
  const int nsteps = (Kw + step_k - 1) / step_k;
  // W staging.  TWO: the 128 threads of a source's two waves stage that source's 64 columns x 16 k
  // (thread -> column c and c + 32, k chunk kq_s).  !TWO: 256 threads stage 64 columns x 32 k
  // (thread -> column c, k chunk kq_s of block 0 and of block 1).
  const int ts = TWO ? (tid & 127) : tid;
  const int col = TWO ? (ts >> 2) : (ts >> 2);     // TWO: 0..31, else 0..63
  const int kq_s = ts & 3;
  const int colB = TWO ? col + 32 : col;           // second float4: other column / same column
  const int gA = col >> 4, uA = u0 + (col & 15), gB = colB >> 4, uB = u0 + (colB & 15);
  const bool okA = uA < P.R, okB = uB < P.R;
  const float* Wb = C.W[src];
  const float* wpA = Wb + ((size_t)gA * P.R + (okA ? uA : 0)) * K;
  const float* wpB = Wb + ((size_t)gB * P.R + (okB ? uB : 0)) * K;
  float* ldA = Ws + (TWO ? src * FBK * FLD : 0) + kq_s * FLD + col;
  float* ldB = TWO ? ldA + 32 : ldA + FBK * FLD;
  // A fragments: lane (row lr, quarter lq) takes k = 16 blk + 4 lq .. +3
  const int arow = row0 + lr;
  const bool aok = arow < P.B;
  const float* ap = C.A[src] + (size_t)(aok ? arow : 0) * K;

  auto kofW = [&](int T, int which) { return T * step_k + (TWO ? 0 : which * FBK) + kq_s * 4; };
  auto kofA = [&](int T, int which) { return T * step_k + which * FBK + lq * 4; };
  float4 rwA[FDEPTH], rwB[FDEPTH], ra[FDEPTH][NA];
  auto issue = [&](int T, int slot) {
    rwA[slot] = *reinterpret_cast<const float4*>(wpA + min(kofW(T, 0), K - 4));
    rwB[slot] = *reinterpret_cast<const float4*>(wpB + min(kofW(T, 1), K - 4));
#pragma unroll
    for (int i = 0; i < NA; ++i)
      ra[slot][i] = *reinterpret_cast<const float4*>(ap + min(kofA(T, i), K - 4));
  };
#pragma unroll
  for (int j = 0; j < FDEPTH; ++j) issue(j, j);
  const float* wsrc = Ws + (TWO ? src * FBK * FLD : 0) + lq * FLD + lr;
  auto step = [&](const int T, const int j) {
    __syncthreads();              // everyone is done reading the LDS stage
    {
      // element k = 4 kq + jj of a chunk goes to k-row (4 jj + kq): MFMA jj of a block then
      // reads rows 4 jj + (l >> 4), i.e. the same k the lane's A float4 component jj carries
      float4 vA = sel4(okA && kofW(T, 0) < K, rwA[j]);
      float4 vB = sel4(okB && kofW(T, 1) < K, rwB[j]);
      if (P.bf16) { vA = rb16(vA); vB = rb16(vB); }
      ldA[0] = vA.x; ldA[4 * FLD] = vA.y; ldA[8 * FLD] = vA.z; ldA[12 * FLD] = vA.w;
      ldB[0] = vB.x; ldB[4 * FLD] = vB.y; ldB[8 * FLD] = vB.z; ldB[12 * FLD] = vB.w;
    }
    float4 a[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      a[i] = sel4(aok && kofA(T, i) < K, ra[j][i]);
      if (P.bf16) a[i] = rb16(a[i]);
    }
    issue(T + FDEPTH, j);
    __syncthreads();
    // all of the step's W fragments first (one LDS latency per step, not one per MFMA pair:
    // there is a single wave per SIMD here, nothing else hides it), then the MFMAs back to back
    float bfr[NA][4][4];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) bfr[i][jj][g] = wsrc[i * FBK * FLD + (4 * jj) * FLD + g * 16];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const float av[4] = {a[i].x, a[i].y, a[i].z, a[i].w};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], bfr[i][jj][g], acc[g], 0, 0, 0);
    }
  };
  if constexpr (NSTEPS > 0) {
#pragma unroll
    for (int T = 0; T < NSTEPS; ++T) step(T, T % FDEPTH);
  } else {
    for (int T0 = 0; T0 < nsteps; T0 += FDEPTH) {
#pragma unroll
      for (int j = 0; j < FDEPTH; ++j)
        if (T0 + j < nsteps) step(T0 + j, j);   // uniform
    }
  }
  if (TWO) {                          // source-1 waves hand their partial sums to the source-0 waves
    __syncthreads();
    float* hand = Ws + (size_t)rg * 16 * 64;   // [16 values][64 lanes] per row group
    if (src == 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) hand[(g * 4 + i) * 64 + l] = acc[g][i];
    }
    __syncthreads();
    if (src == 1) return false;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[g][i] += hand[(g * 4 + i) * 64 + l];
  }
  return true;
}

// One (unit tile ut, row tile rt) of one cell.  COH: the outputs other workgroups read in the NEXT step
// of a persistent launch (h, and the dropped-out copy) leave with agent-scope stores (sc1: written
// through to where every XCD's L2 will find them), see k_enc_persist.
template <int ORDER, bool COH>
__device__ __forceinline__ void step_tile(const LstmStepParams& P, const LstmStepSide& C, float* Ws,
                                          const int ut, const int rt) {
  using GS = Slots<ORDER>;
  const int nsrc = C.nsrc;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int u0 = ut * 16;
  const int lr = l & 15, lq = l >> 4;

  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  int row0 = rt * 64 + w * 16;
  if (nsrc == 2) {
    const bool full = C.K[0] == 512 && C.K[1] == 512;
    if (full ? !gate_gemm<true, 32>(P, C, Ws, rt, u0, acc, row0)
             : !gate_gemm<true, 0>(P, C, Ws, rt, u0, acc, row0))
      return;
  } else if (nsrc == 1) {
    if (C.K[0] == 512)
      gate_gemm<false, 16>(P, C, Ws, rt, u0, acc, row0);
    else
      gate_gemm<false, 0>(P, C, Ws, rt, u0, acc, row0);
  }

  // ---- cell: lane holds rows row0 + 4 lq + i (i = 0..3) of unit u0 + lr, all four gates
  const int u = u0 + lr;
  if (u >= P.R) return;
  const int R = P.R;
  float bi = 0.f, bf = 0.f, bo = 0.f, bg = 0.f;
  if (!C.pre) {
    bi = C.b1[GS::I * R + u] + C.b2[GS::I * R + u];
    bf = C.b1[GS::F * R + u] + C.b2[GS::F * R + u];
    bo = C.b1[GS::O * R + u] + C.b2[GS::O * R + u];
    bg = C.b1[GS::G * R + u] + C.b2[GS::G * R + u];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = row0 + 4 * lq + i;
    if (b >= P.B) continue;
    float* gt = C.gates + (size_t)b * 4 * R;
    float pi = acc[GS::I][i], pf = acc[GS::F][i], po = acc[GS::O][i], pg = acc[GS::G][i];
    if (C.pre) {
      const float* pr = C.pre + (size_t)b * 4 * R;
      pi += pr[GS::I * R + u]; pf += pr[GS::F * R + u]; po += pr[GS::O * R + u]; pg += pr[GS::G * R + u];
    } else {
      pi += bi; pf += bf; po += bo; pg += bg;
    }
    const float gi = sigmoidf_(pi), gf = sigmoidf_(pf), go = sigmoidf_(po), gg = tanh_fast(pg);
    gt[GS::I * R + u] = gi; gt[GS::F * R + u] = gf; gt[GS::O * R + u] = go; gt[GS::G * R + u] = gg;
    const float cn = gf * C.c_prev[(size_t)b * R + u] + gi * gg;
    const float tc = tanh_fast(cn);
    const float hn = go * tc;
    const size_t e = (size_t)b * R + u;
    C.c[e] = cn;
    if (COH) __hip_atomic_store(C.h + e, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else C.h[e] = hn;
    C.tanhc[e] = tc;
    if (C.drop_out) {
      float v = hn;
      if (C.mask) v = mask_bit(C.mask, C.mask_e0 + e) ? hn * C.mscale : 0.f;
      if (COH) __hip_atomic_store(C.drop_out + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else C.drop_out[e] = v;
    }
  }
}

template <int ORDER>
__global__ __launch_bounds__(256) void k_lstm_step_fused(const LstmStepParams P) {
  RAU_CHAIN_PRIO();
  const LstmStepSide& C = P.s[blockIdx.y];
  const int rows_per_wg = C.nsrc == 2 ? 32 : 64;
  const int tiles_u = (P.R + 15) / 16;
  const int ut = blockIdx.x % tiles_u, rt = blockIdx.x / tiles_u;
  if (rt * rows_per_wg >= P.B) return;           // grid is sized for the 32-row tiling
  __shared__ __attribute__((aligned(16))) float Ws[2 * FBK * FLD];   // also the hand-over buffer
  step_tile<ORDER, false>(P, C, Ws, ut, rt);
}

}  // namespace

hipError_t lstm_step_fused(
hipStream_t st, int order, const LstmStepParams& P) {
  if (P.n < 1) return hipSuccess;
  if (P.n > 2 || P.R % 4 != 0) return hipErrorInvalidValue;
  for (int i = 0; i < P.n; ++i) {
    const LstmStepSide& s = P.s[i];
    if (s.nsrc < 0 || s.nsrc > 2) return hipErrorInvalidValue;
    for (int k = 0; k < s.nsrc; ++k)
      if (s.K[k] % 4 != 0 || s.K[k] <= 0) return hipErrorInvalidValue;
  }
  const int tiles_u = (P.R + 15) / 16;
  const dim3 grid(tiles_u * ((P.B + 31) / 32), P.n), block(256);
  if (order == GATES_ATT)
    hipLaunchKernelGGL(k_lstm_step_fused<GATES_ATT>, grid, block, 0, st, P);
  else
    hipLaunchKernelGGL(k_lstm_step_fused<GATES_DEEP>, grid, block, 0, st, P);
  return hipGetLastError();
}

}  // namespace rau
