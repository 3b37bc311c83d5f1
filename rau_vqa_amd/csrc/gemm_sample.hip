// gemm_sample.hip -- the position-flattened 1x1-conv GEMMs with ONE SAMPLE PER TILE COLUMN BLOCK.
//
//   C[b, m, s] = epi( sum_k Wt[k, m] * X[b, k, s] )        m < M, s < S, one sample b per tile
//
// (i_embed, reference SS:240: Wt = Wi^T, X = dropped-out feature map, epi = tanh(. + bi);
//  ifeatproj, SS:247: Wt = Wp^T, X = I, epi = . + bp;  its input gradient: Wt = Wp, X = dS,
//  epi = . + dj[b,m] a[b,s].)
//
// Why a second tiling next to gemm_core.h's 128 x 128 flattened-column tiles: a 14 x 14 map has
// S = 196 = 4 * 49 positions, so the flattened column count B * 196 is a multiple of 49 and the
// number of 128-wide tiles per hop (1568 = 32 * 49) never fills a whole number of rounds of the
// 512 resident workgroups (3.06 rounds per hop: 12-23 % of a launch is a ragged last round), and
// every tile edge cuts 784-byte rows of C at unaligned offsets (1.35x HBM write traffic measured).
// Here a tile is 128 rows x ALL positions of one sample:
//   * tiles per hop = (M / 128) * B = 1024 for M = 512, B = 256: exactly two rounds;
//   * a tile reads whole 784-byte rows of X and writes one contiguous 128 x 784-byte block of C
//     (full lines except the block's two ends), no (sample, position) index arithmetic anywhere;
//   * 196 columns = 12.25 MFMA blocks of 16: computed as 13 (208 columns, 6 % padded work, the pad
//     columns of the LDS tile are zero) with v_mfma_f32_16x16x4_f32 -- same FLOP rate as 32x32x2.
// 4 waves x (32 rows x 208 columns): 2 x 13 accumulators of 4 registers.  K-step 16, LDS tiles
// k-major with pitches = 16 mod 32 floats (conflict-free 16-lane fragment reads), double buffered,
// one barrier per K-step, register-staged global loads one step ahead -- the structure of
// gemm_core.h.  Exact f32 (fmaf chains).  Used when 176 < S <= 208 and S % 4 == 0.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace rau {

namespace {

constexpr int SBK = 16, SNCB = 13, SBN = SNCB * 16;   // 208 columns
constexpr int SLDB = SBN + 32;      // 240 = 16 mod 32

struct SampleParams {
  int M, K, S, nB, tiles_m;
  const float* Wt; long w_rs;          // [K][M]
  const float* X; long x_bs;           // [b][K][S]
  float* C; long c_bs;                 // [b][M][S]
  const float* bias; int act;          // EPI 0
  const float* dj; const float* av;    // EPI 1, 2: + dj[b,m] * a[b,s]
  const float* Y; float* rs;           // EPI 2: * (1 - Y[b,m,s]^2); rs[b,m] = sum_s of the result
  int c16;                             // EPI 2: C is stored as bf16 ([b][M][S] 2-byte elements, RNE)
};

// RB = 16-row blocks per wave: tile rows SBM = 4 waves x RB x 16 (128 or 64)
template <int EPI, int RB>
__global__ __launch_bounds__(256, 2) void k_conv_sample(const SampleParams P) {
  constexpr int SBM = 64 * RB;
  constexpr int SLDA = SBM + 16;      // 144 / 80 = 16 mod 32
  constexpr int SSTAGE = SBK * (SLDA + SLDB);   // floats per stage
  __shared__ __attribute__((aligned(16))) float smem[2 * SSTAGE];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int lr = l & 15, lq = l >> 4;
  const int nwg = P.tiles_m * P.nB;
  const int id = xcd_remap(blockIdx.x, nwg);      // tiles of one sample share an XCD's L2
  const int tm = id % P.tiles_m, b = id / P.tiles_m;
  const int m0 = tm * SBM;
  const int S = P.S, S4 = S >> 2;
  const float* Xb = P.X + (size_t)b * P.x_bs;

  // staging maps (the same every K-step): A = 16 k-rows x SBM/4 float4, B = 16 k-rows x S4 float4
  constexpr int ACL = SBM / 4;                                  // float4 per A row (32 or 16)
  constexpr int ARS = 256 / ACL;                                // A rows per pass (8 or 16)
  const int a_row = tid / ACL, a_c4 = (tid % ACL) * 4;          // rows a_row (+ ARS)
  const bool a_ok = m0 + a_c4 < P.M;                           // M % 4 == 0
  const float* a_ptr = P.Wt + (size_t)a_row * P.w_rs + m0 + (a_ok ? a_c4 : 0);
  int b_row[4], b_q[4];
  bool b_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int item = tid + i * 256;
    b_row[i] = item / S4;
    b_q[i] = item - b_row[i] * S4;
    b_ok[i] = b_row[i] < SBK;
    if (!b_ok[i]) { b_row[i] = 0; b_q[i] = 0; }
  }
  // zero the pad columns [S, 208) of both B stages once: no load ever writes them
  for (int e = tid; e < 2 * SBK * (SBN - S); e += 256) {
    const int st = e / (SBK * (SBN - S)), r = e % (SBK * (SBN - S));
    smem[st * SSTAGE + SBK * SLDA + (r / (SBN - S)) * SLDB + S + r % (SBN - S)] = 0.f;
  }

  f32x4 acc[RB][SNCB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < SNCB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = (P.K + SBK - 1) / SBK;
  float4 ra[RB], rb[4];
  auto load = [&](int T) {
    const int k0 = T * SBK;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int k = k0 + a_row + ARS * i;
      ra[i] = (a_ok && k < P.K) ? *reinterpret_cast<const float4*>(a_ptr + (size_t)(k0 + ARS * i) * P.w_rs)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + b_row[i];
      rb[i] = (b_ok[i] && k < P.K) ? *reinterpret_cast<const float4*>(Xb + (size_t)k * S + b_q[i] * 4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store = [&](int stage) {
    float* As = smem + stage * SSTAGE;
    float* Bs = As + SBK * SLDA;
#pragma unroll
    for (int i = 0; i < RB; ++i)
      *reinterpret_cast<float4*>(As + (a_row + ARS * i) * SLDA + a_c4) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (b_ok[i]) *reinterpret_cast<float4*>(Bs + b_row[i] * SLDB + b_q[i] * 4) = rb[i];
  };
  auto compute = [&](int stage) {
    const float* As = smem + stage * SSTAGE + lq * SLDA + w * 16 * RB + lr;
    const float* Bs = smem + stage * SSTAGE + SBK * SLDA + lq * SLDB + lr;
#pragma unroll
    for (int kb = 0; kb < SBK / 4; ++kb) {
      float a[RB], bb[SNCB];
#pragma unroll
      for (int i = 0; i < RB; ++i) a[i] = As[kb * 4 * SLDA + i * 16];
#pragma unroll
      for (int j = 0; j < SNCB; ++j) bb[j] = Bs[kb * 4 * SLDB + j * 16];
#pragma unroll
      for (int j = 0; j < SNCB; ++j)
#pragma unroll
        for (int i = 0; i < RB; ++i)
          // X as the MFMA's A operand, W as its B operand: the accumulator block is C^T, i.e. a
          // lane's 4 registers are 4 CONSECUTIVE positions of one row m -> 16-byte stores
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[j], a[i], acc[i][j], 0, 0, 0);
    }
  };

  if (nsteps > 0) {
    load(0);
    store(0);
  }
  __syncthreads();
  for (int T = 0; T + 1 < nsteps; ++T) {
    const int cur = T & 1;
    load(T + 1);
    compute(cur);
    store(cur ^ 1);
    __syncthreads();
  }
  // EPI 2 reads this tile's block of Y (100 KB) in its epilogue: the first row block's 13 loads per
  // lane go out in front of the last K-step's MFMAs (the staging registers are dead by now), the next
  // block's once the previous block is stored -- one exposed round trip per further block instead of
  // one per batch of loads the compiler forms inside the epilogue loop, and the tile stays under ~180
  // registers (all blocks in flight at once: 226, and two tiles then leave the recurrence's kernels 60).
  float4 yv[SNCB];
  auto yload = [&](int i) {
#pragma unroll
    for (int j = 0; j < SNCB; ++j) {
      const int s = j * 16 + 4 * lq, m = m0 + w * 16 * RB + i * 16 + lr;
      yv[j] = (s < S && m < P.M) ? *reinterpret_cast<const float4*>(P.Y + (size_t)b * P.c_bs + (size_t)m * S + s)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (EPI == 2) {
    yload(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (nsteps > 0) compute((nsteps - 1) & 1);
  __syncthreads();

  // ---- epilogue: accumulator (i, j) register r = C[m0 + w*16*RB + i*16 + lr][j*16 + 4*lq + r]
  float* rowv = smem;            // [SBM] bias / dj of this sample's rows
  float* colv = smem + SBM;      // [208] a of this sample's positions
  if (tid < SBM) {
    const int m = m0 + tid;
    float v = 0.f;
    if (m < P.M) {
      if (EPI == 0) v = P.bias ? P.bias[m] : 0.f;
      else v = P.dj[(size_t)b * P.M + m];
    }
    rowv[tid] = v;
  }
  if (EPI >= 1 && tid < SBN) colv[tid] = tid < S ? P.av[(size_t)b * S + tid] : 0.f;
  __syncthreads();
  float* Cb = P.C + (size_t)b * P.c_bs;
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    if (EPI == 2 && i > 0) {
      asm volatile("" ::: "memory");          // the previous block's stores are issued: its registers are free
      __builtin_amdgcn_sched_barrier(0);
      yload(i);
    }
    const int rl = w * 16 * RB + i * 16 + lr;
    const int m = m0 + rl;
    const bool mok = m < P.M;
    const float rv = rowv[rl];
    float* crow = Cb + (size_t)(mok ? m : 0) * S;
    float rsum = 0.f;
#pragma unroll
    for (int j = 0; j < SNCB; ++j) {
      const int s = j * 16 + 4 * lq;
      if (s >= S || !mok) continue;   // S % 4 == 0: a float4 is all valid or all pad
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (EPI == 0) {
        v.x += rv; v.y += rv; v.z += rv; v.w += rv;
        if (P.act) { v.x = tanh_fast(v.x); v.y = tanh_fast(v.y); v.z = tanh_fast(v.z); v.w = tanh_fast(v.w); }
      } else {
        const float4 c4 = *reinterpret_cast<const float4*>(colv + s);
        v.x += rv * c4.x; v.y += rv * c4.y; v.z += rv * c4.z; v.w += rv * c4.w;
        if (EPI == 2) {   // gradient through i_embed's tanh, and its row sums (the bias gradient)
          const float4 y = yv[j];
          v.x *= 1.f - y.x * y.x; v.y *= 1.f - y.y * y.y;
          v.z *= 1.f - y.z * y.z; v.w *= 1.f - y.w * y.w;
          rsum += (v.x + v.y) + (v.z + v.w);
        }
      }
      if (EPI == 2 && P.c16) {
        typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
        b16x4 o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        uint16_t* c16 = reinterpret_cast<uint16_t*>(P.C) + (size_t)b * P.c_bs + (size_t)m * S + s;
        *reinterpret_cast<uint2*>(c16) = __builtin_bit_cast(uint2, o);
      } else {
        *reinterpret_cast<float4*>(crow + s) = v;
      }
    }
    if (EPI == 2) {   // the four lanes lr, lr+16, lr+32, lr+48 hold the row's four position quarters
      rsum += __shfl_xor(rsum, 16, 64);
      rsum += __shfl_xor(rsum, 32, 64);
      if (lq == 0 && mok) P.rs[(size_t)b * P.M + m] = rsum;
    }
  }
}

}  // namespace

// which: 1 = i_embed forward, 2 = ifeatproj forward, 4 = attention dgrad, 8 = i_embed dgrad.
// RAU_CONV_SAMPLE=<mask> selects which of them use this tiling (default below).
bool conv_sample_ok(int S, int which) {
  static const int mask = [] { const char* e = std::getenv("RAU_CONV_SAMPLE");
                               return e ? std::atoi(e) : 12; }();
  return (mask & which) && S % 4 == 0 && S > 176 && S <= SBN;
}

// EPI 0: C = act(acc + bias[m]);  EPI 1: C = acc + dj[b,m] a[b,s];
// EPI 2: C = (acc + dj[b,m] a[b,s]) (1 - Y[b,m,s]^2) and rs[b,m] = sum_s C[b,m,s]
hipError_t conv_sample(hipStream_t st, int epi, int nB, int M, int K, int S, const float* Wt,
                       long w_rs, const float* X, long x_bs, float* C, long c_bs,
                       const float* bias, int act, const float* dj, const float* av,
                       const float* Y, float* rs, int c16) {
  if (!(S % 4 == 0 && S > 176 && S <= SBN) || M % 4 != 0) return hipErrorInvalidValue;
  constexpr int rb = 2;   // 128-row tiles (64-row tiles measured slower in the step)
  SampleParams P{};
  P.M = M; P.K = K; P.S = S; P.nB = nB;
  P.tiles_m = (M + 64 * rb - 1) / (64 * rb);
  P.Wt = Wt; P.w_rs = w_rs;
  P.X = X; P.x_bs = x_bs;
  P.C = C; P.c_bs = c_bs;
  P.bias = bias; P.act = act;
  P.dj = dj; P.av = av;
  P.Y = Y; P.rs = rs; P.c16 = c16;
  const dim3 grid(P.tiles_m * nB), block(256);
#define LAUNCH(E, R) hipLaunchKernelGGL((k_conv_sample<E, R>), grid, block, 0, st, P)
  if (epi == 0) LAUNCH(0, 2); else if (epi == 1) LAUNCH(1, 2); else LAUNCH(2, 2);
#undef LAUNCH
  return hipGetLastError();
}

}  // namespace rau
