// dgrad16.hip -- RAU_BF16 mode: attbycontent's input gradient on 14 x 14 maps with bf16 MFMA operands.
//
//   dZ[b, m, s] = ( sum_k Wp[k, m] * dS[b, k, s]  +  dj[b, m] * a[b, s] ) * (1 - I[b, m, s]^2)
//   rs[b, m]    = sum_s dZ[b, m, s]
//
// (reference SS:565-579: the gradient of ifeatproj + attselect w.r.t. i_embed's output, then through
//  i_embed's tanh; rs is the per-sample part of i_embed's bias gradient.)  Rounds 1-3 ran this product
// on the exact-f32 per-sample kernel in every dtype (gemm_sample.hip EPI 2): 105 GFLOP per step on the
// f32 matrix pipe, 1.6 of the 9.1 ms of the configs[2] step.  Here both GEMM operands are rounded to
// bf16 (RNE) while they are staged into LDS, as the mode's other four conv GEMMs do, and the product
// runs on v_mfma_f32_16x16x32_bf16 with f32 accumulation; the epilogue (rank-1 term, tanh derivative,
// row sums) stays f32.  The kernel is then bound by its operands' bytes, not by the matrix pipe.
//
// Tiling = gemm_sample.hip's: 128 rows x ALL positions of ONE sample per workgroup (1024 tiles per
// hop at M = 512, B = 256; whole 784-byte rows of dS in, one contiguous block of dZ out), 4 waves x
// (32 rows x 208 columns = 2 x 13 accumulator blocks).  K-step 32: an LDS element is the 16-byte
// k-octet {k .. k+7} of one position (X image, 4 octets x 208) or one row (W image, 4 x 128 + padding), which
// is exactly one lane's MFMA operand (lane l: position / row l & 15, octet l >> 4), read with ds_read_b128;
// within an octet the elements are quad-planar (see GXQ below) so that the staging stores and the fragment
// reads are both free of bank conflicts.
// Both sources are k-major in memory ([k][s], [k][m]): a thread loads the same four columns of eight
// consecutive k rows and transposes in registers.  Two stages, register-staged loads one K-step ahead,
// one barrier per K-step.
#include "common.h"
#include "kernels.h"

namespace rau {

namespace {

constexpr int GS = 196, GS4 = GS / 4;          // positions per sample
constexpr int GNCB = 13, GBN = GNCB * 16;      // 208 columns (12 zero pad columns)
constexpr int GBM = 128;                       // rows per tile
constexpr int GBK = 32, GKO = GBK / 8;         // K-step, octets per K-step
// LDS images, 16-byte elements (one k-octet of one position / row), QUAD-PLANAR: a staging thread holds
// four consecutive positions (rows) of one octet, and element (octet ko, position p) sits at
//   ko * GXP + (p & 3) * GXQ + (p >> 2)
// so that the 64 lanes of a staging store write CONSECUTIVE 16-byte elements (round 3 stored a thread's
// four elements side by side: lanes 64 bytes apart, a 4-way bank conflict on every store, 0.50 conflict
// cycles per LDS cycle in the round-3 counters).  A fragment read takes lanes lr = 0..15 at
// (lr & 3) * GXQ + 4 j + (lr >> 2): with GXQ = 52 = 4 mod 16 (GWQ = 36 = 4 mod 16) the sixteen lanes of
// every ds_read_b128 lane group land in sixteen different 16-byte bank groups.
constexpr int GXQ = GBN / 4;                   // 52 elements per quad plane of the X image
constexpr int GXP = 4 * GXQ;                   // 208 per octet
constexpr int GWQ = GBM / 4 + 4;               // 36 per quad plane of the W image (32 + 4: = 4 mod 16)
constexpr int GWP = 4 * GWQ;                   // 144 per octet
constexpr int GXST = GKO * GXP;                // uint4 elements of the X image (832)
constexpr int GWST = GKO * GWP;                // of the W image (576)
constexpr int GSTAGE = GXST + GWST;            // 1408 x 16 B = 22.5 KB
static_assert(GXQ % 16 == 4 && GWQ % 16 == 4, "quad-plane pitch: conflict-free ds_read_b128 fragments");

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct Dgrad16Params {
  int M, K, nB, tiles_m;
  const float* Wt; long w_rs;          // [K][M]  (Wp as stored)
  const void* X; long x_bs;            // [b][K][S]  (dS), f32 or (x16) bf16 elements
  void* C; long c_bs;                  // [b][M][S]  dZ, f32 or bf16 elements
  const float* dj; const float* av;    // [b][M], [b][S]
  const float* Y; float* rs;           // I [b][M][S]; rs [b][M]
  int c16, x16;
};

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}
// the same from rows that already hold bf16: c-th 16-bit field of eight uint2 rows
__device__ __forceinline__ uint32_t pick2(uint2 lo, uint2 hi, int c) {
  const uint32_t a = c < 2 ? lo.x : lo.y, b = c < 2 ? hi.x : hi.y;
  return (c & 1) ? __builtin_amdgcn_perm(b, a, 0x07060302u) : __builtin_amdgcn_perm(b, a, 0x05040100u);
}
#define OCTET16(r, c) make_uint4(pick2(r[0], r[1], c), pick2(r[2], r[3], c), pick2(r[4], r[5], c), pick2(r[6], r[7], c))
// column c (0..3) of eight float4 rows -> the k-octet of that column
#define OCTET(r, c)                                                                          \
  make_uint4(pack2(r[0].c, r[1].c), pack2(r[2].c, r[3].c), pack2(r[4].c, r[5].c), pack2(r[6].c, r[7].c))

template <bool X16>
__global__ __launch_bounds__(256, 2) void k_dgrad16(const Dgrad16Params P) {
  __shared__ __attribute__((aligned(16))) uint4 smem[2 * GSTAGE];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int lr = l & 15, lq = l >> 4;
  const int nwg = P.tiles_m * P.nB;
  const int id = xcd_remap(blockIdx.x, nwg);      // the row tiles of one sample share an XCD's L2
  const int tm = id % P.tiles_m, b = id / P.tiles_m;
  const int m0 = tm * GBM;
  const float* Xb = reinterpret_cast<const float*>(P.X) + (size_t)b * P.x_bs;
  const uint16_t* Xb16 = reinterpret_cast<const uint16_t*>(P.X) + (size_t)b * P.x_bs;

  // staging items: X (octet ko, position quad q4) for tid < 4 * 49; W (octet ko, row quad mq) for tid < 128
  const bool x_on = tid < GKO * GS4, w_on = tid < GKO * (GBM / 4);
  const int x_ko = x_on ? tid / GS4 : 0, x_q4 = x_on ? tid - x_ko * GS4 : 0;
  const int w_ko = w_on ? tid / (GBM / 4) : 0, w_mq = w_on ? tid % (GBM / 4) : 0;
  const float* x_ptr = Xb + (size_t)(8 * x_ko) * GS + 4 * x_q4;
  const uint16_t* x_ptr16 = Xb16 + (size_t)(8 * x_ko) * GS + 4 * x_q4;
  const float* w_ptr = P.Wt + (size_t)(8 * w_ko) * P.w_rs + m0 + 4 * w_mq;

  // zero the pad positions [196, 208) of both X images once (quads 49..51 of every plane): no load ever
  // writes them
  for (int e = tid; e < 2 * GKO * (GBN - GS); e += 256) {
    const int st = e / (GKO * (GBN - GS)), r = e % (GKO * (GBN - GS));
    const int ko = r / (GBN - GS), p = GS + r % (GBN - GS);
    smem[st * GSTAGE + ko * GXP + (p & 3) * GXQ + (p >> 2)] = make_uint4(0u, 0u, 0u, 0u);
  }

  f32x4 acc[2][GNCB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < GNCB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = P.K / GBK;
  float4 rx[8], rw[8];
  uint2 rx16[8];
  auto load = [&](int T) {
    const size_t k0 = (size_t)T * GBK;
    if (X16) {
      if (x_on) {
#pragma unroll
        for (int r = 0; r < 8; ++r) rx16[r] = *reinterpret_cast<const uint2*>(x_ptr16 + (k0 + r) * GS);
      }
    } else if (x_on) {
#pragma unroll
      for (int r = 0; r < 8; ++r) rx[r] = *reinterpret_cast<const float4*>(x_ptr + (k0 + r) * GS);
    }
    if (w_on) {
#pragma unroll
      for (int r = 0; r < 8; ++r) rw[r] = *reinterpret_cast<const float4*>(w_ptr + (k0 + r) * P.w_rs);
    }
  };
  auto store = [&](int stage) {
    uint4* Xs = smem + stage * GSTAGE;
    uint4* Ws = Xs + GXST;
    if (x_on) {   // positions 4 q4 .. 4 q4 + 3 -> quad q4 of planes 0..3
      uint4* d = Xs + x_ko * GXP + x_q4;
      if (X16) {
        d[0] = OCTET16(rx16, 0); d[GXQ] = OCTET16(rx16, 1); d[2 * GXQ] = OCTET16(rx16, 2); d[3 * GXQ] = OCTET16(rx16, 3);
      } else {
        d[0] = OCTET(rx, x); d[GXQ] = OCTET(rx, y); d[2 * GXQ] = OCTET(rx, z); d[3 * GXQ] = OCTET(rx, w);
      }
    }
    if (w_on) {
      uint4* d = Ws + w_ko * GWP + w_mq;
      d[0] = OCTET(rw, x); d[GWQ] = OCTET(rw, y); d[2 * GWQ] = OCTET(rw, z); d[3 * GWQ] = OCTET(rw, w);
    }
  };
  auto compute = [&](int stage) {
    // position 16 j + lr -> plane lr & 3, quad 4 j + (lr >> 2); row 32 w + 16 i + lr -> plane lr & 3,
    // quad 8 w + 4 i + (lr >> 2)
    const uint4* Xs = smem + stage * GSTAGE + lq * GXP + (lr & 3) * GXQ + (lr >> 2);
    const uint4* Ws = smem + stage * GSTAGE + GXST + lq * GWP + (lr & 3) * GWQ + 8 * w + (lr >> 2);
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, Ws[0]);
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, Ws[4]);
#pragma unroll
    for (int j = 0; j < GNCB; ++j) {
      // dS as the MFMA's A operand, Wp as its B operand: the accumulator block is C^T, a lane's four
      // registers are four CONSECUTIVE positions of one row m (16-byte / 8-byte stores)
      const bf16x8 xb = __builtin_bit_cast(bf16x8, Xs[4 * j]);
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, a0, acc[0][j], 0, 0, 0);
      acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, a1, acc[1][j], 0, 0, 0);
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
  // The epilogue reads this tile's 100 KB of I.  Its 26 loads per lane go out HERE, in front of the
  // last K-step's MFMAs (the staging registers are dead by now), so their latency is covered once
  // instead of once per batch the compiler would otherwise form inside the epilogue.
  const float* Yb = P.Y + (size_t)b * P.c_bs;
  // Register budget: with both row blocks' 26 loads of I in flight the kernel needs 242 registers, two
  // workgroups fill a SIMD's 512, and no kernel of the recurrence can start on the CU until a tile
  // retires (measured: one tile per CU is 0.2 ms slower here and the step 0.11 ms FASTER).  So the
  // first row block's loads go out ahead of the last K-step, and the second block's only when the
  // first block's accumulators and I values are dead: ~190 registers, and two tiles leave a third of
  // the register file free.
  float4 yv[GNCB];
  auto yload = [&](int i) {
#pragma unroll
    for (int j = 0; j < GNCB; ++j) {
      const int s = 16 * j + 4 * lq;
      yv[j] = s < GS ? *reinterpret_cast<const float4*>(Yb + (size_t)(m0 + 32 * w + 16 * i + lr) * GS + s)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  yload(0);
  __builtin_amdgcn_sched_barrier(0);
  if (nsteps > 0) compute((nsteps - 1) & 1);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  // ---- epilogue (as gemm_sample.hip EPI 2): accumulator (i, j) register r =
  // C[m0 + 32 w + 16 i + lr][16 j + 4 lq + r]
  float* rowv = reinterpret_cast<float*>(smem);   // [128] dj of this sample's rows
  float* colv = rowv + GBM;                        // [208] a of this sample's positions
  if (tid < GBM) rowv[tid] = P.dj[(size_t)b * P.M + m0 + tid];
  if (tid < GBN) colv[tid] = tid < GS ? P.av[(size_t)b * GS + tid] : 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i == 1) {
      asm volatile("" ::: "memory");          // block 0's stores are issued: its registers are free
      __builtin_amdgcn_sched_barrier(0);
      yload(1);
    }
    const int rl = 32 * w + 16 * i + lr;
    const int m = m0 + rl;
    const float rv = rowv[rl];
    float rsum = 0.f;
#pragma unroll
    for (int j = 0; j < GNCB; ++j) {
      const int s = 16 * j + 4 * lq;
      if (s >= GS) continue;   // 196 % 4 == 0: a quad is all valid or all pad
      const float4 c4 = *reinterpret_cast<const float4*>(colv + s);
      const float4 y = yv[j];
      float4 v = make_float4(acc[i][j][0] + rv * c4.x, acc[i][j][1] + rv * c4.y,
                             acc[i][j][2] + rv * c4.z, acc[i][j][3] + rv * c4.w);
      v.x *= 1.f - y.x * y.x; v.y *= 1.f - y.y * y.y;
      v.z *= 1.f - y.z * y.z; v.w *= 1.f - y.w * y.w;
      rsum += (v.x + v.y) + (v.z + v.w);
      const size_t e = (size_t)b * P.c_bs + (size_t)m * GS + s;
      if (P.c16)
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(P.C) + e) =
            make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
      else
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(P.C) + e) = v;
    }
    // the four lanes lr, lr + 16, lr + 32, lr + 48 hold the row's four position quarters
    rsum += __shfl_xor(rsum, 16, 64);
    rsum += __shfl_xor(rsum, 32, 64);
    if (lq == 0) P.rs[(size_t)b * P.M + m] = rsum;
  }
}
#undef OCTET
#undef OCTET16

}  // namespace

bool dgrad16_ok(int M, int K, int S, long w_rs) {
  return S == GS && M % GBM == 0 && K % GBK == 0 && K >= GBK && w_rs % 4 == 0;
}

hipError_t dgrad16(hipStream_t st, int nB, int M, int K, int S, const float* Wt, long w_rs,
                   const void* X, long x_bs, void* C, long c_bs, const float* dj, const float* av,
                   const float* Y, float* rs, int c16, int x16) {
  if (!dgrad16_ok(M, K, S, w_rs)) return hipErrorInvalidValue;
  if (nB == 0) return hipSuccess;
  Dgrad16Params P{};
  P.M = M; P.K = K; P.nB = nB; P.tiles_m = M / GBM;
  P.Wt = Wt; P.w_rs = w_rs;
  P.X = X; P.x_bs = x_bs;
  P.C = C; P.c_bs = c_bs;
  P.dj = dj; P.av = av; P.Y = Y; P.rs = rs; P.c16 = c16; P.x16 = x16;
  const int pad = 0;
  if (x16) hipLaunchKernelGGL(k_dgrad16<true>, dim3(P.tiles_m * nB), dim3(256), pad, st, P);
  else hipLaunchKernelGGL(k_dgrad16<false>, dim3(P.tiles_m * nB), dim3(256), pad, st, P);
  return hipGetLastError();
}

}  // namespace rau
