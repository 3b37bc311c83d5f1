// gemm_core.h -- f32-MFMA GEMM engine for gfx950 (v_mfma_f32_32x32x2_f32).
//
// One workgroup = 256 threads = 4 waves (2x2) computes a BM x BN tile of
//   C[m,n] = sum_k A(m,k) * B(k,n)
// Operands are staged global -> registers -> LDS as k-major tiles
// As[BK][BM+pad], Bs[BK][BN+pad] (double buffered, one barrier per K-step),
// so every MFMA fragment read is a conflict-free ds_read_b32 (lane l reads
// row k = l>>5, column l&31).  f32 MFMA runs at 256 FLOP/clk/CU, i.e. one
// 32x32x2 MFMA per 64 cycles per SIMD, so LDS bandwidth is never the limit
// here; the tile shape is chosen for HBM/L2 intensity instead.
//
// "Loader" classes define how A / B elements are produced (plain, dropout-
// masked, or computed on the fly), "Epilogue" classes define what happens to
// the accumulators.  Numerics: exact f32 (bitwise a k-ordered fmaf chain).
#pragma once
#include <type_traits>

#include "common.h"

namespace rau {

constexpr int LPAD = 4;   // row padding (floats) of tiles staged with 16-byte LDS stores
// Tiles staged by TRANSPOSING scalar stores (LoadKC, LoadSC: a thread's float4 of 4 consecutive k
// goes to 4 k-rows of one column) use a pad of 1 instead: with a pitch of 4 mod 32 the 8 k-chunks
// of a row land on only two banks (4-way conflicts, measured SQ_LDS_BANK_CONFLICT = 52 % of the LDS
// cycles of the conv weight-gradient kernels); with an odd pitch (k-chunk + row) spreads over all 32.
constexpr int SPAD = 1;

// ---- bf16-operand mode (rau_dtype RAU_BF16): operands are rounded to bf16 (RNE) while they
// are staged into LDS, products accumulate in f32 (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA
// rate, so these kernels become HBM-bound).  LDS image of a [BT rows][BKT k] operand tile:
// BKT/4 planes of 8-byte elements (pitch and in-plane order: Img policies below), element (p, r) = the four bf16 values
// k = 4p..4p+3 of row r.  A lane's MFMA fragment (8 consecutive k of one row) is two
// conflict-free ds_read_b64 from planes 2h and 2h+1.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BPAD = 4;
// Where element (plane p, row / column c) of a bf16 image sits: p * PITCH + at(c), 8-byte elements.  The
// LOADER chooses (its staging stores must be free of bank conflicts); the fragment read -- 32 lanes taking
// 32 consecutive rows of one plane with ds_read_b64 -- is conflict-free under both choices.
//  * ImgRows: a plane is its rows in order.  For loaders whose 8 lanes of a row write 8 PLANES (reduction
//    index contiguous in memory: LoadKC, LoadSC, LoadSC16): a 16-lane store group is 8 planes x 2 rows, and
//    2 * PITCH = 4 (mod 32) dwords spreads it over all 32 banks (round 3 had PITCH = BT + 4: planes p and
//    p + 4 on the same banks, 0.25 conflict cycles per LDS cycle in the counters).
//  * ImgQuads: a plane is four quarter-planes, row c in quarter c & 3 at quad (c >> 2) ^ 4 [c & 2].  For
//    loaders that hold four CONSECUTIVE rows of one plane per thread (k-major sources transposed in
//    registers: LoadRC<K4>, LoadRC16): the lanes of a store then write 16 consecutive elements (permuted)
//    instead of elements 32 bytes apart (a 2-way conflict on every store: 0.24 in the counters).  The
//    fragment read must be conflict-free in BOTH forms hipcc emits it in: ds_read_b64 (32 lanes, 64 banks:
//    Q = 8 (mod 32) puts the four quarters 16 banks apart) and ds_read2_b64, which it forms from two row
//    blocks (16-lane groups, 32 banks: quarters 0 / 2 and 1 / 3 would then share banks -- 0.39 conflicts
//    measured with the plain quad order -- so quarters 2, 3 swap the halves of every group of 8 quads).
template <int BT> struct ImgRows {
  static constexpr int PITCH = BT + 2;
  static_assert(BT != 128 || (2 * PITCH) % 32 == 4, "8 planes x 2 rows of a store group on 32 different banks");
  static __device__ __forceinline__ int at(int c) { return c; }
};
template <int BT> struct ImgQuads {
  static constexpr int Q = BT / 4 + 8;
  static constexpr int PITCH = 4 * Q;
  static_assert(BT != 128 || Q % 32 == 8, "32 consecutive rows of a fragment read on 32 different bank pairs");
  // (the bf16 modes only exist for 128-wide tiles; narrower instantiations never use their image)
  static __device__ __forceinline__ int at(int c) { return (c & 3) * Q + ((c >> 2) ^ ((c & 2) << 1)); }
};
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
  bf16x4 v;
  v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
  return __builtin_bit_cast(uint2, v);
}
// ---- split-operand mode (rau_dtype RAU_F32S): every f32 operand x is staged as THREE bf16 terms
// x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8 significand
// bits = all 24 bits of the f32 value), and each product a*b is formed as the six bf16 MFMA terms
// of relative weight >= 2^-16: hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi, accumulated in f32.
// The terms left out (mid*lo, lo*mid, lo*lo) are <= 2^-24 of the product -- the size of one f32
// rounding -- so the result has f32-grade accuracy (measured against the fp64 oracle in
// tests/test_gpu_split.py) at 6 bf16 MFMAs per 8 f32 MFMAs of a quarter the length: 2.7x fewer
// matrix-pipe cycles.  LDS image: three plane sets (hi, mid, lo) of the bf16 layout above.
// put_bf16<NPL>: NPL = 1 rounds (RAU_BF16), NPL = 3 splits (RAU_F32S).
template <int NPL>
__device__ __forceinline__ void put_bf16(uint2* img, int idx, int set_stride, float a, float b,
                                         float c, float d) {
  if constexpr (NPL == 1) {
    img[idx] = pack_bf16x4(a, b, c, d);
  } else {
    bf16x4 h, m, l;
    const float x[4] = {a, b, c, d};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h[i] = (__bf16)x[i];
      const float r1 = x[i] - (float)h[i];
      m[i] = (__bf16)r1;
      l[i] = (__bf16)(r1 - (float)m[i]);
    }
    img[idx] = __builtin_bit_cast(uint2, h);
    img[idx + set_stride] = __builtin_bit_cast(uint2, m);
    img[idx + 2 * set_stride] = __builtin_bit_cast(uint2, l);
  }
}


// Superset of the arguments any loader/epilogue combination needs.
struct GemmParams {
  int M, N, K;          // output rows / cols, reduction length (per sample in SC mode)
  int nk;               // number of BK-steps in the whole reduction
  int nk_per_split;     // BK-steps handled by one blockIdx.z
  int tiles_m, tiles_n;
  // batched launch (blockIdx.y): same shapes, different operand / slab bases
  int nbatch; const float* Ab[3]; const float* Bb[3]; long slab_batch_stride;
  // heterogeneous batch (Nb[0] != 0): problem y has Nb[y] output columns and its split-K
  // partials at C + slab_off[y] ([split][M][Nb[y]]); N is the widest (sizes the grid)
  int Nb[3]; long slab_off[3];
  // A operand
  const float* A; long a_rs; long a_bs;
  const float* A2;      // SC_DTANH loader: second source, A * (1 - A2^2) (same indexing as A)
  // B operand
  const float* B; long b_rs; long b_bs;
  // flattened (sample, position) column space: n -> (n / S, n % S)
  int S;
  int cps;              // BK-chunks per sample (SC loaders)
    // epilogue
  float* C; long c_rs; long c_bs; long slab_stride;
  const float* bias; const float* bias2;
  const float* addend; long add_rs;
  const float* ymul; long y_rs;       // multiply by (1 - y^2)
  const uint32_t* emask; size_t emask_e0; float emscale;
  int accumulate; int act;            // act: 0 none, 1 tanh
  float alpha;
  // conv epilogue extras
  const float* v1;     // dj [sample][M]
  const float* v2;     // a  [sample][S]
  float* rs_out;       // SC_DTANH A operand: row sums of the staged operand, partial [split][M]
  int round16;         // f32 tiles: both operands rounded to bf16 while they are staged (lin_bf16(), kernels.h)
  int dbg;             // tools/kbench only: 1 = no global loads in the loop, 2 = no barriers
};

// ---------------------------------------------------------------- loaders
// f32 staging registers of a loader rounded to bf16 in place (GemmParams::round16; loaders whose registers
// are float4: the Linear layers' LoadKC / LoadRC)
template <typename Regs>
__device__ __forceinline__ auto round_regs(Regs& R) -> decltype((void)R.v[0].x) {
  constexpr int n = sizeof(R.v) / sizeof(R.v[0]);
#pragma unroll
  for (int i = 0; i < n; ++i) R.v[i] = rb16(R.v[i]);
}
__device__ __forceinline__ void round_regs(...) {}

// Every loader: init(...), load(step) global->regs, store(lds) regs->LDS tile
// [BKT][BT+LPAD].  BKT = K-step per LDS stage.

// Operand stored [rows][K], K contiguous.
template <int BT, int BKT>
struct LoadKC {
  using Img = ImgRows<BT>;
  static constexpr int PAD = SPAD;
  static constexpr int LPR = BKT / 4;          // lanes (float4) per row
  static constexpr int RPP = 256 / LPR;        // rows per pass
  static constexpr int NI = BT / RPP;
  struct Regs { float4 v[NI]; };
  const float* p[NI];
  bool ok[NI];
  int kc, K;
  __device__ __forceinline__ void init(const GemmParams& P, const float* base, long rs, long bs,
                                       int row0, int rows, int tid) {
    K = P.K;
    kc = (tid % LPR) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = tid / LPR + i * RPP;
      ok[i] = row0 + r < rows;
      p[i] = base + (long)(row0 + r) * rs + kc;
    }
  }
  template <bool FAST>
  __device__ __forceinline__ void load(int step, Regs& R) const {
    const int k0 = step * BKT;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      R.v[i] = (FAST || (ok[i] && k0 + kc < K)) ? *reinterpret_cast<const float4*>(p[i] + k0)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void store(float* lds, int tid, const Regs& R) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = tid / LPR + i * RPP;
      float* d = lds + kc * (BT + PAD) + r;
      d[0] = R.v[i].x;
      d[BT + PAD] = R.v[i].y;
      d[2 * (BT + PAD)] = R.v[i].z;
      d[3 * (BT + PAD)] = R.v[i].w;
    }
  }
  template <int NPL>
  __device__ __forceinline__ void store_bf16(uint2* img, int set_stride, int tid, const Regs& R) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = tid / LPR + i * RPP;
      put_bf16<NPL>(img, (tid % LPR) * Img::PITCH + r, set_stride, R.v[i].x, R.v[i].y, R.v[i].z,
                    R.v[i].w);
    }
  }
};

// Operand stored [K][cols], cols contiguous.  FLAT: cols are a flattened
// (sample, position) index, element (k, n) at base + (n/S)*bs + k*rs + n%S.
// K4 (bf16 mode): a thread's NI rows are the consecutive k = 4*kr .. 4*kr+3 instead of
// kr + i*RPP, so that it holds a whole 4-k plane element for each of its four columns.
// CS: keep running column sums of everything staged (the bias gradient of a Linear falls out of
// its weight-gradient GEMM's dY operand).
template <int BT, int BKT, bool FLAT, bool K4 = false, bool CS = false>
struct LoadRC {
  using Img = ImgQuads<BT>;
  static constexpr int PAD = LPAD;
  static constexpr int CPR = BT / 4;
  static constexpr int RPP = 256 / CPR;
  static constexpr int NI = BKT / RPP;
  static_assert(!K4 || NI == 4, "K4 mapping needs four rows per thread");
  struct Regs { float4 v[NI]; };
  const float* p;
  long rs;
  bool ok;
  int kr, c4, K;
  mutable float4 csum;
  __device__ __forceinline__ void init(const GemmParams& P, const float* base, long rs_,
                                       long bs, int col0, int cols, int tid) {
    K = P.K;
    rs = rs_;
    csum = make_float4(0.f, 0.f, 0.f, 0.f);
    c4 = (tid % CPR) * 4;
    kr = tid / CPR;
    const int col = col0 + c4;
    ok = col < cols;
    long off = col;
    if (FLAT) off = (long)(col / P.S) * bs + (col % P.S);
    if (!ok) off = 0;
    p = base + off;
  }
  template <bool FAST>
  __device__ __forceinline__ void load(int step, Regs& R) const {
    const int k0 = step * BKT;
    if (FAST) {  // interior tile: no predicates, one pointer per step
      const float* q = p + (long)(k0 + (K4 ? 4 * kr : kr)) * rs;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        R.v[i] = *reinterpret_cast<const float4*>(q + (long)i * (K4 ? 1 : RPP) * rs);
      return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = K4 ? k0 + 4 * kr + i : k0 + kr + i * RPP;
      R.v[i] = (ok && k < K) ? *reinterpret_cast<const float4*>(p + (long)k * rs)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid, const Regs& R) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      *reinterpret_cast<float4*>(lds + (kr + i * RPP) * (BT + LPAD) + c4) = R.v[i];
      if (CS) {
        csum.x += R.v[i].x; csum.y += R.v[i].y; csum.z += R.v[i].z; csum.w += R.v[i].w;
      }
    }
  }
  template <int NPL>
  __device__ __forceinline__ void store_bf16(uint2* img, int set_stride, int tid, const Regs& R) const {
    if constexpr (K4) {
      if (CS) {   // column sums of the UNROUNDED values: a Linear's bias gradient stays exact f32
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          csum.x += R.v[i].x; csum.y += R.v[i].y; csum.z += R.v[i].z; csum.w += R.v[i].w;
        }
      }
      const int d = kr * Img::PITCH;   // plane kr; columns c4..c4+3 = quad c4/4 of the four quarter-planes
      put_bf16<NPL>(img, d + Img::at(c4), set_stride, R.v[0].x, R.v[1].x, R.v[2].x, R.v[3].x);
      put_bf16<NPL>(img, d + Img::at(c4 + 1), set_stride, R.v[0].y, R.v[1].y, R.v[2].y, R.v[3].y);
      put_bf16<NPL>(img, d + Img::at(c4 + 2), set_stride, R.v[0].z, R.v[1].z, R.v[2].z, R.v[3].z);
      put_bf16<NPL>(img, d + Img::at(c4 + 3), set_stride, R.v[0].w, R.v[1].w, R.v[2].w, R.v[3].w);
    }
  }
};

// "Sample-chunk" loader for the weight gradients of the 1x1 convolutions:
// operand stored [sample][rows][S], reduction over (sample, position); step g
// covers positions [BKT*(g % cps), +BKT) of sample g / cps, zero-filled past S.
// 8 lanes per row (BKT <= 32); lanes whose chunk lies beyond BKT stay idle.
// DT: the operand is x * (1 - y^2) with y read from a second tensor of the same
// layout (gradient through tanh applied while staging, at LDS-store time).
template <int BT, int BKT, bool DT = false>
struct LoadSC {
  using Img = ImgRows<BT>;
  static constexpr int PAD = SPAD;
  static constexpr int LPR = 8;
  static constexpr int RPP = 256 / LPR;
  static constexpr int NI = BT / RPP;
  struct Regs { float4 v[NI]; float4 y[DT ? NI : 1]; };
  long roff[NI];
  bool ok[NI];
  const float* base;
  const float* base2;
  long bs;
  int kc, S, cps;
  mutable float rsum[NI];   // DT: running sum over k of this thread's share of row r (bias gradient)
  __device__ __forceinline__ void init(const GemmParams& P, const float* base_, long rs,
                                       long bs_, int row0, int rows, int tid) {
    base = base_;
    base2 = P.A2;
    bs = bs_;
    S = P.S;
    cps = P.cps;
    kc = (tid % LPR) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = tid / LPR + i * RPP;
      ok[i] = row0 + r < rows;
      roff[i] = (long)(row0 + r) * S + kc;
      rsum[i] = 0.f;
    }
  }
  template <bool FAST>
  __device__ __forceinline__ void load(int g, Regs& R) const {
    const int b = g / cps;
    const int s0 = (g - b * cps) * BKT;
    if (FAST) {  // full rows, S % BKT == 0: idle lanes (kc >= BKT) re-read the last chunk
      const long e0 = (long)b * bs + s0 - (kc >= BKT ? kc - (BKT - 4) : 0);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        R.v[i] = *reinterpret_cast<const float4*>(base + e0 + roff[i]);
        if (DT) R.y[i] = *reinterpret_cast<const float4*>(base2 + e0 + roff[i]);
      }
      return;
    }
    const bool kin = kc < BKT && s0 + kc < S;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const long e = (long)b * bs + roff[i] + s0;
      R.v[i] = (ok[i] && kin) ? *reinterpret_cast<const float4*>(base + e)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
      if (DT)
        R.y[i] = (ok[i] && kin) ? *reinterpret_cast<const float4*>(base2 + e)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid, const Regs& R) const {
    if (kc >= BKT) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = tid / LPR + i * RPP;
      float4 x = R.v[i];
      if (DT) {
        x.x *= (1.f - R.y[i].x * R.y[i].x);
        x.y *= (1.f - R.y[i].y * R.y[i].y);
        x.z *= (1.f - R.y[i].z * R.y[i].z);
        x.w *= (1.f - R.y[i].w * R.y[i].w);
        rsum[i] += (x.x + x.y) + (x.z + x.w);
      }
      float* d = lds + kc * (BT + PAD) + r;
      d[0] = x.x;
      d[BT + PAD] = x.y;
      d[2 * (BT + PAD)] = x.z;
      d[3 * (BT + PAD)] = x.w;
    }
  }
  template <int NPL>
  __device__ __forceinline__ void store_bf16(uint2* img, int set_stride, int tid, const Regs& R) const {
    if (kc >= BKT) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = tid / LPR + i * RPP;
      float4 x = R.v[i];
      if (DT) {
        x.x *= (1.f - R.y[i].x * R.y[i].x);
        x.y *= (1.f - R.y[i].y * R.y[i].y);
        x.z *= (1.f - R.y[i].z * R.y[i].z);
        x.w *= (1.f - R.y[i].w * R.y[i].w);
        rsum[i] += (x.x + x.y) + (x.z + x.w);
      }
      put_bf16<NPL>(img, (kc >> 2) * Img::PITCH + r, set_stride, x.x, x.y, x.z, x.w);
    }
  }
};

// bf16 mode, operand ALREADY STORED AS bf16 in HBM (the dropped-out feature maps, xd16): the same
// two mappings as LoadRC<FLAT, K4> and LoadSC, with half the bytes per load and no conversion --
// the [K][cols] form only regroups the 16-bit halves of its four k-rows into 4-k plane elements
// (two v_perm_b32 per element), the sample-chunk form stores what it loaded.
template <int BT, int BKT, bool FLAT = true>
struct LoadRC16 {
  using Img = ImgQuads<BT>;
  static constexpr int PAD = LPAD;
  static constexpr int CPR = BT / 4;
  static constexpr int RPP = 256 / CPR;
  static constexpr int NI = BKT / RPP;
  static_assert(NI == 4, "4-k plane mapping needs four rows per thread");
  struct Regs { uint2 v[NI]; };
  const uint16_t* p;
  long rs;
  bool ok;
  int kr, c4, K;
  __device__ __forceinline__ void init(const GemmParams& P, const float* base, long rs_,
                                       long bs, int col0, int cols, int tid) {
    K = P.K;
    rs = rs_;
    c4 = (tid % CPR) * 4;
    kr = tid / CPR;
    const int col = col0 + c4;
    ok = col < cols;
    long off = col;
    if (FLAT) off = (long)(col / P.S) * bs + (col % P.S);
    if (!ok) off = 0;
    p = reinterpret_cast<const uint16_t*>(base) + off;
  }
  template <bool FAST>
  __device__ __forceinline__ void load(int step, Regs& R) const {
    const int k0 = step * BKT + 4 * kr;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      R.v[i] = (FAST || (ok && k0 + i < K)) ? *reinterpret_cast<const uint2*>(p + (long)(k0 + i) * rs)
                                            : make_uint2(0u, 0u);
  }
  __device__ __forceinline__ void store(float*, int, const Regs&) const {}
  template <int NPL>
  __device__ __forceinline__ void store_bf16(uint2* img, int, int, const Regs& R) const {
    static_assert(NPL == 1, "stored-bf16 operands exist in RAU_BF16 mode only");
    constexpr uint32_t LO = 0x05040100u, HI = 0x07060302u;   // (src1.lo16 | src0.lo16 << 16), same of hi16
    const int d = kr * Img::PITCH;   // plane kr; columns c4..c4+3 = quad c4/4 of the four quarter-planes
    img[d + Img::at(c4)] = make_uint2(__builtin_amdgcn_perm(R.v[1].x, R.v[0].x, LO),
                                      __builtin_amdgcn_perm(R.v[3].x, R.v[2].x, LO));
    img[d + Img::at(c4 + 1)] = make_uint2(__builtin_amdgcn_perm(R.v[1].x, R.v[0].x, HI),
                                          __builtin_amdgcn_perm(R.v[3].x, R.v[2].x, HI));
    img[d + Img::at(c4 + 2)] = make_uint2(__builtin_amdgcn_perm(R.v[1].y, R.v[0].y, LO),
                                          __builtin_amdgcn_perm(R.v[3].y, R.v[2].y, LO));
    img[d + Img::at(c4 + 3)] = make_uint2(__builtin_amdgcn_perm(R.v[1].y, R.v[0].y, HI),
                                          __builtin_amdgcn_perm(R.v[3].y, R.v[2].y, HI));
  }
};
template <int BT, int BKT>
struct LoadSC16 {
  using Img = ImgRows<BT>;
  static constexpr int PAD = SPAD;
  static constexpr int LPR = 8;
  static constexpr int RPP = 256 / LPR;
  static constexpr int NI = BT / RPP;
  struct Regs { uint2 v[NI]; };
  long roff[NI];
  bool ok[NI];
  const uint16_t* base;
  long bs;
  int kc, S, cps;
  __device__ __forceinline__ void init(const GemmParams& P, const float* base_, long rs,
                                       long bs_, int row0, int rows, int tid) {
    base = reinterpret_cast<const uint16_t*>(base_);
    bs = bs_;
    S = P.S;
    cps = P.cps;
    kc = (tid % LPR) * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = tid / LPR + i * RPP;
      ok[i] = row0 + r < rows;
      roff[i] = (long)(row0 + r) * S + kc;
    }
  }
  template <bool FAST>
  __device__ __forceinline__ void load(int g, Regs& R) const {
    const int b = g / cps;
    const int s0 = (g - b * cps) * BKT;
    const bool kin = kc < BKT && s0 + kc < S;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      R.v[i] = ((FAST || ok[i]) && kin) ? *reinterpret_cast<const uint2*>(base + (long)b * bs + roff[i] + s0)
                                        : make_uint2(0u, 0u);
  }
  __device__ __forceinline__ void store(float*, int, const Regs&) const {}
  template <int NPL>
  __device__ __forceinline__ void store_bf16(uint2* img, int, int tid, const Regs& R) const {
    static_assert(NPL == 1, "stored-bf16 operands exist in RAU_BF16 mode only");
    if (kc >= BKT) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) img[(kc >> 2) * Img::PITCH + tid / LPR + i * RPP] = R.v[i];
  }
};

// A/B "source kinds" used to pick a loader in the kernel template.
enum Src : int {
  SRC_KC = 0,        // [rows][K]
  SRC_RC = 1,        // [K][cols]
  SRC_RC_FLAT = 2,   // [sample][K][S], flattened columns
  SRC_SC = 3,        // [sample][rows][S], reduction over (sample, position)
  SRC_SC_DTANH = 4,  // same, operand = A * (1 - A2^2)
  SRC_RC_SUM = 5,    // [K][cols] + column sums of the operand handed back (P.rs_out)
  SRC_RC_FLAT_B16 = 6,  // SRC_RC_FLAT, operand stored as bf16 (RAU_BF16 mode)
  SRC_SC_B16 = 7,    // SRC_SC, operand stored as bf16
  SRC_RC_B16 = 8     // SRC_RC, operand stored as bf16 (pre-converted weights)
};
template <int BT, int BKT, int SRC, bool K4 = false> struct LoaderOf;
template <int BT, int BKT, bool K4> struct LoaderOf<BT, BKT, SRC_KC, K4> { using type = LoadKC<BT, BKT>; };
template <int BT, int BKT, bool K4> struct LoaderOf<BT, BKT, SRC_RC, K4> { using type = LoadRC<BT, BKT, false, K4>; };
template <int BT, int BKT, bool K4> struct LoaderOf<BT, BKT, SRC_RC_FLAT, K4> { using type = LoadRC<BT, BKT, true, K4>; };
template <int BT, int BKT, bool K4> struct LoaderOf<BT, BKT, SRC_SC, K4> { using type = LoadSC<BT, BKT>; };
template <int BT, int BKT, bool K4> struct LoaderOf<BT, BKT, SRC_SC_DTANH, K4> { using type = LoadSC<BT, BKT, true>; };
template <int BT, int BKT, bool K4> struct LoaderOf<BT, BKT, SRC_RC_SUM, K4> { using type = LoadRC<BT, BKT, false, K4, true>; };
template <int BT, int BKT, bool K4> struct LoaderOf<BT, BKT, SRC_RC_FLAT_B16, K4> { using type = LoadRC16<BT, BKT>; };
template <int BT, int BKT, bool K4> struct LoaderOf<BT, BKT, SRC_SC_B16, K4> { using type = LoadSC16<BT, BKT>; };
template <int BT, int BKT, bool K4> struct LoaderOf<BT, BKT, SRC_RC_B16, K4> { using type = LoadRC16<BT, BKT, false>; };

// -------------------------------------------------------------- epilogues
enum Epi : int {
  EPI_LIN = 0,       // generic pointwise epilogue, row-major C
  EPI_SLAB = 1,      // split-K partial: raw accumulators to slab blockIdx.z
  EPI_CONV = 2,      // C = act(acc + bias[m]) in [sample][M][S]   (act: 0 none, 1 tanh)
  EPI_OUTER = 3      // C = acc + dj[b,m] a[n]   (dI' of the attention backward)
};

// One workgroup's tile; (bx, by, bz) = (tile id, batch problem, K split) -- blockIdx in the plain
// launch, a table lookup in the grouped launch (gemm_group_kernel).
template <int BM, int BN, int BKT, int ASRC, int BSRC, int EPI,
          int DT = 0 /* 1: bf16-rounded operands, 2: 3 x bf16 split operands */>
__device__ __forceinline__ void gemm_tile(const GemmParams& P, const int bx, const int by,
                                          const int bz) {
  constexpr int BK = BKT;
  static_assert(DT == 0 || (BM == 128 && BN == 128 && BKT == 32), "bf16 mode: 128x128x32 tiles");
  // long-reduction kernels (conv weight gradients) only: measured +9% there, 0 on the short-K convs
  constexpr bool PIN = (ASRC == SRC_SC || ASRC == SRC_SC_DTANH);
  if (BM < 128) RAU_CHAIN_PRIO();  // skinny tiles = chain-stream GEMMs
  constexpr int WM = BM / 2, WN = BN / 2;   // wave tile
  constexpr int IM = WM / 32, JN = WN / 32; // 32x32 blocks per wave
  using LAT = typename LoaderOf<BM, BKT, ASRC, DT != 0>::type;
  using LBT = typename LoaderOf<BN, BKT, BSRC, DT != 0>::type;
  constexpr int LDA = BM + LAT::PAD, LDB = BN + LBT::PAD;   // k-row pitch of the f32 LDS tiles
  // 128-wide (bulk) tiles double-buffer their LDS stages; the 64-wide chain tiles keep ONE
  // stage (8.7 KB): next to two resident bulk workgroups (2 x 67.6 of 160 KB) only a
  // footprint under 12 KB lets two chain workgroups share a CU, and these launches are
  // latency-bound, so the second barrier per K-step costs nothing measurable.
  // (split mode: three plane sets per operand = 50.7 KB per stage, so a single stage)
  constexpr int NST = DT == 2 ? 1 : (BM >= 128 ? 2 : 1);
  constexpr int NPL = DT == 2 ? 3 : 1;          // bf16 plane sets per operand (hi | hi, mid, lo)
  // bf16 modes: BK/4 planes of Img::PITCH 8-byte elements per operand (the loader's image policy), plane
  // set and stage
  constexpr int PPA = LAT::Img::PITCH, PPB = LBT::Img::PITCH;                // uint2 per plane
  constexpr int PLA = PPA * (BK / 4), PLB = PPB * (BK / 4);                   // uint2 per set
  constexpr int kStage = DT ? 2 * NST * NPL * (PLA + PLB)
                            : NST * BK * LDA + NST * BK * LDB;  // floats of operand staging
  // epilogue scratch lives in the (then idle) staging area: per-row vectors,
  // per-(sample,row) vectors of the samples this tile's columns touch, column sums
  constexpr int kUCap = kStage - BM;
  __shared__ __attribute__((aligned(16))) float smem[kStage];
  float* As = smem;
  float* Bs = smem + NST * BK * LDA;
  uint2* Ab16 = reinterpret_cast<uint2*>(smem);
  uint2* Bb16 = Ab16 + NST * NPL * PLA;

  const int tid = threadIdx.x;
  const int l = tid & 63, w = tid >> 6;
  const int wm = w & 1, wn = w >> 1;

  const int nwg = P.tiles_m * P.tiles_n;
  const int id = xcd_remap(bx, nwg);
  const int tm = id % P.tiles_m, tn = id / P.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  const int step0 = bz * P.nk_per_split;
  int nsteps = P.nk - step0;
  if (nsteps > P.nk_per_split) nsteps = P.nk_per_split;

  int PN = P.N;                 // this problem's output width / row pitch / slab placement
  long Pcrs = P.c_rs, Pslab_stride = P.slab_stride;
  long Cbatch = P.nbatch ? (long)by * P.slab_batch_stride : 0;
  if (P.nbatch && P.Nb[0]) {
    PN = P.Nb[by];
    Pcrs = PN;
    Pslab_stride = (long)P.M * PN;
    Cbatch = P.slab_off[by];
    if (n0 >= PN) return;       // tile column beyond this (narrower) problem: whole workgroup
  }
  typename LoaderOf<BM, BKT, ASRC, DT != 0>::type LA;
  typename LoaderOf<BN, BKT, BSRC, DT != 0>::type LB;
  const float* Abase = P.nbatch ? P.Ab[by] : P.A;
  const float* Bbase = P.nbatch ? P.Bb[by] : P.B;
  LA.init(P, Abase, P.a_rs, P.a_bs, m0, P.M, tid);
  LB.init(P, Bbase, P.b_rs, P.b_bs, n0, PN, tid);

  f32x16 acc[IM][JN];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fa = (l >> 5) * LDA + wm * WM + (l & 31);
  const int fb = (l >> 5) * LDB + wn * WN + (l & 31);
  // one K-step of MFMAs on LDS stage `cur`; fragments of k-step kk+1 are read while the
  // MFMAs of k-step kk issue
  auto compute = [&](int cur) {
    const float* as = As + cur * BK * LDA + fa;
    const float* bs = Bs + cur * BK * LDB + fb;
    float a[2][IM], b[2][JN];
#pragma unroll
    for (int i = 0; i < IM; ++i) a[0][i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < JN; ++j) b[0][j] = bs[j * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < IM; ++i) a[nx][i] = as[(kk + 1) * 2 * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < JN; ++j) b[nx][j] = bs[(kk + 1) * 2 * LDB + j * 32];
        // bulk tiles: pin the issue order -- hipcc otherwise sinks these reads below the
        // MFMAs of k-step kk and waits lgkmcnt(0) in front of every MFMA group
        if (BM >= 128 && PIN) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
    }
  };

  // bf16 mode: per 16-deep k-step, fragment = planes 4s+2h and 4s+2h+1 of the lane's row
  auto compute_bf16 = [&](int cur) {
    const int r = l & 31, h = l >> 5;
    const uint2* as = Ab16 + cur * NPL * PLA;
    const uint2* bs = Bb16 + cur * NPL * PLB;
    int ia[IM], jb[JN];   // in-plane index of this lane's row of fragment i / j
#pragma unroll
    for (int i = 0; i < IM; ++i) ia[i] = LAT::Img::at(wm * WM + i * 32 + r);
#pragma unroll
    for (int j = 0; j < JN; ++j) jb[j] = LBT::Img::at(wn * WN + j * 32 + r);
    if constexpr (DT == 2) {   // split operands: six products per 16-deep k-step
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        bf16x8 a[IM][3], b[JN][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int i = 0; i < IM; ++i) {
            const uint2 lo = as[p * PLA + (4 * s + 2 * h) * PPA + ia[i]];
            const uint2 hi = as[p * PLA + (4 * s + 2 * h + 1) * PPA + ia[i]];
            a[i][p] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          }
#pragma unroll
          for (int j = 0; j < JN; ++j) {
            const uint2 lo = bs[p * PLB + (4 * s + 2 * h) * PPB + jb[j]];
            const uint2 hi = bs[p * PLB + (4 * s + 2 * h + 1) * PPB + jb[j]];
            b[j][p] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          }
        }
        // smallest terms first, so the large hi*hi term is added to an already formed correction
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0};
        constexpr int TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < JN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], b[j][TB[t]],
                                                                  acc[i][j], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a[IM], b[JN];
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        const uint2 lo = as[(4 * s + 2 * h) * PPA + ia[i]];
        const uint2 hi = as[(4 * s + 2 * h + 1) * PPA + ia[i]];
        a[i] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const uint2 lo = bs[(4 * s + 2 * h) * PPB + jb[j]];
        const uint2 hi = bs[(4 * s + 2 * h + 1) * PPB + jb[j]];
        b[j] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  typename LAT::Regs ra0;
  typename LBT::Regs rb0;
  // Interior tiles (every row/column valid, reduction a whole number of K-steps: all
  // the bulk shapes) take unpredicated loads: the per-load exec-mask branches and
  // address recomputation of the general path cost ~15% of the K-loop's issue slots.
  const bool kfull = (ASRC == SRC_SC) || (ASRC == SRC_SC_DTANH) ? (P.S % BKT == 0) : (P.K % BKT == 0);
  const bool interior = kfull && m0 + BM <= P.M && n0 + BN <= PN;
  auto mainloop = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    if (nsteps > 0) {
      LA.template load<FAST>(step0, ra0);
      LB.template load<FAST>(step0, rb0);
      if constexpr (DT != 0) {
        LA.template store_bf16<NPL>(Ab16, PLA, tid, ra0);
        LB.template store_bf16<NPL>(Bb16, PLB, tid, rb0);
      } else {
        if (P.round16) { round_regs(ra0); round_regs(rb0); }
        LA.store(As, tid, ra0);
        LB.store(Bs, tid, rb0);
      }
    }
    __syncthreads();
    for (int it = 0; it < nsteps; ++it) {
      const int cur = NST == 2 ? (it & 1) : 0;
      const bool more = it + 1 < nsteps;
      if (more && !(P.dbg & 1)) {
        LA.template load<FAST>(step0 + it + 1, ra0);
        LB.template load<FAST>(step0 + it + 1, rb0);
      }
      if constexpr (DT != 0) compute_bf16(cur); else compute(cur);
      if (NST == 1) __syncthreads();   // everyone is done reading the single stage
      if (more && !(P.dbg & 1)) {
        if constexpr (DT != 0) {
          LA.template store_bf16<NPL>(Ab16 + (NST == 2 ? (cur ^ 1) : 0) * NPL * PLA, PLA, tid, ra0);
          LB.template store_bf16<NPL>(Bb16 + (NST == 2 ? (cur ^ 1) : 0) * NPL * PLB, PLB, tid, rb0);
        } else {
          if (P.round16) { round_regs(ra0); round_regs(rb0); }
          LA.store(As + (NST == 2 ? (cur ^ 1) * BK * LDA : 0), tid, ra0);
          LB.store(Bs + (NST == 2 ? (cur ^ 1) * BK * LDB : 0), tid, rb0);
        }
      }
      if (!(P.dbg & 2)) __syncthreads();
    }
  };
  if (interior)
    mainloop(std::true_type{});
  else
    mainloop(std::false_type{});

  // i_embed bias gradient for free: the staged A operand of the i_embed weight gradient IS
  // dZ = dI (1 - I^2); the workgroups of the first tile column hand back its row sums over
  // their K range (fixed-order lane tree, then a split reduction by the launcher).
  if constexpr (ASRC == SRC_SC_DTANH) {
    if (P.rs_out && tn == 0) {
      constexpr int NIA = LAT::NI;
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        float v = LA.rsum[i];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        const int m = m0 + tid / 8 + i * 32;
        if ((tid & 7) == 0 && m < P.M) P.rs_out[(long)bz * P.M + m] = v;
      }
    }
  }

  // Linear bias gradient for free: column sums of the staged dY operand (first tile column
  // only), combined over the RPP threads of a column group through the idle staging LDS in
  // fixed order, one partial row per K split.
  if constexpr (ASRC == SRC_RC_SUM) {
    if (P.rs_out && tn == 0) {
      float4* red = reinterpret_cast<float4*>(smem);        // [RPP][BM/4]
      red[LA.kr * (BM / 4) + (LA.c4 >> 2)] = LA.csum;
      __syncthreads();
      if (tid < BM / 4) {
        float4 v = red[tid];
#pragma unroll
        for (int q = 1; q < LAT::RPP; ++q) {
          const float4 w4 = red[q * (BM / 4) + tid];
          v.x += w4.x; v.y += w4.y; v.z += w4.z; v.w += w4.w;
        }
        const int m = m0 + tid * 4;
        float* o = P.rs_out + (long)bz * P.M + m;
        if (m + 3 < P.M) {
          o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        } else {
          if (m < P.M) o[0] = v.x;
          if (m + 1 < P.M) o[1] = v.y;
          if (m + 2 < P.M) o[2] = v.z;
        }
      }
    }
  }

  // ------------------------------------------------------------ epilogue
  // accumulator element r of block (i,j): row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31
  const int rloc = wm * WM + 4 * (l >> 5);   // row of (i=0, r=0) inside the tile
  const int cloc = wn * WN + (l & 31);       // column of j=0 inside the tile

  if (EPI == EPI_LIN || EPI == EPI_SLAB) {
    float* C = P.C + (EPI == EPI_SLAB ? (long)bz * Pslab_stride + Cbatch : 0);
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int n = n0 + cloc + j * 32;
      if (n >= PN) continue;
      float bsum = 0.f;
      if (EPI == EPI_LIN) {
        if (P.bias) bsum += P.bias[n];
        if (P.bias2) bsum += P.bias2[n];
      }
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + rloc + i * 32 + (r & 3) + 8 * (r >> 2);
          if (m >= P.M) continue;
          float v = acc[i][j][r];
          const long ci = (long)m * Pcrs + n;
          if (EPI == EPI_LIN) {
            v = v * P.alpha + bsum;
            if (P.addend) v += P.addend[(long)m * P.add_rs + n];
            if (P.accumulate) v += C[ci];
            if (P.act == 1) v = tanh_fast(v);
            if (P.ymul) {
              const float y = P.ymul[(long)m * P.y_rs + n];
              v *= (1.f - y * y);
            }
            if (P.emask)
              v = mask_bit(P.emask, P.emask_e0 + (size_t)m * P.N + n) ? v * P.emscale : 0.f;
          }
          C[ci] = v;
        }
    }
  } else {
    // EPI_CONV on maps with S % 4 == 0: the accumulator tile goes through the (idle) staging LDS and
    // leaves as whole rows -- a thread stores float4s of four consecutive positions, 32 lanes one
    // 512-byte run of a row -- instead of 128-byte pieces at the 4 S-byte row pitch written by
    // different waves at different times (measured round 1: 1.37x the algorithmic bytes reached HBM,
    // all of it partial-line writes).  bf16 modes stage half the tile at a time (their LDS is smaller).
    if constexpr (EPI == EPI_CONV && BM == 128 && BN == 128) {
      if (P.S % 4 == 0) {
        constexpr int TP = BN + 4;                               // staged row pitch (floats)
        constexpr int RP = kStage / TP >= BM ? BM : 64;          // rows per pass
        static_assert(RP * TP <= kStage, "staging area too small for the epilogue tile");
        float* T = smem;
        const int c4 = (tid & 31) * 4;
        const int n = n0 + c4;
        const bool nok = n < P.N;                                // N = nB * S is a multiple of 4
        const int nb = (nok ? n : 0) / P.S, ns = (nok ? n : 0) - nb * P.S;
        float* crow = P.C + (long)nb * P.c_bs + ns;
        __syncthreads();
#pragma unroll 1
        for (int p0 = 0; p0 < BM; p0 += RP) {
#pragma unroll
          for (int i = 0; i < IM; ++i) {
            const int rb = wm * WM + i * 32;                     // first tile row of this 32x32 block
            if (rb >= p0 && rb < p0 + RP) {
#pragma unroll
              for (int j = 0; j < JN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                  T[(rloc + i * 32 + (r & 3) + 8 * (r >> 2) - p0) * TP + cloc + j * 32] = acc[i][j][r];
            }
          }
          __syncthreads();
          for (int row = tid >> 5; row < RP; row += 8) {
            const int m = m0 + p0 + row;
            if (m < P.M && nok) {
              float4 v = *reinterpret_cast<const float4*>(T + row * TP + c4);
              const float bv = P.bias ? P.bias[m] : 0.f;
              v.x += bv; v.y += bv; v.z += bv; v.w += bv;
              if (P.act) { v.x = tanh_fast(v.x); v.y = tanh_fast(v.y); v.z = tanh_fast(v.z); v.w = tanh_fast(v.w); }
              *reinterpret_cast<float4*>(crow + (long)m * P.S) = v;
            }
          }
          __syncthreads();
        }
        return;
      }
    }
    // Flattened-column epilogues: n -> (sample b, position s).  The per-row bias and
    // the per-(sample,row) vector dj of the few samples a tile's columns span are
    // staged in LDS once, instead of one dependent global load per element.
    float* rowv = smem;                 // [BM] bias
    float* uv = smem + BM;              // [nsamp][BM] dj
    const int b0 = n0 / P.S;
    int nlast = n0 + BN - 1;
    if (nlast > P.N - 1) nlast = P.N - 1;
    const int nsamp = nlast / P.S - b0 + 1;
    const bool staged = nsamp * BM <= kUCap;
    if (tid < BM) {
      const int m = m0 + tid;
      rowv[tid] = (EPI == EPI_CONV && m < P.M && P.bias) ? P.bias[m] : 0.f;
    }
    if (EPI == EPI_OUTER && staged)
      for (int e = tid; e < nsamp * BM; e += 256) {
        const int sb = e / BM, r = e - sb * BM;
        uv[e] = (m0 + r < P.M) ? P.v1[(long)(b0 + sb) * P.M + m0 + r] : 0.f;
      }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int n = n0 + cloc + j * 32;
      const bool nok = n < P.N;
      const int nn = nok ? n : P.N - 1;
      const int b = nn / P.S, s = nn - b * P.S;
      const long cb = (long)b * P.c_bs + s;
      const float* uvb = uv + (b - b0) * BM;
      float an = 0.f;
      if (EPI == EPI_OUTER) an = P.v2[nn];
#pragma unroll
      for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = rloc + i * 32 + (r & 3) + 8 * (r >> 2);
          const int m = m0 + rl;
          const bool ok = nok && m < P.M;
          const long ci = cb + (long)(m < P.M ? m : P.M - 1) * P.S;
          const float v = acc[i][j][r];
          if (EPI == EPI_CONV) {
            const float t = v + rowv[rl];
            if (ok) P.C[ci] = P.act ? tanh_fast(t) : t;
          } else {
            const float pvm = staged ? uvb[rl] : (m < P.M ? P.v1[(long)b * P.M + m] : 0.f);
            if (ok) P.C[ci] = v + pvm * an;
          }
        }
    }
  }
}

#ifndef RAU_B16_FWD_WGS
#define RAU_B16_FWD_WGS 4   // workgroups per CU of the forward bf16 kernels with pre-converted weights
#endif
template <int BM, int BN, int BKT, int ASRC, int BSRC, int EPI, int DT = 0>
__global__ __launch_bounds__(256, DT == 1 ? (ASRC == SRC_RC_B16 ? RAU_B16_FWD_WGS : 3) : 2) void gemm_kernel(const GemmParams P) {
  gemm_tile<BM, BN, BKT, ASRC, BSRC, EPI, DT>(P, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Split-K launch whose workgroups are laid out so that ALL TILES OF ONE K-SPLIT SHARE AN XCD.
// The conv weight gradients have few output tiles (8 or 16) and many K splits (whole samples per
// split); every tile of a split streams the same samples' operand rows.  With a (tiles, 1, splits)
// grid the linear workgroup id is x + tiles * z, and workgroups are dealt to the 8 XCDs round-robin
// by that id: tiles a multiple of 8 puts tile x of EVERY split on XCD x % 8, i.e. the 16 tiles that
// share operand data sit on 8 different L2s and each fetches it again (measured: 3.9x / 5.7x the
// algorithmic bytes leave L2 for conv_att_wgrad / conv_embed_wgrad).  Here the grid is 1-D:
// workgroup L -> XCD r = L % 8, q = L / 8, split = r + 8 * (q / tiles), tile = q % tiles, so a
// split's tiles are consecutive on one XCD (placement is a speed matter only: any mapping is
// correct).  Grid = 8 * ceil(splits / 8) * tiles; surplus workgroups exit.
template <int BM, int BN, int BKT, int ASRC, int BSRC, int EPI, int DT = 0>
__global__ __launch_bounds__(256, DT == 1 ? 3 : 2) void gemm_split_xcd_kernel(const GemmParams P,
                                                                               const int splits) {
  const int tiles = P.tiles_m * P.tiles_n;
  const int L = blockIdx.x, r = L & 7, q = L >> 3;
  const int split = r + 8 * (q / tiles), tile = q % tiles;
  if (split >= splits) return;
  gemm_tile<BM, BN, BKT, ASRC, BSRC, EPI, DT>(P, tile, 0, split);
}

// Grouped launch: up to kGroupMax independent split-K problems C_p = A_p^T B_p (the Linear weight
// gradients of one parameter group) in ONE grid.  Workgroup g belongs to problem p with
// wg0[p] <= g < wg0[p+1]; inside it, tile = (g - wg0[p]) % tiles, split = (g - wg0[p]) / tiles.
constexpr int kGroupMax = 13;
struct GroupProb {
  const float* A; const float* B; float* slab; float* rs_out;
  long lda, ldb;
  int M, N, tiles_m, tiles_n;
};
struct GroupParams {
  int np, K, nk, nk_per_split;
  int wg0[kGroupMax + 1];
  GroupProb p[kGroupMax];
};
template <int BM, int BN, int BKT, int ASRC, int BSRC, int DT = 0>
__global__ __launch_bounds__(256, 2) void gemm_group_kernel(const GroupParams G) {
  int pi = 0;
#pragma unroll 1
  while (pi + 1 < G.np && (int)blockIdx.x >= G.wg0[pi + 1]) ++pi;
  const GroupProb& q = G.p[pi];
  const int local = (int)blockIdx.x - G.wg0[pi];
  const int tiles = q.tiles_m * q.tiles_n;
  GemmParams P{};
  P.M = q.M; P.N = q.N; P.K = G.K;
  P.nk = G.nk; P.nk_per_split = G.nk_per_split;
  P.tiles_m = q.tiles_m; P.tiles_n = q.tiles_n;
  P.A = q.A; P.a_rs = q.lda;
  P.B = q.B; P.b_rs = q.ldb;
  P.C = q.slab; P.c_rs = q.N; P.slab_stride = (long)q.M * q.N;
  P.rs_out = q.rs_out;
  P.alpha = 1.f;
  gemm_tile<BM, BN, BKT, ASRC, BSRC, EPI_SLAB, DT>(P, local % tiles, 0, local / tiles);
}

// ------------------------------------------------------------ host launch
template <int BM, int BN, int BKT, int ASRC, int BSRC, int EPI, int DT = 0>
inline hipError_t launch_gemm(hipStream_t st, GemmParams P, int splits, int dyn_lds = 0) {
  P.tiles_m = (P.M + BM - 1) / BM;
  P.tiles_n = (P.N + BN - 1) / BN;
  if (splits < 1) splits = 1;
  if (splits > P.nk) splits = P.nk > 0 ? P.nk : 1;
  P.nk_per_split = (P.nk + splits - 1) / splits;
  splits = P.nk_per_split > 0 ? (P.nk + P.nk_per_split - 1) / P.nk_per_split : 1;
  dim3 grid(P.tiles_m * P.tiles_n, P.nbatch ? P.nbatch : 1, splits);
  // dyn_lds: unused dynamic LDS that only lowers how many of these workgroups share a CU
  hipLaunchKernelGGL((gemm_kernel<BM, BN, BKT, ASRC, BSRC, EPI, DT>), grid, dim3(256), dyn_lds, st, P);
  return hipGetLastError();
}

}  // namespace rau
