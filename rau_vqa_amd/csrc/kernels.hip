// kernels.hip -- pointwise, reduction, softmax and gather kernels of the RAU path.
// All are HBM/latency-bound byte movers: coalesced (float4 where the layout
// allows), one wave per row for row reductions (64-lane shuffle trees), and
// two-stage deterministic reductions instead of float atomics.
#include "common.h"
#include <cstdlib>

#include "kernels.h"
#include "philox.h"

namespace rau {

static inline int grid_for(size_t n, int block = 256, int cap = 256 * 8) {
  size_t g = (n + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------- dropout masks
__global__ void k_fill_masks(uint64_t seed, uint32_t site, uint32_t step, uint32_t thr,
                             size_t nwords, uint32_t* __restrict__ bits,
                             const uint64_t* __restrict__ key) {
  RAU_CHAIN_PRIO();
  if (key) {   // (seed, step) live in device memory: a captured graph replays with fresh keys
    seed = key[0];
    step = (uint32_t)key[1];
  }
  for (size_t w = blockIdx.x * (size_t)blockDim.x + threadIdx.x; w < nwords;
       w += (size_t)gridDim.x * blockDim.x) {
    const uint32_t lo = philox_keep16(seed, site, step, 2 * w, thr);
    const uint32_t hi = philox_keep16(seed, site, step, 2 * w + 1, thr);
    bits[w] = lo | (hi << 16);
  }
}
hipError_t fill_masks(hipStream_t st, uint64_t seed, uint32_t site, uint32_t step, float p,
                      size_t n, uint32_t* bits, const uint64_t* key_dev) {
  const size_t nwords = (n + 31) / 32;
  const uint32_t thr = (uint32_t)lroundf(p * 256.0f);
  hipLaunchKernelGGL(k_fill_masks, dim3(grid_for(nwords)), dim3(256), 0, st, seed, site, step,
                     thr, nwords, bits, key_dev);
  return hipGetLastError();
}

__global__ void k_uniform_fill(uint64_t seed, uint32_t stream, size_t n, float lo, float hi,
                               float* __restrict__ x) {
  RAU_CHAIN_PRIO();
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q * 4 < n;
       q += (size_t)gridDim.x * blockDim.x) {
    const Philox4 o = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), stream, 0x55AAu,
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < n) x[q * 4 + j] = lo + (hi - lo) * ((o.v[j] >> 8) * (1.0f / 16777216.0f));
  }
}
hipError_t uniform_fill(hipStream_t st, uint64_t seed, uint32_t stream, size_t n, float lo,
                        float hi, float* x) {
  hipLaunchKernelGGL(k_uniform_fill, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, seed, stream,
                     n, lo, hi, x);
  return hipGetLastError();
}

// ------------------------------------------------------------ word embedding
// we[row, e] = tanh(drop(E[token[row]-1, e]))      reference SS:203-206
// Token ids of the step-level path are range-checked on the host (rau_set_batch); the module-level calls
// take DEVICE id tensors, which nobody has looked at: ids are clamped into [1, V] so that a bad id
// reads / updates a wrong row instead of faulting the GPU (the reference's LookupTable would raise).
__global__ void k_embed_fwd(int rows, int E, int V, const float* __restrict__ emb,
                            const int32_t* __restrict__ tokens,
                            const uint32_t* __restrict__ mask, size_t mask_e0, float mscale,
                            float* __restrict__ we) {
  RAU_CHAIN_PRIO();
  const size_t n = (size_t)rows * E;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / E), e = (int)(i - (size_t)row * E);
    const int tk = min(max(tokens[row], 1), V);
    float v = emb[(size_t)(tk - 1) * E + e];
    if (mask) v = mask_bit(mask, mask_e0 + i) ? v * mscale : 0.f;
    we[i] = tanh_fast(v);
  }
}
hipError_t embed_fwd(hipStream_t st, int rows, int E, int V, const float* emb, const int32_t* tokens,
                     const uint32_t* mask, float mscale, float* we, size_t mask_e0) {
  hipLaunchKernelGGL(k_embed_fwd, dim3(grid_for((size_t)rows * E)), dim3(256), 0, st, rows, E, V,
                     emb, tokens, mask, mask_e0, mscale, we);
  return hipGetLastError();
}

// LookupTable gradient of ONE token row set (module-level word_embed:backward, SS:593):
// thread e walks the rows in order, so repeated tokens accumulate deterministically.
__global__ void k_embed_bwd_rows(int rows, int E, int V, const int32_t* __restrict__ tokens,
                                 const float* __restrict__ dwe, const float* __restrict__ we,
                                 const uint32_t* __restrict__ mask, size_t mask_e0, float mscale,
                                 float* __restrict__ gE) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  for (int r = 0; r < rows; ++r) {
    const size_t i = (size_t)r * E + e;
    const float y = we[i];
    float d = dwe[i] * (1.f - y * y);
    if (mask) d = mask_bit(mask, mask_e0 + i) ? d * mscale : 0.f;
    gE[(size_t)(min(max(tokens[r], 1), V) - 1) * E + e] += d;
  }
}
hipError_t embed_bwd_rows(hipStream_t st, int rows, int E, int V, const int32_t* tokens, const float* dwe,
                          const float* we, const uint32_t* mask, size_t mask_e0, float mscale,
                          float* gE) {
  hipLaunchKernelGGL(k_embed_bwd_rows, dim3((E + 63) / 64), dim3(64), 0, st, rows, E, V, tokens, dwe,
                     we, mask, mask_e0, mscale, gE);
  return hipGetLastError();
}

// LookupTable gradient as a gather-sum: block u owns one distinct token and
// adds its positions' gradients in a fixed order (deterministic, no atomics).
__global__ void k_embed_bwd(int E, const int32_t* __restrict__ utok,
                            const int32_t* __restrict__ ustart,
                            const int32_t* __restrict__ upos, const float* __restrict__ dwe,
                            const float* __restrict__ we, const uint32_t* __restrict__ mask,
                            float mscale, float* __restrict__ gE) {
  RAU_CHAIN_PRIO();
  const int u = blockIdx.x;
  const int p0 = ustart[u], p1 = ustart[u + 1];
  if (p1 <= p0) return;   // padding entry (the grid may cover the maximum token count)
  const int tok = utok[u];
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float acc = 0.f;
    for (int p = p0; p < p1; ++p) {
      const size_t i = (size_t)upos[p] * E + e;
      const float y = we[i];
      float d = dwe[i] * (1.f - y * y);
      if (mask) d = mask_bit(mask, i) ? d * mscale : 0.f;
      acc += d;
    }
    gE[(size_t)(tok - 1) * E + e] += acc;
  }
}
hipError_t embed_bwd(hipStream_t st, int nuniq, int E, const int32_t* utok, const int32_t* ustart,
                     const int32_t* upos, const float* dwe, const float* we,
                     const uint32_t* mask, float mscale, float* gE) {
  if (nuniq <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_embed_bwd, dim3(nuniq), dim3(64), 0, st, E, utok, ustart, upos, dwe, we,
                     mask, mscale, gE);
  return hipGetLastError();
}

// --------------------------------------------------------------- LSTM cells
// gate slots in the 4R-wide pre-activation row
template <int ORDER> struct GateSlots;
template <> struct GateSlots<GATES_ATT> { enum { I = 0, G = 1, F = 2, O = 3 }; };
template <> struct GateSlots<GATES_DEEP> { enum { I = 0, F = 1, O = 2, G = 3 }; };

template <int ORDER>
__global__ void k_lstm_fwd(int nB, int R, float* __restrict__ g4,
                           const float* __restrict__ c_prev, long cp_rs, float* __restrict__ c,
                           long c_rs, float* __restrict__ h, long h_rs,
                           float* __restrict__ tanhc, float* __restrict__ drop_out,
                           const uint32_t* __restrict__ mask, size_t mask_e0, float mscale,
                           const float* __restrict__ slab, int nsplit) {
  RAU_CHAIN_PRIO();
  using GS = GateSlots<ORDER>;
  const size_t n = (size_t)nB * R;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / R), r = (int)(i - (size_t)b * R);
    float* g = g4 + (size_t)b * 4 * R;
    float pi = g[GS::I * R + r], pf = g[GS::F * R + r], po = g[GS::O * R + r],
          pg = g[GS::G * R + r];
    for (int s = 0; s < nsplit; ++s) {  // split-K partials of h_prev W^T, fixed order
      const float* sl = slab + ((size_t)s * nB + b) * 4 * R;
      pi += sl[GS::I * R + r];
      pf += sl[GS::F * R + r];
      po += sl[GS::O * R + r];
      pg += sl[GS::G * R + r];
    }
    const float gi = sigmoidf_(pi);
    const float gf = sigmoidf_(pf);
    const float go = sigmoidf_(po);
    const float gg = tanh_fast(pg);
    g[GS::I * R + r] = gi;
    g[GS::F * R + r] = gf;
    g[GS::O * R + r] = go;
    g[GS::G * R + r] = gg;
    const float cn = gf * c_prev[(size_t)b * cp_rs + r] + gi * gg;
    const float tc = tanh_fast(cn);
    const float hn = go * tc;
    c[(size_t)b * c_rs + r] = cn;
    h[(size_t)b * h_rs + r] = hn;
    tanhc[i] = tc;
    if (drop_out) {
      float v = hn;
      if (mask) v = mask_bit(mask, mask_e0 + i) ? hn * mscale : 0.f;
      drop_out[i] = v;
    }
  }
}
hipError_t lstm_fwd(hipStream_t st, int order, int nB, int R, float* g4, const float* c_prev,
                    long cp_rs, float* c, long c_rs, float* h, long h_rs, float* tanhc,
                    float* drop_out, const uint32_t* mask, size_t mask_e0, float mscale,
                    const float* slab, int nsplit) {
  if (!split_span_ok(slab, nsplit, (size_t)nB * 4 * R)) return kSplitStateError;
  const dim3 g(grid_for((size_t)nB * R)), b(256);
  if (order == GATES_ATT)
    hipLaunchKernelGGL(k_lstm_fwd<GATES_ATT>, g, b, 0, st, nB, R, g4, c_prev, cp_rs, c, c_rs, h,
                       h_rs, tanhc, drop_out, mask, mask_e0, mscale, slab, nsplit);
  else
    hipLaunchKernelGGL(k_lstm_fwd<GATES_DEEP>, g, b, 0, st, nB, R, g4, c_prev, cp_rs, c, c_rs, h,
                       h_rs, tanhc, drop_out, mask, mask_e0, mscale, slab, nsplit);
  return hipGetLastError();
}

template <int ORDER>
__global__ void k_lstm_bwd(int nB, int R, const float* __restrict__ gates,
                           const float* __restrict__ c_prev, long cp_rs,
                           const float* __restrict__ tanhc, const float* __restrict__ dh,
                           long dh_rs, const float* __restrict__ dh2,
                           const float* __restrict__ dc_next, float* __restrict__ dsum,
                           float* __restrict__ dc_prev, const int32_t* __restrict__ lens, int t,
                           const float* __restrict__ dq_c, const float* __restrict__ dq_h,
                           long dq_rs, const float* __restrict__ slab, int nsplit) {
  RAU_CHAIN_PRIO();
  using GS = GateSlots<ORDER>;
  const size_t n = (size_t)nB * R;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / R), r = (int)(i - (size_t)b * R);
    float dhv, dcv;
    if (lens && lens[b] == t) {  // rows REPLACED by dq, reference SS:584-591
      dhv = dq_h[(size_t)b * dq_rs + r];
      dcv = dq_c[(size_t)b * dq_rs + r];
    } else {
      dhv = dh ? dh[(size_t)b * dh_rs + r] : 0.f;
      for (int s = 0; s < nsplit; ++s) dhv += slab[((size_t)s * nB + b) * R + r];
      dcv = dc_next ? dc_next[i] : 0.f;
    }
    if (dh2) dhv += dh2[i];
    const float* g = gates + (size_t)b * 4 * R;
    const float gi = g[GS::I * R + r], gf = g[GS::F * R + r], go = g[GS::O * R + r],
                gg = g[GS::G * R + r];
    const float tc = tanhc[i];
    const float d_o = dhv * tc;
    const float dc = dcv + dhv * go * (1.f - tc * tc);
    float* ds = dsum + (size_t)b * 4 * R;
    ds[GS::I * R + r] = dc * gg * gi * (1.f - gi);
    ds[GS::F * R + r] = dc * c_prev[(size_t)b * cp_rs + r] * gf * (1.f - gf);
    ds[GS::O * R + r] = d_o * go * (1.f - go);
    ds[GS::G * R + r] = dc * gi * (1.f - gg * gg);
    dc_prev[i] = dc * gf;
  }
}
hipError_t lstm_bwd(hipStream_t st, int order, int nB, int R, const float* gates,
                    const float* c_prev, long cp_rs, const float* tanhc, const float* dh,
                    long dh_rs, const float* dh2, const float* dc_next, float* dsum,
                    float* dc_prev, const int32_t* lens, int t, const float* dq_c,
                    const float* dq_h, long dq_rs, const float* slab, int nsplit) {
  if (!split_span_ok(slab, nsplit, (size_t)nB * R)) return kSplitStateError;
  const dim3 g(grid_for((size_t)nB * R)), b(256);
  if (order == GATES_ATT)
    hipLaunchKernelGGL(k_lstm_bwd<GATES_ATT>, g, b, 0, st, nB, R, gates, c_prev, cp_rs, tanhc,
                       dh, dh_rs, dh2, dc_next, dsum, dc_prev, lens, t, dq_c, dq_h, dq_rs, slab, nsplit);
  else
    hipLaunchKernelGGL(k_lstm_bwd<GATES_DEEP>, g, b, 0, st, nB, R, gates, c_prev, cp_rs, tanhc,
                       dh, dh_rs, dh2, dc_next, dsum, dc_prev, lens, t, dq_c, dq_h, dq_rs, slab, nsplit);
  return hipGetLastError();
}

// Multi-cell variants (blockIdx.y = cell): same arithmetic as k_lstm_fwd / k_lstm_bwd.
template <int ORDER>
__global__ void k_lstm_fwd_multi(int nB, int R, LstmFwdCells cs) {
  RAU_CHAIN_PRIO();
  using GS = GateSlots<ORDER>;
  const LstmFwdCell& C = cs.c[blockIdx.y];
  const size_t n = (size_t)nB * R;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / R), r = (int)(i - (size_t)b * R);
    float* g = C.g4 + (size_t)b * 4 * R;
    float pi, pf, po, pg;
    if (C.has_input) {
      pi = g[GS::I * R + r]; pf = g[GS::F * R + r]; po = g[GS::O * R + r]; pg = g[GS::G * R + r];
    } else {
      pi = C.b1[GS::I * R + r] + C.b2[GS::I * R + r];
      pf = C.b1[GS::F * R + r] + C.b2[GS::F * R + r];
      po = C.b1[GS::O * R + r] + C.b2[GS::O * R + r];
      pg = C.b1[GS::G * R + r] + C.b2[GS::G * R + r];
    }
    for (int s = 0; s < C.nsplit; ++s) {
      const float* sl = C.slab + ((size_t)s * nB + b) * 4 * R;
      pi += sl[GS::I * R + r]; pf += sl[GS::F * R + r];
      po += sl[GS::O * R + r]; pg += sl[GS::G * R + r];
    }
    const float gi = sigmoidf_(pi), gf = sigmoidf_(pf), go = sigmoidf_(po), gg = tanh_fast(pg);
    g[GS::I * R + r] = gi; g[GS::F * R + r] = gf; g[GS::O * R + r] = go; g[GS::G * R + r] = gg;
    const float cn = gf * C.c_prev[(size_t)b * C.cp_rs + r] + gi * gg;
    const float tc = tanh_fast(cn);
    const float hn = go * tc;
    C.c[(size_t)b * C.c_rs + r] = cn;
    C.h[(size_t)b * C.h_rs + r] = hn;
    C.tanhc[i] = tc;
    if (C.drop_out) {
      float v = hn;
      if (C.mask) v = mask_bit(C.mask, C.mask_e0 + i) ? hn * C.mscale : 0.f;
      C.drop_out[i] = v;
    }
  }
}
hipError_t lstm_fwd_multi(hipStream_t st, int order, int nB, int R, const LstmFwdCells& cells) {
  if (cells.n < 1) return hipSuccess;
  if (cells.n > 2) return hipErrorInvalidValue;
  for (int i = 0; i < cells.n; ++i)
    if (!split_span_ok(cells.c[i].slab, cells.c[i].nsplit, (size_t)nB * 4 * R)) return kSplitStateError;
  const dim3 g(grid_for((size_t)nB * R, 256, 512), cells.n), b(256);
  if (order == GATES_ATT)
    hipLaunchKernelGGL(k_lstm_fwd_multi<GATES_ATT>, g, b, 0, st, nB, R, cells);
  else
    hipLaunchKernelGGL(k_lstm_fwd_multi<GATES_DEEP>, g, b, 0, st, nB, R, cells);
  return hipGetLastError();
}

template <int ORDER>
__global__ void k_lstm_bwd_multi(int nB, int R, LstmBwdCells cs) {
  RAU_CHAIN_PRIO();
  using GS = GateSlots<ORDER>;
  const LstmBwdCell& C = cs.c[blockIdx.y];
  const size_t n = (size_t)nB * R;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / R), r = (int)(i - (size_t)b * R);
    float dhv, dcv;
    if (cs.lens[b] == C.t) {  // rows REPLACED by dq, reference SS:584-591
      dhv = C.dq_h[(size_t)b * cs.dq_rs + r];
      dcv = C.dq_c[(size_t)b * cs.dq_rs + r];
    } else {
      dhv = 0.f;
      for (int s = 0; s < C.nA; ++s) dhv += C.slabA[((size_t)s * nB + b) * R + r];
      dcv = C.dc_next ? C.dc_next[i] : 0.f;
    }
    if (C.nBp > 0) {  // gradient arriving through the inter-layer dropout (DeepLSTM.lua:39)
      float e = 0.f;
      for (int s = 0; s < C.nBp; ++s) e += C.slabB[((size_t)s * nB + b) * R + r];
      if (C.maskB) e = mask_bit(C.maskB, C.maskB_e0 + i) ? e * C.mscaleB : 0.f;
      dhv += e;
    }
    const float* g = C.gates + (size_t)b * 4 * R;
    const float gi = g[GS::I * R + r], gf = g[GS::F * R + r], go = g[GS::O * R + r],
                gg = g[GS::G * R + r];
    const float tc = C.tanhc[i];
    const float d_o = dhv * tc;
    const float dc = dcv + dhv * go * (1.f - tc * tc);
    float* ds = C.dsum + (size_t)b * 4 * R;
    ds[GS::I * R + r] = dc * gg * gi * (1.f - gi);
    ds[GS::F * R + r] = dc * C.c_prev[(size_t)b * C.cp_rs + r] * gf * (1.f - gf);
    ds[GS::O * R + r] = d_o * go * (1.f - go);
    ds[GS::G * R + r] = dc * gi * (1.f - gg * gg);
    C.dc_prev[i] = dc * gf;
  }
}
hipError_t lstm_bwd_multi(hipStream_t st, int order, int nB, int R, const LstmBwdCells& cells) {
  if (cells.n < 1) return hipSuccess;
  if (cells.n > 2) return hipErrorInvalidValue;
  for (int i = 0; i < cells.n; ++i)
    if (!split_span_ok(cells.c[i].slabA, cells.c[i].nA, (size_t)nB * R) ||
        !split_span_ok(cells.c[i].slabB, cells.c[i].nBp, (size_t)nB * R))
      return kSplitStateError;
  const dim3 g(grid_for((size_t)nB * R, 256, 512), cells.n), b(256);
  if (order == GATES_ATT)
    hipLaunchKernelGGL(k_lstm_bwd_multi<GATES_ATT>, g, b, 0, st, nB, R, cells);
  else
    hipLaunchKernelGGL(k_lstm_bwd_multi<GATES_DEEP>, g, b, 0, st, nB, R, cells);
  return hipGetLastError();
}

// ------------------------------------------- fused per-sample attention kernels
// One workgroup of kAttWaves waves per sample; everything a hop needs from the big
// per-sample tiles P/T [A,S] and I [M,S] in one pass over them.  16 waves per
// sample keep enough loads in flight to stream the ~0.8 MB a sample touches.
// Wave w owns rows w, w+NW, ...; lane l owns the float4 column groups l, l+64, ...
// Cross-wave sums go through LDS and are added in wave order (deterministic).
// waves per sample: 16 streams a sample fastest when the kernel has the GPU to itself, but a
// 16-wave workgroup needs 4 x its VGPRs per SIMD.  Round 3: the forward convs beside it are the
// wide tiles at ONE workgroup per CU (255 VGPRs on every SIMD), so what is left for the recurrence's
// kernels is 257 VGPRs per SIMD: the forward kernel's 4 x 102 do not fit and its workgroups waited
// for conv tiles (130 us each) to retire -- 117-265 us per launch in the step against 52 alone.
// With 8 waves (2 x 104) it starts at once: 10.02 -> 9.78 ms per step.  The backward kernel (48
// VGPRs) fits either way and stays at 16.  RAU_ATT_WAVES_FWD / RAU_ATT_WAVES_BWD (8 or 16) override.
// hint: the caller's choice where the environment does not override (0 = the default: 8 forward, 16
// backward).  RAU_BF16 mode asks for 8 backward waves: beside that mode's two 206-register dgrad tiles a
// SIMD has ~100 registers free, which two waves of 28 fit and four do not (step 8.28 -> 8.04 ms at
// D = 2048; f32 mode: no difference).
static int att_waves(bool bwd, int hint = 0) {
  static const int v[2] = {
      [] { const char* e = std::getenv("RAU_ATT_WAVES_FWD"); const int n = e ? std::atoi(e) : 0;
           return (n == 8 || n == 16) ? n : 0; }(),
      [] { const char* e = std::getenv("RAU_ATT_WAVES_BWD"); const int n = e ? std::atoi(e) : 0;
           return (n == 8 || n == 16) ? n : 0; }()};
  const int env = v[bwd ? 1 : 0];
  if (env) return env;
  if (hint == 8 || hint == 16) return hint;
  return bwd ? 16 : 8;
}

constexpr int kAttLoads = 8;   // independent row loads a wave issues before it consumes the first
constexpr int kAttLoadsT = 4;  // same, in the tanh-heavy loops (register budget: the kernels must fit
                               // beside two resident bulk-GEMM workgroups of 144-176 VGPRs)

template <int NW>
__device__ __forceinline__ float block_sum_ordered(const float* red, int S, int s) {
  float v = red[s];
#pragma unroll
  for (int w = 1; w < NW; ++w) v += red[(size_t)w * S + s];
  return v;
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_att_fwd_fused(
    int M, int A, int S, const float* __restrict__ P, const float* __restrict__ u,
    const float* __restrict__ ws, const float* __restrict__ bs, const float* __restrict__ zm,
    const float* __restrict__ I, const float* __restrict__ qf, float* __restrict__ T,
    float* __restrict__ a, float* __restrict__ jv, AttPartials ap) {
  // T == nullptr: tanh(P + u) is not kept (the backward recomputes it from P and ap.u_out)
  RAU_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* red = sm;                     // [NW][S]
  float* as = sm + NW * S;      // [S]
  float* sc = as + S;                  // [2*NW] block scalars
  const int b = blockIdx.x, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int S4 = S >> 2;
  const float* Pb = P + (size_t)b * A * S;
  float* Tb = T + (size_t)b * A * S;
  const float* ub = u + (size_t)b * A;
  // K-split partials of u / zm (+ bias) are finished here, once, into LDS
  float* us = sc + 2 * NW;             // [A]
  float* zs = us + A;                  // [S]
  if (ap.u_ns) {
    for (int k = tid; k < A; k += NW * 64) {
      float v = ap.u_bias[k];
      for (int sp = 0; sp < ap.u_ns; ++sp) v += u[((size_t)sp * gridDim.x + b) * A + k];
      us[k] = v;
    }
  }
  const int SL = ap.SL > 0 ? ap.SL : S;   // logical positions; [SL, S) are pad columns
  if (ap.z_ns) {
    for (int s = tid; s < SL; s += NW * 64) {
      float v = ap.z_bias[s];
      for (int sp = 0; sp < ap.z_ns; ++sp) v += zm[((size_t)sp * gridDim.x + b) * SL + s];
      zs[s] = v;
    }
  }
  if (ap.u_ns || ap.z_ns) __syncthreads();
  if (ap.u_out)   // the finished u row, for the backward's tanh(P + u)
    for (int k = tid; k < A; k += NW * 64) ap.u_out[(size_t)b * A + k] = ap.u_ns ? us[k] : ub[k];
  // ---- phase 1: T = tanh(P + u), e[s] = sum_k ws[k] T[k,s]
  // kAttLoadsT rows are loaded before the first one is used: next to the bulk GEMMs a dependent
  // global load takes microseconds, so the bytes in flight per wave set this kernel's speed
  for (int q0 = 0; q0 < S4; q0 += 64) {
    const int q = q0 + l;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < S4) {
      for (int k0 = w; k0 < A; k0 += NW * kAttLoadsT) {
        float4 p[kAttLoadsT];
#pragma unroll
        for (int i = 0; i < kAttLoadsT; ++i) {
          const int k = k0 + i * NW;
          p[i] = k < A ? reinterpret_cast<const float4*>(Pb + (size_t)k * S)[q]
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < kAttLoadsT; ++i) {
          const int k = k0 + i * NW;
          if (k < A) {
            const float uk = ap.u_ns ? us[k] : ub[k];
            const float wk = ws[k];
            float4 t;
            t.x = tanh_fast(p[i].x + uk); t.y = tanh_fast(p[i].y + uk);
            t.z = tanh_fast(p[i].z + uk); t.w = tanh_fast(p[i].w + uk);
            if (T) reinterpret_cast<float4*>(Tb + (size_t)k * S)[q] = t;
            acc.x += wk * t.x; acc.y += wk * t.y; acc.z += wk * t.z; acc.w += wk * t.w;
          }
        }
      }
      reinterpret_cast<float4*>(red + (size_t)w * S)[q] = acc;
    }
  }
  __syncthreads();
  // ---- phase 2: a = softmax(e + bs + zm)
  float mx = -INFINITY;
  for (int s = tid; s < S; s += (NW * 64)) {
    float z = -INFINITY;   // pad positions take no attention (and so no gradient)
    if (s < SL) {
      const float zmv = ap.z_ns ? zs[s] : zm[(size_t)b * S + s];
      z = block_sum_ordered<NW>(red, S, s) + bs[0] + zmv;
    }
    as[s] = z;
    mx = fmaxf(mx, z);
  }
  mx = wave_max(mx);
  if (l == 0) sc[w] = mx;
  __syncthreads();
  mx = sc[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) mx = fmaxf(mx, sc[i]);
  float den = 0.f;
  for (int s = tid; s < S; s += (NW * 64)) {
    const float ex = expf(as[s] - mx);
    as[s] = ex;
    den += ex;
  }
  den = wave_sum(den);
  if (l == 0) sc[NW + w] = den;
  __syncthreads();
  den = sc[NW];
#pragma unroll
  for (int i = 1; i < NW; ++i) den += sc[NW + i];
  const float inv = 1.f / den;
  for (int s = tid; s < S; s += (NW * 64)) {
    const float v = as[s] * inv;
    as[s] = v;
    a[(size_t)b * S + s] = v;
  }
  __syncthreads();
  // ---- phase 3: jv[m] = qf[m] + sum_s I[m,s] a[s]; one wave per row, kAttLoads rows in flight
  const float* Ib = I + (size_t)b * M * S;
  for (int m0 = w * kAttLoads; m0 < M; m0 += NW * kAttLoads) {
    float part[kAttLoads];
#pragma unroll
    for (int r = 0; r < kAttLoads; ++r) part[r] = 0.f;
    for (int q = l; q < S4; q += 64) {
      const float4 av = reinterpret_cast<const float4*>(as)[q];
      float4 x[kAttLoads];
#pragma unroll
      for (int r = 0; r < kAttLoads; ++r)
        x[r] = m0 + r < M ? reinterpret_cast<const float4*>(Ib + (size_t)(m0 + r) * S)[q]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int r = 0; r < kAttLoads; ++r)
        part[r] += x[r].x * av.x + x[r].y * av.y + x[r].z * av.z + x[r].w * av.w;
    }
#pragma unroll
    for (int r = 0; r < kAttLoads; ++r) {
      const float v = wave_sum(part[r]);
      if (l == 0 && m0 + r < M) jv[(size_t)b * M + m0 + r] = v + qf[(size_t)b * M + m0 + r];
    }
  }
}
// ---- the same forward pass with its two row streams (P and I, 200 + 400 KB per sample at the
// reference's sizes) brought in by LDS-DMA (round 3).  The register-staged kernel above keeps
// kAttLoads rows per wave in flight in VGPRs: 25-50 KB per compute unit, which beside a resident
// conv tile (memory latency of several microseconds) bounds it at ~140 us per launch in the step
// against 52 us alone -- and more rows in flight means more registers, i.e. a workgroup that no
// longer fits next to the conv tile at all.  Here every wave owns a private ring of D row slots in
// LDS: `global_load_lds` fills slot (i + D) % D while row i is consumed, no staging registers, no
// barrier (nobody else reads a wave's ring), ~48 VGPRs per wave.  Per-row scalars (u[k], ws[k]) sit
// in one register per lane and are broadcast with v_readlane; results (jv) likewise collect in a
// lane each and leave in one store at the end: no other memory instruction inside the streaming
// loops, so the only vmcnt traffic is the ring's and the counted waits are exact.
// The LDS reads of the ring are inline asm: hipcc would wait vmcnt(0) -- draining the ring -- in
// front of every LDS read it can see while a DMA is in flight.
typedef __attribute__((address_space(3))) void* att_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* att_glb_ptr_t;

template <int NW, int D>
__global__ __launch_bounds__(NW * 64) void k_att_fwd_dma(
    int M, int A, int S, const float* __restrict__ P, const float* __restrict__ u,
    const float* __restrict__ ws, const float* __restrict__ bs, const float* __restrict__ zm,
    const float* __restrict__ I, const float* __restrict__ qf, float* __restrict__ a,
    float* __restrict__ jv, AttPartials ap) {
  RAU_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* red = sm;                     // [NW][S]
  float* as = sm + NW * S;             // [S]
  float* sc = as + S;                  // [2*NW] block scalars
  float* us = sc + 2 * NW;             // [A]
  float* zs = us + A;                  // [S]
  float* ring = zs + S;                // [NW][D][S] (+ slack: lanes past S/4 read beyond a slot)
  const int b = blockIdx.x, tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S4 = S >> 2;
  const float* Pb = P + (size_t)b * A * S;
  const float* ub = u + (size_t)b * A;
  if (ap.u_ns) {
    for (int k = tid; k < A; k += NW * 64) {
      float v = ap.u_bias[k];
      for (int sp = 0; sp < ap.u_ns; ++sp) v += u[((size_t)sp * gridDim.x + b) * A + k];
      us[k] = v;
    }
  }
  const int SL = ap.SL > 0 ? ap.SL : S;   // logical positions; [SL, S) are pad columns
  if (ap.z_ns) {
    for (int s = tid; s < SL; s += NW * 64) {
      float v = ap.z_bias[s];
      for (int sp = 0; sp < ap.z_ns; ++sp) v += zm[((size_t)sp * gridDim.x + b) * SL + s];
      zs[s] = v;
    }
  }
  if (ap.u_ns || ap.z_ns) __syncthreads();
  if (ap.u_out)
    for (int k = tid; k < A; k += NW * 64) ap.u_out[(size_t)b * A + k] = ap.u_ns ? us[k] : ub[k];

  float* myring = ring + (size_t)w * D * S;
  const uint32_t ring_b = (uint32_t)(size_t)(att_lds_ptr_t)myring + (uint32_t)l * 16;
  const bool lane_on = l < S4;
  // row i of this wave = row w + NW * i of the tile at `base`; its slot is i % D
  auto dma_row = [&](const float* base, int i) {
    if (lane_on)
      __builtin_amdgcn_global_load_lds((att_glb_ptr_t)(base + (size_t)(w + NW * i) * S + 4 * l),
                                       (att_lds_ptr_t)(myring + (i % D) * S), 16, 0, 0);
  };
  auto ring_read = [&](int i) {
    float4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ring_b + (uint32_t)((i % D) * S) * 4) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return v;
  };

  // ---- phase 1: e[s] = sum_k ws[k] tanh(P[k,s] + u[k]); lane j holds (u, ws) of the wave's row j
  {
    const int nrows = w < A ? (A - w + NW - 1) / NW : 0;
    float urow = 0.f, wrow = 0.f;
    if (l < nrows) {
      const int k = w + NW * l;
      urow = ap.u_ns ? us[k] : ub[k];
      wrow = ws[k];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // nothing of the above counts below
    for (int i = 0; i < D && i < nrows; ++i) dma_row(Pb, i);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int i = 0; i < nrows; ++i) {
      // rows i+1 .. i+D-1 may still be in flight (fewer at the tail)
      if (i + D <= nrows) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float4 p = ring_read(i);
      if (i + D < nrows) dma_row(Pb, i + D);       // the slot just read is free again
      const float uk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, urow), i));
      const float wk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wrow), i));
      if (lane_on) {
        acc.x += wk * tanh_fast(p.x + uk); acc.y += wk * tanh_fast(p.y + uk);
        acc.z += wk * tanh_fast(p.z + uk); acc.w += wk * tanh_fast(p.w + uk);
      }
    }
    if (lane_on) reinterpret_cast<float4*>(red + (size_t)w * S)[l] = acc;
  }
  __syncthreads();
  // ---- phase 2: a = softmax(e + bs + zm)   (as in k_att_fwd_fused)
  float mx = -INFINITY;
  for (int s = tid; s < S; s += (NW * 64)) {
    float z = -INFINITY;
    if (s < SL) {
      const float zmv = ap.z_ns ? zs[s] : zm[(size_t)b * S + s];
      z = block_sum_ordered<NW>(red, S, s) + bs[0] + zmv;
    }
    as[s] = z;
    mx = fmaxf(mx, z);
  }
  mx = wave_max(mx);
  if (l == 0) sc[w] = mx;
  __syncthreads();
  mx = sc[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) mx = fmaxf(mx, sc[i]);
  float den = 0.f;
  for (int s = tid; s < S; s += (NW * 64)) {
    const float ex = expf(as[s] - mx);
    as[s] = ex;
    den += ex;
  }
  den = wave_sum(den);
  if (l == 0) sc[NW + w] = den;
  __syncthreads();
  den = sc[NW];
#pragma unroll
  for (int i = 1; i < NW; ++i) den += sc[NW + i];
  const float inv = 1.f / den;
  for (int s = tid; s < S; s += (NW * 64)) {
    const float v = as[s] * inv;
    as[s] = v;
    a[(size_t)b * S + s] = v;
  }
  __syncthreads();
  // ---- phase 3: jv[m] = qf[m] + sum_s I[m,s] a[s]; lane j collects the wave's row j
  {
    const float* Ib = I + (size_t)b * M * S;
    const int nrows = w < M ? (M - w + NW - 1) / NW : 0;
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane_on) av = reinterpret_cast<const float4*>(as)[l];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < D && i < nrows; ++i) dma_row(Ib, i);
    float res = 0.f;
#pragma unroll 1
    for (int i = 0; i < nrows; ++i) {
      if (i + D <= nrows) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float4 x = ring_read(i);
      if (i + D < nrows) dma_row(Ib, i + D);
      float part = lane_on ? (x.x * av.x + x.y * av.y) + (x.z * av.z + x.w * av.w) : 0.f;
      part = wave_sum(part);
      if (l == i) res = part;
    }
    if (l < nrows) {
      const size_t e = (size_t)b * M + w + NW * l;
      jv[e] = res + qf[e];
    }
  }
}

hipError_t att_fwd_fused(hipStream_t st, int nB, int M, int A, int S, const float* P,
                         const float* u, const float* ws, const float* bs, const float* zm,
                         const float* I, const float* qf, float* T, float* a, float* jv,
                         const AttPartials& ap) {
  if (!split_span_ok(u, ap.u_ns, (size_t)nB * A) ||
      !split_span_ok(zm, ap.z_ns, (size_t)nB * (ap.SL ? ap.SL : S)))
    return kSplitStateError;
  const int nw = att_waves(false, ap.waves);
  // 14 x 14 (and any S % 4 == 0, S <= 256) maps on the step path (tanh(P + u) not kept): the LDS-DMA kernel
  static const bool dma_off = std::getenv("RAU_ATT_DMA_OFF") != nullptr;   // A/B knob
  if (!dma_off && !T && S % 4 == 0 && S <= 256 && A <= 64 * 8 && M <= 64 * 8 && (nw == 8 || nw == 16)) {
    constexpr int kD8 = 8, kD16 = 4;
    const int d = nw == 8 ? kD8 : kD16;
    const size_t ldsd = ((size_t)(nw + 2) * S + 2 * nw + A + (size_t)nw * d * S + 256) * sizeof(float);
    static const bool attr = [] {   // once per process; magic static (contexts on several host threads)
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_att_fwd_dma<8, kD8>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_att_fwd_dma<16, kD16>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      return true;
    }();
    (void)attr;
    if (ldsd <= 96 * 1024) {
      if (nw == 8)
        hipLaunchKernelGGL((k_att_fwd_dma<8, kD8>), dim3(nB), dim3(512), ldsd, st, M, A, S, P, u, ws, bs, zm, I,
                           qf, a, jv, ap);
      else
        hipLaunchKernelGGL((k_att_fwd_dma<16, kD16>), dim3(nB), dim3(1024), ldsd, st, M, A, S, P, u, ws, bs, zm,
                           I, qf, a, jv, ap);
      return hipGetLastError();
    }
  }
  const size_t lds = ((size_t)(nw + 2) * S + 2 * nw + A) * sizeof(float);
#define ATT_FWD(NW_) hipLaunchKernelGGL(k_att_fwd_fused<NW_>, dim3(nB), dim3(NW_ * 64), lds, st, M, A, \
                                        S, P, u, ws, bs, zm, I, qf, T, a, jv, ap)
  if (nw == 8) ATT_FWD(8); else ATT_FWD(16);
#undef ATT_FWD
  return hipGetLastError();
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_att_bwd_fused(
    int M, int A, int S, const float* __restrict__ I, const float* __restrict__ dj,
    const float* __restrict__ a, const float* __restrict__ da_lin, const float* __restrict__ ws,
    float* __restrict__ T, float* __restrict__ dz, float* __restrict__ du,
    float* __restrict__ dwsp, const float* __restrict__ Psrc, const float* __restrict__ u,
    int da_ns, int SL, int nB, const float* __restrict__ da_add) {
  RAU_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* red = sm;                     // [NW][S]
  float* dzs = sm + NW * S;     // [S]
  float* sc = dzs + S;                 // [NW]
  const int b = blockIdx.x, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int S4 = S >> 2;
  const float* Ib = I + (size_t)b * M * S;
  const float* djb = dj + (size_t)b * M;
  // ---- phase 1: da[s] = da_lin[s] + sum_m dj[m] I[m,s]   (kAttLoads rows of I in flight)
  for (int q0 = 0; q0 < S4; q0 += 64) {
    const int q = q0 + l;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < S4) {
      for (int m0 = w; m0 < M; m0 += NW * kAttLoads) {
        float4 x[kAttLoads];
#pragma unroll
        for (int i = 0; i < kAttLoads; ++i) {
          const int m = m0 + i * NW;
          x[i] = m < M ? reinterpret_cast<const float4*>(Ib + (size_t)m * S)[q]
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < kAttLoads; ++i) {
          const int m = m0 + i * NW;
          const float d = m < M ? djb[m] : 0.f;
          acc.x += d * x[i].x; acc.y += d * x[i].y; acc.z += d * x[i].z; acc.w += d * x[i].w;
        }
      }
      reinterpret_cast<float4*>(red + (size_t)w * S)[q] = acc;
    }
  }
  __syncthreads();
  // ---- phase 2: dz = a * (da - sum_s a da)
  float dot = 0.f;
  for (int s = tid; s < S; s += (NW * 64)) {
    float d;
    if (da_ns > 0) {   // da_lin = K-split partials [split][nB][SL] of dj Wf, summed here in order
      d = 0.f;
      if (s < SL) {
        for (int sp = 0; sp < da_ns; ++sp) d += da_lin[((size_t)sp * nB + b) * SL + s];
        if (da_add) d += da_add[(size_t)b * S + s];
      }
    } else {
      d = da_lin[(size_t)b * S + s];
    }
    d += block_sum_ordered<NW>(red, S, s);
    dzs[s] = d;
    dot += a[(size_t)b * S + s] * d;
  }
  dot = wave_sum(dot);
  if (l == 0) sc[w] = dot;
  __syncthreads();
  dot = sc[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) dot += sc[i];
  for (int s = tid; s < S; s += (NW * 64)) {
    const float v = a[(size_t)b * S + s] * (dzs[s] - dot);
    dzs[s] = v;
    dz[(size_t)b * S + s] = v;
  }
  __syncthreads();
  // ---- phase 3: T -> dS in place, du[k] = sum_s dS, dwsp[k] = sum_s dz T; wave per row.
  // With Psrc the forward did not keep T: it is recomputed as tanh(Psrc + u) (Psrc may be the
  // very buffer dS is written to: same thread, same address, read first).
  float* Tb = T + (size_t)b * A * S;
  const float* Pb = Psrc ? Psrc + (size_t)b * A * S : nullptr;
  constexpr int UB = 4;   // rows of P/T in flight per wave
  for (int k0 = w; k0 < A; k0 += NW * UB) {
    float s1[UB], s2[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) s1[i] = s2[i] = 0.f;
    for (int q = l; q < S4; q += 64) {
      const float4 d = reinterpret_cast<const float4*>(dzs)[q];
      float4 t[UB];
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int k = k0 + i * NW;
        t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < A)
          t[i] = Psrc ? reinterpret_cast<const float4*>(Pb + (size_t)k * S)[q]
                      : reinterpret_cast<const float4*>(Tb + (size_t)k * S)[q];
      }
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int k = k0 + i * NW;
        if (k < A) {
          const float wk = ws[k];
          if (Psrc) {
            const float uk = u[(size_t)b * A + k];
            t[i].x = tanh_fast(t[i].x + uk); t[i].y = tanh_fast(t[i].y + uk);
            t[i].z = tanh_fast(t[i].z + uk); t[i].w = tanh_fast(t[i].w + uk);
          }
          float4 o;
          o.x = d.x * wk * (1.f - t[i].x * t[i].x);
          o.y = d.y * wk * (1.f - t[i].y * t[i].y);
          o.z = d.z * wk * (1.f - t[i].z * t[i].z);
          o.w = d.w * wk * (1.f - t[i].w * t[i].w);
          reinterpret_cast<float4*>(Tb + (size_t)k * S)[q] = o;
          s1[i] += (o.x + o.y) + (o.z + o.w);
          s2[i] += d.x * t[i].x + d.y * t[i].y + d.z * t[i].z + d.w * t[i].w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int k = k0 + i * NW;
      const float v1 = wave_sum(s1[i]);
      const float v2 = wave_sum(s2[i]);
      if (l == 0 && k < A) {
        du[(size_t)b * A + k] = v1;
        dwsp[(size_t)b * A + k] = v2;
      }
    }
  }
}
// ---- the backward pass on the LDS-DMA structure of k_att_fwd_dma (round 3): I rows (phase 1) and
// P rows (phase 3) arrive in per-wave rings of D slots.  Phase 3 also STORES a row of dS per
// iteration (in place over P), so a wave's vmcnt sequence is DMA, store, DMA, store, ...: with one
// of each per iteration -- the tail keeps issuing (unused) loads of its last row so the pattern
// never changes -- row i has landed once at most 2 D - 2 younger operations are outstanding (D - 1
// while the ring's first lap is still in flight; waiting for a smaller count is always safe).
// Only the recompute form (Psrc != nullptr: the forward kept P, not tanh(P + u)) is built this way.
// DS16: dS leaves as bf16 (RNE) into its own [b][A][S] buffer instead of f32 in place over P -- RAU_BF16
// mode, where both consumers of dS (attention dgrad, att_i weight gradient) round it to bf16 while
// staging anyway: same values, half the bytes written here and read there.
template <int NW, int D, bool DS16>
__global__ __launch_bounds__(NW * 64) void k_att_bwd_dma(
    int M, int A, int S, const float* __restrict__ I, const float* __restrict__ dj,
    const float* __restrict__ a, const float* __restrict__ da_lin, const float* __restrict__ ws,
    float* __restrict__ T, float* __restrict__ dz, float* __restrict__ du, float* __restrict__ dwsp,
    const float* __restrict__ Psrc, const float* __restrict__ u, int da_ns, int SL, int nB,
    const float* __restrict__ da_add, uint16_t* __restrict__ dS16) {
  RAU_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* red = sm;                     // [NW][S]
  float* dzs = sm + NW * S;            // [S]
  float* sc = dzs + S;                 // [NW]
  float* ring = sc + NW;               // [NW][D][S] (+ slack)
  const int b = blockIdx.x, tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S4 = S >> 2;
  const bool lane_on = l < S4;
  float* myring = ring + (size_t)w * D * S;
  const uint32_t ring_b = (uint32_t)(size_t)(att_lds_ptr_t)myring + (uint32_t)l * 16;
  auto dma_row = [&](const float* base, int i, int row) {   // row `row` of the wave into slot i % D
    if (lane_on)
      __builtin_amdgcn_global_load_lds((att_glb_ptr_t)(base + (size_t)(w + NW * row) * S + 4 * l),
                                       (att_lds_ptr_t)(myring + (i % D) * S), 16, 0, 0);
  };
  auto ring_read = [&](int i) {
    float4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ring_b + (uint32_t)((i % D) * S) * 4) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return v;
  };
  auto lane_val = [&](float v, int i) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
  };
  // ---- phase 1: da[s] = da_lin[s] + sum_m dj[m] I[m,s]
  {
    const float* Ib = I + (size_t)b * M * S;
    const int nrows = w < M ? (M - w + NW - 1) / NW : 0;
    float drow = 0.f;
    if (l < nrows) drow = dj[(size_t)b * M + w + NW * l];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < D && i < nrows; ++i) dma_row(Ib, i, i);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int i = 0; i < nrows; ++i) {
      if (i + D <= nrows) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float4 x = ring_read(i);
      if (i + D < nrows) dma_row(Ib, i + D, i + D);
      const float d = lane_val(drow, i);
      acc.x += d * x.x; acc.y += d * x.y; acc.z += d * x.z; acc.w += d * x.w;
    }
    if (lane_on) reinterpret_cast<float4*>(red + (size_t)w * S)[l] = acc;
  }
  __syncthreads();
  // ---- phase 2: dz = a * (da - sum_s a da)   (as in k_att_bwd_fused)
  float dot = 0.f;
  for (int s = tid; s < S; s += (NW * 64)) {
    float d;
    if (da_ns > 0) {
      d = 0.f;
      if (s < SL) {
        for (int sp = 0; sp < da_ns; ++sp) d += da_lin[((size_t)sp * nB + b) * SL + s];
        if (da_add) d += da_add[(size_t)b * S + s];
      }
    } else {
      d = da_lin[(size_t)b * S + s];
    }
    d += block_sum_ordered<NW>(red, S, s);
    dzs[s] = d;
    dot += a[(size_t)b * S + s] * d;
  }
  dot = wave_sum(dot);
  if (l == 0) sc[w] = dot;
  __syncthreads();
  dot = sc[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) dot += sc[i];
  for (int s = tid; s < S; s += (NW * 64)) {
    const float v = a[(size_t)b * S + s] * (dzs[s] - dot);
    dzs[s] = v;
    dz[(size_t)b * S + s] = v;
  }
  __syncthreads();
  // ---- phase 3: dS = dz ws (1 - tanh(P + u)^2) in place over P; du[k] = sum_s dS; dwsp[k] = sum_s dz T
  {
    float* Tb = T + (size_t)b * A * S;
    const float* Pb = Psrc + (size_t)b * A * S;
    const int nrows = w < A ? (A - w + NW - 1) / NW : 0;
    float urow = 0.f, wrow = 0.f;
    if (l < nrows) {
      urow = u[(size_t)b * A + w + NW * l];
      wrow = ws[w + NW * l];
    }
    float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane_on) d4 = reinterpret_cast<const float4*>(dzs)[l];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (nrows > 0)
      for (int i = 0; i < D; ++i) dma_row(Pb, i, i < nrows ? i : nrows - 1);
    float r1 = 0.f, r2 = 0.f;
#pragma unroll 1
    for (int i = 0; i < nrows; ++i) {
      if (i >= D - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * D - 2) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
      float4 t = ring_read(i);
      dma_row(Pb, i + D, i + D < nrows ? i + D : nrows - 1);   // one load per iteration, always
      const float uk = lane_val(urow, i), wk = lane_val(wrow, i);
      float s1 = 0.f, s2 = 0.f;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane_on) {
        t.x = tanh_fast(t.x + uk); t.y = tanh_fast(t.y + uk);
        t.z = tanh_fast(t.z + uk); t.w = tanh_fast(t.w + uk);
        o.x = d4.x * wk * (1.f - t.x * t.x);
        o.y = d4.y * wk * (1.f - t.y * t.y);
        o.z = d4.z * wk * (1.f - t.z * t.z);
        o.w = d4.w * wk * (1.f - t.w * t.w);
        s1 = (o.x + o.y) + (o.z + o.w);
        s2 = d4.x * t.x + d4.y * t.y + d4.z * t.z + d4.w * t.w;
      }
      // one store per iteration, always (its lanes past S/4 are masked off)
      if (DS16) {
        typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
        b16x4 ob;
        ob[0] = (__bf16)o.x; ob[1] = (__bf16)o.y; ob[2] = (__bf16)o.z; ob[3] = (__bf16)o.w;
        if (lane_on)
          reinterpret_cast<uint2*>(dS16 + ((size_t)b * A + w + NW * i) * S)[l] = __builtin_bit_cast(uint2, ob);
      } else {
        if (lane_on) reinterpret_cast<float4*>(Tb + (size_t)(w + NW * i) * S)[l] = o;
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (l == i) { r1 = s1; r2 = s2; }
    }
    if (l < nrows) {
      du[(size_t)b * A + w + NW * l] = r1;
      dwsp[(size_t)b * A + w + NW * l] = r2;
    }
  }
}

// whether att_bwd_fused takes the LDS-DMA kernel for these sizes (the only one that can write dS as bf16)
static bool att_bwd_dma_sizes(int nw, int M, int A, int S) {
  static const bool dma_off = std::getenv("RAU_ATT_DMA_OFF") != nullptr;   // A/B knob
  if (dma_off || S % 4 != 0 || S > 256 || A > 64 * 8 || M > 64 * 8 || (nw != 8 && nw != 16)) return false;
  const int d = nw == 8 ? 8 : 4;
  return ((size_t)(nw + 1) * S + nw + (size_t)nw * d * S + 256) * sizeof(float) <= 96 * 1024;
}
bool att_bwd_dma_ok(int M, int A, int S, int waves_hint) { return att_bwd_dma_sizes(att_waves(true, waves_hint), M, A, S); }

hipError_t att_bwd_fused(hipStream_t st, int nB, int M, int A, int S, const float* I,
                         const float* dj, const float* a, const float* da_lin,
                         const float* ws, float* T_to_dS, float* dz, float* du, float* dwsp,
                         const float* Psrc, const float* u, int da_ns, int SL,
                         const float* da_add, void* dS16, int waves_hint) {
  if (!split_span_ok(da_lin, da_ns, (size_t)nB * (SL ? SL : S))) return kSplitStateError;
  const int nw = att_waves(true, waves_hint);
  if (Psrc && u && att_bwd_dma_sizes(nw, M, A, S)) {
    constexpr int kD8 = 8, kD16 = 4;
    const int d = nw == 8 ? kD8 : kD16;
    const size_t ldsd = ((size_t)(nw + 1) * S + nw + (size_t)nw * d * S + 256) * sizeof(float);
#define ATT_BWD_DMA(NW_, D_, B16_)                                                                     \
  do {                                                                                                 \
    static const bool attr = [] {   /* once per process, thread-safe */                                \
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_att_bwd_dma<NW_, D_, B16_>),                 \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                      \
      return true;                                                                                     \
    }();                                                                                               \
    (void)attr;                                                                                        \
    hipLaunchKernelGGL((k_att_bwd_dma<NW_, D_, B16_>), dim3(nB), dim3(NW_ * 64), ldsd, st, M, A, S, I, \
                       dj, a, da_lin, ws, T_to_dS, dz, du, dwsp, Psrc, u, da_ns, SL, nB, da_add,       \
                       reinterpret_cast<uint16_t*>(dS16));                                             \
  } while (0)
    if (nw == 8) { if (dS16) ATT_BWD_DMA(8, kD8, true); else ATT_BWD_DMA(8, kD8, false); }
    else { if (dS16) ATT_BWD_DMA(16, kD16, true); else ATT_BWD_DMA(16, kD16, false); }
#undef ATT_BWD_DMA
    return hipGetLastError();
  }
  if (dS16) return hipErrorInvalidValue;   // callers ask for bf16 dS only where att_bwd_dma_ok says so
  const size_t lds = ((size_t)(nw + 1) * S + nw) * sizeof(float);
#define ATT_BWD(NW_) hipLaunchKernelGGL(k_att_bwd_fused<NW_>, dim3(nB), dim3(NW_ * 64), lds, st, M, A, \
                                        S, I, dj, a, da_lin, ws, T_to_dS, dz, du, dwsp, Psrc, u, \
                                        da_ns, SL, nB, da_add)
  if (nw == 8) ATT_BWD(8); else ATT_BWD(16);
#undef ATT_BWD
  return hipGetLastError();
}

// ------------------------------------------- split per-sample attention kernels
// Same arithmetic as the fused kernels above, cut into small workgroups: NC row chunks per
// sample, one 4-wave workgroup each (<= 64 VGPRs, ~3 KB LDS).  A 8- or 16-wave workgroup has to
// wait until a compute unit can take all of its waves at once, which next to two resident
// bulk-GEMM workgroups (144-176 VGPRs per wave) means waiting for a GEMM tile to retire (measured
// in the overlapped step: 60 us -> 120-470 us per launch); four-wave workgroups always fit, and
// NC x more of them spread over the chip.  Each pass is two launches: per-chunk partial column
// sums over the sample's [rows][S] tile, then a launch whose workgroups all finish the (cheap)
// softmax step redundantly and stream their own chunk of the second tile.
constexpr int kSplitLoads = 4;   // row loads in flight per wave

// part[b][c][s] = sum_{k in chunk c} ws[k] tanh(P[b,k,s] + u[b,k])
__global__ __launch_bounds__(256) void k_att_score_part(
    int nB, int A, int S, const float* __restrict__ P, const float* __restrict__ u,
    const float* __restrict__ ws, float* __restrict__ part, AttPartials ap) {
  RAU_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][S]
  const int c = blockIdx.x, NC = gridDim.x, b = blockIdx.y;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int S4 = S >> 2;
  const int rows = (A + NC - 1) / NC, k_lo = c * rows, k_hi = min(A, k_lo + rows);
  const float* Pb = P + (size_t)b * A * S;
  for (int q0 = 0; q0 < S4; q0 += 64) {
    const int q = q0 + l;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = k_lo + w; k0 < k_hi; k0 += 4 * kSplitLoads) {
      float4 p[kSplitLoads];
      float uk[kSplitLoads];
#pragma unroll
      for (int i = 0; i < kSplitLoads; ++i) {
        const int k = k0 + 4 * i;
        p[i] = (k < k_hi && q < S4) ? reinterpret_cast<const float4*>(Pb + (size_t)k * S)[q]
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
        float v = 0.f;
        if (k < k_hi) {
          if (ap.u_ns) {   // K-split partials of qf Wa^T (+ bias), summed in order
            v = ap.u_bias[k];
            for (int sp = 0; sp < ap.u_ns; ++sp) v += u[((size_t)sp * nB + b) * A + k];
          } else {
            v = u[(size_t)b * A + k];
          }
        }
        uk[i] = v;
      }
#pragma unroll
      for (int i = 0; i < kSplitLoads; ++i) {
        const int k = k0 + 4 * i;
        if (k < k_hi) {
          const float wk = ws[k];
          acc.x += wk * tanh_fast(p[i].x + uk[i]); acc.y += wk * tanh_fast(p[i].y + uk[i]);
          acc.z += wk * tanh_fast(p[i].z + uk[i]); acc.w += wk * tanh_fast(p[i].w + uk[i]);
          if (ap.u_out && q0 == 0 && l == 0) ap.u_out[(size_t)b * A + k] = uk[i];
        }
      }
    }
    if (q < S4) reinterpret_cast<float4*>(sm + (size_t)w * S)[q] = acc;
  }
  __syncthreads();
  for (int s = tid; s < S; s += 256)
    part[((size_t)b * NC + c) * S + s] = (sm[s] + sm[S + s]) + (sm[2 * S + s] + sm[3 * S + s]);
}

// a = softmax(sum_c part + bs + zm) (every chunk's workgroup, redundantly; chunk 0 stores it);
// jv[m] = qf[m] + sum_s I[m,s] a[s] for the rows m of chunk c
__global__ __launch_bounds__(256) void k_att_ctx(
    int nB, int M, int S, const float* __restrict__ part, const float* __restrict__ bs,
    const float* __restrict__ zm, const float* __restrict__ I, const float* __restrict__ qf,
    float* __restrict__ a, float* __restrict__ jv, AttPartials ap) {
  RAU_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];   // as[S] + 8 scalars
  float* as = sm;
  float* sc = sm + S;
  const int c = blockIdx.x, NC = gridDim.x, b = blockIdx.y;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int S4 = S >> 2;
  const int SL = ap.SL > 0 ? ap.SL : S;
  float mx = -INFINITY;
  for (int s = tid; s < S; s += 256) {
    float z = -INFINITY;   // pad positions take no attention (and so no gradient)
    if (s < SL) {
      float zmv;
      if (ap.z_ns) {
        zmv = ap.z_bias[s];
        for (int sp = 0; sp < ap.z_ns; ++sp) zmv += zm[((size_t)sp * nB + b) * SL + s];
      } else {
        zmv = zm[(size_t)b * S + s];
      }
      float e = part[(size_t)b * NC * S + s];
      for (int cc = 1; cc < NC; ++cc) e += part[((size_t)b * NC + cc) * S + s];
      z = e + bs[0] + zmv;
    }
    as[s] = z;
    mx = fmaxf(mx, z);
  }
  mx = wave_max(mx);
  if (l == 0) sc[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
  float den = 0.f;
  for (int s = tid; s < S; s += 256) {
    const float ex = expf(as[s] - mx);
    as[s] = ex;
    den += ex;
  }
  den = wave_sum(den);
  if (l == 0) sc[4 + w] = den;
  __syncthreads();
  den = ((sc[4] + sc[5]) + sc[6]) + sc[7];
  const float inv = 1.f / den;
  for (int s = tid; s < S; s += 256) {
    const float v = as[s] * inv;
    as[s] = v;
    if (c == 0) a[(size_t)b * S + s] = v;
  }
  __syncthreads();
  const int rows = (M + NC - 1) / NC, m_lo = c * rows, m_hi = min(M, m_lo + rows);
  const float* Ib = I + (size_t)b * M * S;
  for (int m0 = m_lo + w * kSplitLoads; m0 < m_hi; m0 += 4 * kSplitLoads) {
    float pr[kSplitLoads];
#pragma unroll
    for (int r = 0; r < kSplitLoads; ++r) pr[r] = 0.f;
    for (int q = l; q < S4; q += 64) {
      const float4 av = reinterpret_cast<const float4*>(as)[q];
      float4 x[kSplitLoads];
#pragma unroll
      for (int r = 0; r < kSplitLoads; ++r)
        x[r] = m0 + r < m_hi ? reinterpret_cast<const float4*>(Ib + (size_t)(m0 + r) * S)[q]
                             : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int r = 0; r < kSplitLoads; ++r)
        pr[r] += x[r].x * av.x + x[r].y * av.y + x[r].z * av.z + x[r].w * av.w;
    }
#pragma unroll
    for (int r = 0; r < kSplitLoads; ++r) {
      const float v = wave_sum(pr[r]);
      if (l == 0 && m0 + r < m_hi) jv[(size_t)b * M + m0 + r] = v + qf[(size_t)b * M + m0 + r];
    }
  }
}

// row chunks per sample: 8 for small batches (the launch should still cover the chip), else 4
static int att_chunks(int nB) {
  const char* e = std::getenv("RAU_ATT_CHUNKS");   // read per call: tests switch it between contexts
  const int n = e ? std::atoi(e) : 0;
  return (n >= 1 && n <= 8) ? n : (nB <= 64 ? 8 : 4);
}
size_t att_split_part_floats(int nB, int S) { return (size_t)nB * 8 * S; }

hipError_t att_fwd_split(hipStream_t st, int nB, int M, int A, int S, const float* P,
                         const float* u, const float* ws, const float* bs, const float* zm,
                         const float* I, const float* qf, float* a, float* jv, float* part,
                         const AttPartials& ap) {
  if (!split_span_ok(u, ap.u_ns, (size_t)nB * A) ||
      !split_span_ok(zm, ap.z_ns, (size_t)nB * (ap.SL ? ap.SL : S)))
    return kSplitStateError;
  const int nc = att_chunks(nB);
  hipLaunchKernelGGL(k_att_score_part, dim3(nc, nB), dim3(256), (size_t)4 * S * sizeof(float), st, nB,
                     A, S, P, u, ws, part, ap);
  hipLaunchKernelGGL(k_att_ctx, dim3(nc, nB), dim3(256), (size_t)(S + 8) * sizeof(float), st, nB, M, S,
                     part, bs, zm, I, qf, a, jv, ap);
  return hipGetLastError();
}

// part[b][c][s] = sum_{m in chunk c} dj[b,m] I[b,m,s]
__global__ __launch_bounds__(256) void k_att_da_part(int M, int S, const float* __restrict__ I,
                                                     const float* __restrict__ dj,
                                                     float* __restrict__ part) {
  RAU_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][S]
  const int c = blockIdx.x, NC = gridDim.x, b = blockIdx.y;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int S4 = S >> 2;
  const int rows = (M + NC - 1) / NC, m_lo = c * rows, m_hi = min(M, m_lo + rows);
  const float* Ib = I + (size_t)b * M * S;
  const float* djb = dj + (size_t)b * M;
  for (int q0 = 0; q0 < S4; q0 += 64) {
    const int q = q0 + l;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m0 = m_lo + w; m0 < m_hi; m0 += 4 * kSplitLoads) {
      float4 x[kSplitLoads];
#pragma unroll
      for (int i = 0; i < kSplitLoads; ++i) {
        const int m = m0 + 4 * i;
        x[i] = (m < m_hi && q < S4) ? reinterpret_cast<const float4*>(Ib + (size_t)m * S)[q]
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < kSplitLoads; ++i) {
        const int m = m0 + 4 * i;
        const float d = m < m_hi ? djb[m] : 0.f;
        acc.x += d * x[i].x; acc.y += d * x[i].y; acc.z += d * x[i].z; acc.w += d * x[i].w;
      }
    }
    if (q < S4) reinterpret_cast<float4*>(sm + (size_t)w * S)[q] = acc;
  }
  __syncthreads();
  for (int s = tid; s < S; s += 256)
    part[((size_t)b * NC + c) * S + s] = (sm[s] + sm[S + s]) + (sm[2 * S + s] + sm[3 * S + s]);
}

// da = da_lin + sum_c part; dz = a (da - sum_s a da) (every chunk's workgroup; chunk 0 stores it);
// rows k of chunk c: T -> dS = dz ws[k] (1 - T^2) in place, du[k] = sum_s dS, dwsp[k] = sum_s dz T
__global__ __launch_bounds__(256) void k_att_ds(
    int nB, int A, int S, const float* __restrict__ part, const float* __restrict__ a,
    const float* __restrict__ da_lin, const float* __restrict__ ws, float* __restrict__ T,
    float* __restrict__ dz, float* __restrict__ du, float* __restrict__ dwsp,
    const float* __restrict__ Psrc, const float* __restrict__ u, int da_ns, int SL,
    const float* __restrict__ da_add) {
  RAU_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float sm[];   // dzs[S] + 4 scalars
  float* dzs = sm;
  float* sc = sm + S;
  const int c = blockIdx.x, NC = gridDim.x, b = blockIdx.y;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int S4 = S >> 2;
  float dot = 0.f;
  for (int s = tid; s < S; s += 256) {
    float d;
    if (da_ns > 0) {   // K-split partials [split][nB][SL] of dj Wf, summed in order
      d = 0.f;
      if (s < SL) {
        for (int sp = 0; sp < da_ns; ++sp) d += da_lin[((size_t)sp * nB + b) * SL + s];
        if (da_add) d += da_add[(size_t)b * S + s];
      }
    } else {
      d = da_lin[(size_t)b * S + s];
    }
    float e = part[(size_t)b * NC * S + s];
    for (int cc = 1; cc < NC; ++cc) e += part[((size_t)b * NC + cc) * S + s];
    d += e;
    dzs[s] = d;
    dot += a[(size_t)b * S + s] * d;
  }
  dot = wave_sum(dot);
  if (l == 0) sc[w] = dot;
  __syncthreads();
  dot = ((sc[0] + sc[1]) + sc[2]) + sc[3];
  for (int s = tid; s < S; s += 256) {
    const float v = a[(size_t)b * S + s] * (dzs[s] - dot);
    dzs[s] = v;
    if (c == 0) dz[(size_t)b * S + s] = v;
  }
  __syncthreads();
  const int rows = (A + NC - 1) / NC, k_lo = c * rows, k_hi = min(A, k_lo + rows);
  float* Tb = T + (size_t)b * A * S;
  const float* Pb = Psrc ? Psrc + (size_t)b * A * S : nullptr;
  constexpr int UB = 2;
  for (int k0 = k_lo + w; k0 < k_hi; k0 += 4 * UB) {
    float s1[UB], s2[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) s1[i] = s2[i] = 0.f;
    for (int q = l; q < S4; q += 64) {
      const float4 d = reinterpret_cast<const float4*>(dzs)[q];
      float4 t[UB];
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int k = k0 + 4 * i;
        t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < k_hi)
          t[i] = Psrc ? reinterpret_cast<const float4*>(Pb + (size_t)k * S)[q]
                      : reinterpret_cast<const float4*>(Tb + (size_t)k * S)[q];
      }
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int k = k0 + 4 * i;
        if (k < k_hi) {
          const float wk = ws[k];
          if (Psrc) {
            const float uk = u[(size_t)b * A + k];
            t[i].x = tanh_fast(t[i].x + uk); t[i].y = tanh_fast(t[i].y + uk);
            t[i].z = tanh_fast(t[i].z + uk); t[i].w = tanh_fast(t[i].w + uk);
          }
          float4 o;
          o.x = d.x * wk * (1.f - t[i].x * t[i].x);
          o.y = d.y * wk * (1.f - t[i].y * t[i].y);
          o.z = d.z * wk * (1.f - t[i].z * t[i].z);
          o.w = d.w * wk * (1.f - t[i].w * t[i].w);
          reinterpret_cast<float4*>(Tb + (size_t)k * S)[q] = o;
          s1[i] += (o.x + o.y) + (o.z + o.w);
          s2[i] += d.x * t[i].x + d.y * t[i].y + d.z * t[i].z + d.w * t[i].w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int k = k0 + 4 * i;
      const float v1 = wave_sum(s1[i]);
      const float v2 = wave_sum(s2[i]);
      if (l == 0 && k < k_hi) {
        du[(size_t)b * A + k] = v1;
        dwsp[(size_t)b * A + k] = v2;
      }
    }
  }
}

hipError_t att_bwd_split(hipStream_t st, int nB, int M, int A, int S, const float* I,
                         const float* dj, const float* a, const float* da_lin, const float* ws,
                         float* T_to_dS, float* dz, float* du, float* dwsp, const float* Psrc,
                         const float* u, float* part, int da_ns, int SL, const float* da_add) {
  if (!split_span_ok(da_lin, da_ns, (size_t)nB * (SL ? SL : S))) return kSplitStateError;
  const int nc = att_chunks(nB);
  hipLaunchKernelGGL(k_att_da_part, dim3(nc, nB), dim3(256), (size_t)4 * S * sizeof(float), st, M, S, I,
                     dj, part);
  hipLaunchKernelGGL(k_att_ds, dim3(nc, nB), dim3(256), (size_t)(S + 4) * sizeof(float), st, nB, A, S,
                     part, a, da_lin, ws, T_to_dS, dz, du, dwsp, Psrc, u, da_ns, SL, da_add);
  return hipGetLastError();
}

// 32x32 LDS-tiled transpose (bulk stream; no chain priority)
__global__ void k_transpose2d(int rows, int cols, const float* __restrict__ in,
                              float* __restrict__ out, __bf16* __restrict__ out16) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8)
    if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = in[(size_t)(r0 + j) * cols + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < cols && r0 + tx < rows) {
      out[(size_t)(c0 + j) * rows + r0 + tx] = tile[tx][j];
      if (out16) out16[(size_t)(c0 + j) * rows + r0 + tx] = (__bf16)tile[tx][j];
    }
}
hipError_t transpose2d(hipStream_t st, int rows, int cols, const float* in, float* out, void* out16) {
  hipLaunchKernelGGL(k_transpose2d, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st,
                     rows, cols, in, out, reinterpret_cast<__bf16*>(out16));
  return hipGetLastError();
}

// feature-map dropout for every hop in one pass (reference SS:239, one mask per clone)
__global__ void k_dropout_features(int H, size_t per4, const float4* __restrict__ X,
                                   const uint32_t* __restrict__ mask, size_t e0, float mscale,
                                   float4* __restrict__ xd) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < per4;
       q += (size_t)gridDim.x * blockDim.x) {
    const float4 x = X[q];
    for (int h = 0; h < H; ++h) {
      const size_t e = e0 + ((size_t)h * per4 + q) * 4;
      const uint32_t nib = mask_nib(mask, e);
      float4 o;
      o.x = (nib & 1u) ? x.x * mscale : 0.f;
      o.y = (nib & 2u) ? x.y * mscale : 0.f;
      o.z = (nib & 4u) ? x.z * mscale : 0.f;
      o.w = (nib & 8u) ? x.w * mscale : 0.f;
      xd[(size_t)h * per4 + q] = o;
    }
  }
}
// RAU_BF16 mode: the same pass writing bf16 (RNE of the f32 value the other form stores, i.e. exactly
// what the bf16 operand staging would have made of it), the only form the bf16 step path reads
__global__ void k_dropout_features_b16(int H, size_t per4, const float4* __restrict__ X,
                                       const uint32_t* __restrict__ mask, size_t e0, float mscale,
                                       uint2* __restrict__ xd) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < per4;
       q += (size_t)gridDim.x * blockDim.x) {
    const float4 x = X[q];
    for (int h = 0; h < H; ++h) {
      const size_t e = e0 + ((size_t)h * per4 + q) * 4;
      const uint32_t nib = mask_nib(mask, e);
      typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
      b16x4 o;
      o[0] = (__bf16)((nib & 1u) ? x.x * mscale : 0.f);
      o[1] = (__bf16)((nib & 2u) ? x.y * mscale : 0.f);
      o[2] = (__bf16)((nib & 4u) ? x.z * mscale : 0.f);
      o[3] = (__bf16)((nib & 8u) ? x.w * mscale : 0.f);
      xd[(size_t)h * per4 + q] = __builtin_bit_cast(uint2, o);
    }
  }
}
// The same two passes drawing their keep bits themselves (the step path when the site's mask is not
// caller-supplied): a thread owns 16 consecutive elements = one Philox block per hop, so the bits never
// travel through memory and fill_masks' pass over the site (44 us at D = 512, 163 us at D = 2048 per
// step, in front of the first conv GEMM) disappears under this pass's HBM time.  Same (seed, site,
// step, element) -> bit function as fill_masks: the masks are bit-identical.
template <bool B16>
__global__ void k_dropout_features_gen(uint64_t seed, uint32_t site, uint32_t step, uint32_t thr,
                                       const uint64_t* __restrict__ key, int H, size_t per16,
                                       const float4* __restrict__ X, float mscale,
                                       void* __restrict__ xd) {
  if (key) {
    seed = key[0];
    step = (uint32_t)key[1];
  }
  // A wave takes 64 consecutive Philox blocks (lane l draws block base + l) but moves the data in the
  // order memory likes: in round r lane l handles quad l & 3 of block base + 16 r + (l >> 2), so every
  // load / store instruction of the wave covers one contiguous KB (the owner's bits come by shuffle).
  // With each lane storing its own block's four quads the lanes of an instruction are 64 bytes apart
  // and every 128-byte line is written in four pieces (262 us at D = 512 against 190 us for a plain
  // copy of the same bytes).
  const int l = threadIdx.x & 63, quad = l & 3, sub = l >> 2;
  for (size_t base = (blockIdx.x * (size_t)blockDim.x + (threadIdx.x & ~63u)); base < per16;
       base += (size_t)gridDim.x * blockDim.x) {
    float4 x[4];
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t blk = base + 16 * r + sub;
      ok[r] = blk < per16;
      x[r] = ok[r] ? X[4 * blk + quad] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const size_t mine = base + l < per16 ? base + l : per16 - 1;
    for (int h = 0; h < H; ++h) {
      const uint32_t bits = philox_keep16(seed, site, step, (size_t)h * per16 + mine, thr);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t nib = (uint32_t)__shfl((int)bits, 16 * r + sub, 64) >> (4 * quad);
        float4 o;
        o.x = (nib & 1u) ? x[r].x * mscale : 0.f;
        o.y = (nib & 2u) ? x[r].y * mscale : 0.f;
        o.z = (nib & 4u) ? x[r].z * mscale : 0.f;
        o.w = (nib & 8u) ? x[r].w * mscale : 0.f;
        const size_t q = ((size_t)h * per16 + base + 16 * r + sub) * 4 + quad;
        if (!ok[r]) continue;
        if (B16) {
          typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
          b16x4 v;
          v[0] = (__bf16)o.x; v[1] = (__bf16)o.y; v[2] = (__bf16)o.z; v[3] = (__bf16)o.w;
          reinterpret_cast<uint2*>(xd)[q] = __builtin_bit_cast(uint2, v);
        } else {
          reinterpret_cast<float4*>(xd)[q] = o;
        }
      }
    }
  }
}
hipError_t dropout_features_gen(hipStream_t st, uint64_t seed, uint32_t site, uint32_t step, float p,
                                const uint64_t* key_dev, int H, size_t per_hop, const float* X,
                                float mscale, void* xd, int b16) {
  if (per_hop % 16 != 0) return hipErrorInvalidValue;
  const uint32_t thr = (uint32_t)lroundf(p * 256.0f);
  const size_t per16 = per_hop / 16;
  if (b16)
    hipLaunchKernelGGL(k_dropout_features_gen<true>, dim3(grid_for(per16)), dim3(256), 0, st, seed, site,
                       step, thr, key_dev, H, per16, reinterpret_cast<const float4*>(X), mscale, xd);
  else
    hipLaunchKernelGGL(k_dropout_features_gen<false>, dim3(grid_for(per16)), dim3(256), 0, st, seed, site,
                       step, thr, key_dev, H, per16, reinterpret_cast<const float4*>(X), mscale, xd);
  return hipGetLastError();
}
// Pitched rows (S not a multiple of 4): the mask is defined over the LOGICAL tensor
// [.., rows, SL], the data has row pitch Sp; pad columns are written as zeros.
__global__ void k_dropout_features_pitch(int H, size_t rows, int SL, int Sp,
                                         const float* __restrict__ X,
                                         const uint32_t* __restrict__ mask, size_t e0, float mscale,
                                         float* __restrict__ xd) {
  const size_t n = rows * Sp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / Sp;
    const int s = (int)(i - r * Sp);
    const float x = X[i];
    for (int h = 0; h < H; ++h) {
      float o = 0.f;
      if (s < SL) o = mask_bit(mask, e0 + ((size_t)h * rows + r) * SL + s) ? x * mscale : 0.f;
      xd[(size_t)h * n + i] = o;
    }
  }
}
hipError_t dropout_features(hipStream_t st, int H, size_t per_hop, const float* X,
                            const uint32_t* mask, float mscale, float* xd, size_t mask_e0, int SL,
                            int Sp) {
  if (SL != Sp) {
    hipLaunchKernelGGL(k_dropout_features_pitch, dim3(grid_for(per_hop)), dim3(256), 0, st, H,
                       per_hop / Sp, SL, Sp, X, mask, mask_e0, mscale, xd);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_dropout_features, dim3(grid_for(per_hop / 4)), dim3(256), 0, st, H,
                     per_hop / 4, reinterpret_cast<const float4*>(X), mask, mask_e0, mscale,
                     reinterpret_cast<float4*>(xd));
  return hipGetLastError();
}

hipError_t dropout_features_b16(hipStream_t st, int H, size_t per_hop, const float* X,
                                const uint32_t* mask, float mscale, void* xd16) {
  hipLaunchKernelGGL(k_dropout_features_b16, dim3(grid_for(per_hop / 4)), dim3(256), 0, st, H,
                     per_hop / 4, reinterpret_cast<const float4*>(X), mask, (size_t)0, mscale,
                     reinterpret_cast<uint2*>(xd16));
  return hipGetLastError();
}

// ----------------------------------------------- deterministic column sums
// dst[n] += sum_r X[r, n]: stage 1 sums kColChunks row chunks (fixed order inside a chunk),
// stage 2 adds the chunk partials in order.  The vector path (N, ld multiples of 4) gives a
// 256-thread workgroup 256 columns x one row chunk: each wave takes every 4th row of the chunk
// with float4 loads, four independent accumulators, then the waves combine through LDS.
constexpr int kColChunks = 32;
__global__ void k_colsum_stage1(int rows, int N, const float* __restrict__ X, long ld,
                                float* __restrict__ tmp) {
  RAU_CHAIN_PRIO();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int per = (rows + kColChunks - 1) / kColChunks;
  const int r0 = blockIdx.y * per;
  int r1 = r0 + per;
  if (r1 > rows) r1 = rows;
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc += X[(size_t)r * ld + n];
  tmp[(size_t)blockIdx.y * N + n] = acc;
}
__global__ __launch_bounds__(256) void k_colsum_stage1_v4(int rows, int N,
                                                         const float* __restrict__ X, long ld,
                                                         float* __restrict__ tmp) {
  __shared__ float4 part[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + l) * 4;
  const int per = (rows + kColChunks - 1) / kColChunks;
  const int r0 = blockIdx.y * per;
  int r1 = r0 + per;
  if (r1 > rows) r1 = rows;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (n < N) {
    const float* p = X + n;
    int r = r0 + w;
    for (; r + 12 < r1; r += 16) {
      const float4 x0 = *reinterpret_cast<const float4*>(p + (size_t)r * ld);
      const float4 x1 = *reinterpret_cast<const float4*>(p + (size_t)(r + 4) * ld);
      const float4 x2 = *reinterpret_cast<const float4*>(p + (size_t)(r + 8) * ld);
      const float4 x3 = *reinterpret_cast<const float4*>(p + (size_t)(r + 12) * ld);
      a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
      a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
      a2.x += x2.x; a2.y += x2.y; a2.z += x2.z; a2.w += x2.w;
      a3.x += x3.x; a3.y += x3.y; a3.z += x3.z; a3.w += x3.w;
    }
    for (; r < r1; r += 4) {
      const float4 x0 = *reinterpret_cast<const float4*>(p + (size_t)r * ld);
      a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
    }
  }
  float4 a;
  a.x = (a0.x + a1.x) + (a2.x + a3.x);
  a.y = (a0.y + a1.y) + (a2.y + a3.y);
  a.z = (a0.z + a1.z) + (a2.z + a3.z);
  a.w = (a0.w + a1.w) + (a2.w + a3.w);
  part[w][l] = a;
  __syncthreads();
  if (w == 0 && n < N) {
    const float4 b = part[1][l], c = part[2][l], d = part[3][l];
    float4 o;
    o.x = (a.x + b.x) + (c.x + d.x);
    o.y = (a.y + b.y) + (c.y + d.y);
    o.z = (a.z + b.z) + (c.z + d.z);
    o.w = (a.w + b.w) + (c.w + d.w);
    *reinterpret_cast<float4*>(tmp + (size_t)blockIdx.y * N + n) = o;
  }
}
__global__ void k_colsum_stage2(int N, const float* __restrict__ tmp, float* __restrict__ dst) {
  RAU_CHAIN_PRIO();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int c = 0; c < kColChunks; ++c) acc += tmp[(size_t)c * N + n];
  dst[n] += acc;
}
hipError_t colsum_acc(hipStream_t st, int rows, int N, const float* X, long ld, float* dst,
                      float* tmp) {
  const bool vec = (N % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15u) == 0) &&
                   rows >= 4 * kColChunks;
  if (vec)
    hipLaunchKernelGGL(k_colsum_stage1_v4, dim3((N + 255) / 256, kColChunks), dim3(256), 0, st,
                       rows, N, X, ld, tmp);
  else
    hipLaunchKernelGGL(k_colsum_stage1, dim3((N + 63) / 64, kColChunks), dim3(64), 0, st, rows, N,
                       X, ld, tmp);
  hipLaunchKernelGGL(k_colsum_stage2, dim3((N + 63) / 64), dim3(64), 0, st, N, tmp, dst);
  return hipGetLastError();
}

// ------------------------------------------------------- classifier loss head
// One block per sample.  CrossEntropyCriterion (SS:310,518) + torch.max first-max
// argmax (SS:488) + do_pred = sigmoid(mf . wd + bd) (SS:281).
__global__ void k_ce_fwd(int nB, int K, int M, const float* __restrict__ logits,
                         const int32_t* __restrict__ labels, const float* __restrict__ mf,
                         const float* __restrict__ wd, const float* __restrict__ bd,
                         float* __restrict__ dl, float* __restrict__ lossrow,
                         int32_t* __restrict__ argmax, float* __restrict__ dopred,
                         const float* __restrict__ part, int nsplit,
                         const float* __restrict__ bias, float* __restrict__ logits_out,
                         int Bper) {
  RAU_CHAIN_PRIO();
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ float s_sum[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  if (nsplit) {  // logits arrive as K-split partials [split][nB][K] of mf Wc^T: finish them here
    for (int k = tid; k < K; k += 256) {
      float v = bias[k];
      for (int sp = 0; sp < nsplit; ++sp) v += part[((size_t)sp * nB + b) * K + k];
      logits_out[(size_t)b * K + k] = v;
    }
    __syncthreads();   // the label's logit below is read by thread 0
    logits = logits_out;
  }
  const float* lg = logits + (size_t)b * K;
  // max + first argmax
  float mx = -INFINITY;
  int ai = 0x7fffffff;
  for (int k = tid; k < K; k += 256) {
    const float v = lg[k];
    if (v > mx) { mx = v; ai = k; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(ai, o, 64);
    if (ov > mx || (ov == mx && oi < ai)) { mx = ov; ai = oi; }
  }
  if (l == 0) { s_val[w] = mx; s_idx[w] = ai; }
  __syncthreads();
  mx = s_val[0]; ai = s_idx[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (s_val[i] > mx || (s_val[i] == mx && s_idx[i] < ai)) { mx = s_val[i]; ai = s_idx[i]; }
  float den = 0.f;
  for (int k = tid; k < K; k += 256) den += expf(lg[k] - mx);
  den = wave_sum(den);
  if (l == 0) s_sum[w] = den;
  __syncthreads();
  den = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
  const float lse = mx + logf(den);
  if (tid == 0) argmax[b] = ai + 1;
  if (labels) {
    // rows = [hop][sample]: labels repeat every Bper rows; clamped like the token ids above
    const int y = min(max(labels[b % Bper], 1), K) - 1;
    const float invB = 1.f / (float)Bper;
    for (int k = tid; k < K; k += 256) {
      float p = expf(lg[k] - lse) * invB;
      if (k == y) p -= invB;
      dl[(size_t)b * K + k] = p;
    }
    if (tid == 0) lossrow[b] = lse - lg[y];
  }
  // do_pred
  if (!mf) return;   // criterion-only use (uniform per launch)
  float acc = 0.f;
  for (int m = tid; m < M; m += 256) acc += mf[(size_t)b * M + m] * wd[m];
  acc = wave_sum(acc);
  __syncthreads();
  if (l == 0) s_sum[w] = acc;
  __syncthreads();
  if (tid == 0) dopred[b] = sigmoidf_(((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3])) + bd[0]);
}
hipError_t ce_fwd(hipStream_t st, int nB, int K, int M, const float* logits,
                  const int32_t* labels, const float* mf, const float* wd, const float* bd,
                  float* dl, float* lossrow, int32_t* argmax, float* dopred, const float* part,
                  int nsplit, const float* bias, float* logits_out, int Bper) {
  if (!split_span_ok(part, nsplit, (size_t)nB * K)) return kSplitStateError;
  hipLaunchKernelGGL(k_ce_fwd, dim3(nB), dim3(256), 0, st, nB, K, M, logits, labels, mf, wd, bd,
                     dl, lossrow, argmax, dopred, part, nsplit, bias, logits_out,
                     Bper > 0 ? Bper : nB);
  return hipGetLastError();
}

__global__ void k_loss_reduce(int nB, const float* __restrict__ lossrow,
                              float* __restrict__ losses) {
  RAU_CHAIN_PRIO();
  const int h = blockIdx.x;
  float acc = 0.f;
  for (int b = threadIdx.x; b < nB; b += 64) acc += lossrow[(size_t)h * nB + b];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) losses[h] = acc / (float)nB;
}
hipError_t loss_reduce(hipStream_t st, int H, int nB, const float* lossrow, float* losses) {
  hipLaunchKernelGGL(k_loss_reduce, dim3(H), dim3(64), 0, st, nB, lossrow, losses);
  return hipGetLastError();
}

__global__ void k_scale_hops(size_t per_hop, size_t n, const float* __restrict__ w,
                             float* __restrict__ x) {
  RAU_CHAIN_PRIO();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] *= w[i / per_hop];
}
// the same for up to three hop-major tensors in one launch: x_i[h][..] *= w[h]
__global__ void k_scale_hops3(size_t p0, size_t p1, size_t p2, int H, const float* __restrict__ w,
                              float* __restrict__ x0, float* __restrict__ x1, float* __restrict__ x2) {
  RAU_CHAIN_PRIO();
  const size_t n0 = p0 * H, n1 = p1 * H, n2 = p2 * H;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n0 + n1 + n2;
       i += (size_t)gridDim.x * blockDim.x) {
    if (i < n0) x0[i] *= w[i / p0];
    else if (i < n0 + n1) x1[i - n0] *= w[(i - n0) / p1];
    else x2[i - n0 - n1] *= w[(i - n0 - n1) / p2];
  }
}
hipError_t scale_hops3(hipStream_t st, int H, const float* w_dev, size_t p0, float* x0, size_t p1, float* x1,
                       size_t p2, float* x2) {
  hipLaunchKernelGGL(k_scale_hops3, dim3(grid_for((p0 + p1 + p2) * H)), dim3(256), 0, st, p0, p1, p2, H, w_dev,
                     x0, x1, x2);
  return hipGetLastError();
}
hipError_t scale_hops(hipStream_t st, int H, size_t per_hop, const float* w_dev, float* x) {
  const size_t n = per_hop * H;
  hipLaunchKernelGGL(k_scale_hops, dim3(grid_for(n)), dim3(256), 0, st, per_hop, n, w_dev, x);
  return hipGetLastError();
}

// q[b] = [c1 h1 c2 h2] at t = lens[b] (state arrays are [T+1][nB][Rq], slot 0 = zeros)
__global__ void k_gather_q(int nB, int Rq, const int32_t* __restrict__ lens,
                           const float* __restrict__ c1, const float* __restrict__ h1,
                           const float* __restrict__ c2, const float* __restrict__ h2,
                           float* __restrict__ q) {
  RAU_CHAIN_PRIO();
  const size_t n = (size_t)nB * 4 * Rq;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / (4 * Rq));
    const int j = (int)(i - (size_t)b * 4 * Rq);
    const int slot = j / Rq, r = j - slot * Rq;
    const float* src = slot == 0 ? c1 : slot == 1 ? h1 : slot == 2 ? c2 : h2;
    const int t = lens[b];
    q[i] = t > 0 ? src[((size_t)t * nB + b) * Rq + r] : 0.f;
  }
}
hipError_t gather_q(hipStream_t st, int nB, int Rq, int T, const int32_t* lens, const float* c1,
                    const float* h1, const float* c2, const float* h2, float* q) {
  hipLaunchKernelGGL(k_gather_q, dim3(grid_for((size_t)nB * 4 * Rq)), dim3(256), 0, st, nB, Rq,
                     lens, c1, h1, c2, h2, q);
  return hipGetLastError();
}

// y[i] = x[i % period] * mask(i) * scale      (i over n; mask may be null)
__global__ void k_apply_mask(size_t n, size_t period, const float* __restrict__ x,
                             const uint32_t* __restrict__ mask, size_t e0, float mscale,
                             float* __restrict__ y) {
  RAU_CHAIN_PRIO();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i % period];
    y[i] = mask ? (mask_bit(mask, e0 + i) ? v * mscale : 0.f) : v;
  }
}
hipError_t apply_mask(hipStream_t st, size_t n, size_t period, const float* x,
                      const uint32_t* mask, float mscale, float* y, size_t mask_e0) {
  hipLaunchKernelGGL(k_apply_mask, dim3(grid_for(n)), dim3(256), 0, st, n, period, x, mask,
                     mask_e0, mscale, y);
  return hipGetLastError();
}

// ---- small helpers of the module-level entry points (rau_modules.hip)
// out[r, n] = s[r] * (X ? X[r, n] : 1) * (v ? v[n] : 1)
__global__ void k_row_scale(int rows, int N, const float* __restrict__ s,
                            const float* __restrict__ X, const float* __restrict__ v,
                            float* __restrict__ out) {
  const size_t n = (size_t)rows * N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / N), col = (int)(i - (size_t)r * N);
    float o = s[r];
    if (X) o *= X[i];
    if (v) o *= v[col];
    out[i] = o;
  }
}
hipError_t row_scale(hipStream_t st, int rows, int N, const float* s, const float* X,
                     const float* v, float* out) {
  hipLaunchKernelGGL(k_row_scale, dim3(grid_for((size_t)rows * N)), dim3(256), 0, st, rows, N, s,
                     X, v, out);
  return hipGetLastError();
}
// s[b] = ddp[b] * dp[b] (1 - dp[b])      (Sigmoid backward of out_do_pred, SS:281-282)
__global__ void k_sigmoid_bwd(int n, const float* __restrict__ dy, const float* __restrict__ y,
                              float* __restrict__ dx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = dy[i] * y[i] * (1.f - y[i]);
}
hipError_t sigmoid_bwd(hipStream_t st, int n, const float* dy, const float* y, float* dx) {
  hipLaunchKernelGGL(k_sigmoid_bwd, dim3((n + 255) / 256), dim3(256), 0, st, n, dy, y, dx);
  return hipGetLastError();
}
// out[i] = dI[i] * (1 - I[i]^2)   (Tanh backward of i_embed, materialised for the feature-map gradient)
__global__ void k_mul_dtanh(size_t n4, const float4* __restrict__ d, const float4* __restrict__ y,
                            float4* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = d[i], b = y[i];
    float4 o;
    o.x = a.x * (1.f - b.x * b.x);
    o.y = a.y * (1.f - b.y * b.y);
    o.z = a.z * (1.f - b.z * b.z);
    o.w = a.w * (1.f - b.w * b.w);
    out[i] = o;
  }
}
hipError_t mul_dtanh(hipStream_t st, size_t n, const float* d, const float* y, float* out) {
  hipLaunchKernelGGL(k_mul_dtanh, dim3(grid_for(n / 4)), dim3(256), 0, st, n / 4,
                     reinterpret_cast<const float4*>(d), reinterpret_cast<const float4*>(y),
                     reinterpret_cast<float4*>(out));
  return hipGetLastError();
}
__global__ void k_scale_inplace(size_t n, float s, float* __restrict__ x) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] *= s;
}
hipError_t scale_inplace(hipStream_t st, size_t n, float s, float* x) {
  hipLaunchKernelGGL(k_scale_inplace, dim3(grid_for(n)), dim3(256), 0, st, n, s, x);
  return hipGetLastError();
}

__global__ void k_dq_reduce(int H, size_t n, const float* __restrict__ dQD,
                            const uint32_t* __restrict__ mask, float mscale,
                            float* __restrict__ dq) {
  RAU_CHAIN_PRIO();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int h = H - 1; h >= 0; --h) {  // reference accumulates hop H first (SS:564)
      const size_t e = (size_t)h * n + i;
      const float v = dQD[e];
      acc += mask ? (mask_bit(mask, e) ? v * mscale : 0.f) : v;
    }
    dq[i] = acc;
  }
}
hipError_t dq_reduce(hipStream_t st, int H, size_t n, const float* dQD, const uint32_t* mask,
                     float mscale, float* dq) {
  hipLaunchKernelGGL(k_dq_reduce, dim3(grid_for(n)), dim3(256), 0, st, H, n, dQD, mask, mscale,
                     dq);
  return hipGetLastError();
}

__global__ void k_splitk_reduce_acc(size_t n4, int splits, const float4* __restrict__ slab,
                                    size_t stride4, float4* __restrict__ dst) {
  RAU_CHAIN_PRIO();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * blockDim.x) {
    float4 acc = slab[i];
    for (int s = 1; s < splits; ++s) {
      const float4 v = slab[(size_t)s * stride4 + i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4 d = dst[i];
    d.x += acc.x; d.y += acc.y; d.z += acc.z; d.w += acc.w;
    dst[i] = d;
  }
}
__global__ void k_splitk_reduce_acc1(size_t n, int splits, const float* __restrict__ slab,
                                     size_t stride, float* __restrict__ dst) {
  RAU_CHAIN_PRIO();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float acc = slab[i];
    for (int s = 1; s < splits; ++s) acc += slab[(size_t)s * stride + i];
    dst[i] += acc;
  }
}
hipError_t splitk_reduce_acc(hipStream_t st, size_t n, int splits, const float* slab,
                             size_t slab_stride, float* dst) {
  // split 0 is read unconditionally: at least one partial, each of n floats at pitch slab_stride
  if (splits < 1 || slab_stride < n || !split_span_ok(slab, splits, slab_stride)) return kSplitStateError;
  const bool v4 = (n % 4 == 0) && (slab_stride % 4 == 0) && (((uintptr_t)dst & 15) == 0) &&
                  (((uintptr_t)slab & 15) == 0);
  if (v4)
    hipLaunchKernelGGL(k_splitk_reduce_acc, dim3(grid_for(n / 4)), dim3(256), 0, st, n / 4, splits,
                       reinterpret_cast<const float4*>(slab), slab_stride / 4,
                       reinterpret_cast<float4*>(dst));
  else
    hipLaunchKernelGGL(k_splitk_reduce_acc1, dim3(grid_for(n)), dim3(256), 0, st, n, splits, slab,
                       slab_stride, dst);
  return hipGetLastError();
}

// Fused split-K reduction + Linear epilogue: C[m,n] = epi(sum_s slab[s][m,n]).
// Partials are summed in split order, so the result is bitwise reproducible.
__global__ void k_lin_reduce_epilogue(int M, int N, int splits, const float* __restrict__ slab,
                                      float* __restrict__ C, long ldc, LinOpts o) {
  RAU_CHAIN_PRIO();
  const size_t total = (size_t)M * N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
    float v = slab[i];
    for (int s = 1; s < splits; ++s) v += slab[(size_t)s * total + i];
    v *= o.alpha;
    if (o.bias) v += o.bias[n];
    if (o.bias2) v += o.bias2[n];
    if (o.addend) v += o.addend[(long)m * o.add_rs + n];
    const long ci = (long)m * ldc + n;
    if (o.accumulate) v += C[ci];
    if (o.act == 1) v = tanh_fast(v);
    if (o.ymul) {
      const float y = o.ymul[(long)m * o.y_rs + n];
      v *= (1.f - y * y);
    }
    if (o.emask) v = mask_bit(o.emask, o.emask_e0 + i) ? v * o.emscale : 0.f;
    C[ci] = v;
  }
}
hipError_t lin_reduce_epilogue(hipStream_t st, int M, int N, int splits, const float* slab,
                               float* C, long ldc, const LinOpts& o) {
  if (splits < 1 || !split_span_ok(slab, splits, (size_t)M * N)) return kSplitStateError;
  hipLaunchKernelGGL(k_lin_reduce_epilogue, dim3(grid_for((size_t)M * N)), dim3(256), 0, st, M, N,
                     splits, slab, C, ldc, o);
  return hipGetLastError();
}

// --------------------------------------------------- noise / clip / Adam (next-1)
__device__ __forceinline__ float2 box_muller(uint32_t a, uint32_t b) {
  const float u1 = ((a >> 8) + 1) * (1.0f / 16777216.0f);  // (0,1]
  const float u2 = (b >> 8) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.f * logf(u1));
  float s, c;
  sincosf(6.28318530717958647692f * u2, &s, &c);
  return make_float2(r * c, r * s);
}
__global__ void k_add_noise_sqnorm(size_t n, float* __restrict__ g, float nstd, uint64_t seed,
                                   uint32_t stream, float* __restrict__ partial) {
  RAU_CHAIN_PRIO();
  __shared__ float s_sum[4];
  float acc = 0.f;
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q * 4 < n;
       q += (size_t)gridDim.x * blockDim.x) {
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (nstd > 0.f) {
      const Philox4 o = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), stream, 0xA5A5u,
                                      (uint32_t)seed, (uint32_t)(seed >> 32));
      const float2 n0 = box_muller(o.v[0], o.v[1]), n1 = box_muller(o.v[2], o.v[3]);
      z[0] = n0.x; z[1] = n0.y; z[2] = n1.x; z[3] = n1.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < n) {
        const float v = g[q * 4 + j] + z[j] * nstd;
        g[q * 4 + j] = v;
        acc += v * v;
      }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
}
hipError_t add_noise_sqnorm(hipStream_t st, size_t n, float* g, float nstd, uint64_t seed,
                            uint32_t stream, float* partial) {
  hipLaunchKernelGGL(k_add_noise_sqnorm, dim3(1024), dim3(256), 0, st, n, g, nstd, seed, stream,
                     partial);
  return hipGetLastError();
}
__global__ void k_finish_norm(int nparts, const float* __restrict__ partial,
                              float* __restrict__ norm_out) {
  RAU_CHAIN_PRIO();
  __shared__ float s_sum[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) norm_out[0] = sqrtf((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
}
hipError_t finish_norm(hipStream_t st, int nparts, const float* partial, float* norm_out) {
  hipLaunchKernelGGL(k_finish_norm, dim3(1), dim3(256), 0, st, nparts, partial, norm_out);
  return hipGetLastError();
}
__global__ void k_clip_adam(size_t n, float* __restrict__ x, float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ norm, float clip, float stepsize,
                            float beta1, float beta2, float eps) {
  RAU_CHAIN_PRIO();
  const float nv = norm[0];
  const float sc = nv > clip ? clip / nv : 1.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * sc;
    g[i] = gi;
    const float mi = m[i] * beta1 + (1.f - beta1) * gi;
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    x[i] -= stepsize * mi / (sqrtf(vi) + eps);
  }
}
hipError_t clip_adam(hipStream_t st, size_t n, float* x, float* g, float* m, float* v,
                     const float* norm, float clip, float stepsize, float beta1, float beta2,
                     float eps) {
  hipLaunchKernelGGL(k_clip_adam, dim3(grid_for(n)), dim3(256), 0, st, n, x, g, m, v, norm, clip,
                     stepsize, beta1, beta2, eps);
  return hipGetLastError();
}

}  // namespace rau
