// kernels.h -- host-callable launchers of the HIP kernels (internal to librau.so).
// Every launcher enqueues on `st` and returns hipGetLastError().
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rau {

// ------------------------------------------------ split-K partials: fail closed (split_guard.hip)
// Consumers of K-split partials read slab[split][..] for split < nsplit with no bound of their own.
// Their launchers check the span first: inside ONE registered workspace, count in [1, kMaxSplits]
// (0 = no partials, always fine); on violation they launch nothing and return kSplitStateError.
constexpr long kMaxSplits = 4096;
constexpr hipError_t kSplitStateError = hipErrorIllegalState;
void split_ws_register(const float* base, size_t floats);
void split_ws_unregister(const float* base);
bool split_span_ok(const float* slab, long nsplit, size_t per_split_floats);

// ------------------------------------------------------------ GEMM (gemm_lin.hip)
struct LinOpts {
  const float* bias = nullptr;    // + bias[n]
  const float* bias2 = nullptr;   // + bias2[n]
  const float* addend = nullptr;  // + addend[m*add_rs + n]
  long add_rs = 0;
  int accumulate = 0;             // + C_old
  int act = 0;                    // 1: tanh
  const float* ymul = nullptr;    // * (1 - y^2)
  long y_rs = 0;
  const uint32_t* emask = nullptr;  // dropout on the output (bit index e0 + m*N + n)
  size_t emask_e0 = 0;
  float emscale = 1.f;
  float alpha = 1.f;
  // split-K workspace: when given, skinny problems are split over K across
  // workgroups (partials to slab, then one fused reduce + epilogue kernel)
  float* slab = nullptr;
  size_t slab_floats = 0;
  // deferred reduction: leave the K-split partials in `slab` ([splits][M*N]) and
  // report the split count; the consumer kernel (lstm_fwd / lstm_bwd) sums them
  int* defer_splits = nullptr;
};
// C[M,N] = epi(A[M,K] * W[N,K]^T)      (Linear forward)
hipError_t gemm_nt(hipStream_t st, int M, int N, int K, const float* A, long lda,
                   const float* W, long ldw, float* C, long ldc, const LinOpts& o);
// C[M,N] = epi(A[M,K] * W[K,N])        (Linear input gradient)
hipError_t gemm_nn(hipStream_t st, int M, int N, int K, const float* A, long lda,
                   const float* W, long ldw, float* C, long ldc, const LinOpts& o);
// nb (<= 3) same-shape problems C_i = A_i W_i^T (nt) / A_i W_i (nn) in one launch; the
// K-split partials stay in `slab` as [problem][split][M*N] (*splits per problem) for a
// consumer kernel to sum -- used by the encoder's layer wavefront.
hipError_t gemm_nt_batched_deferred(hipStream_t st, int nb, int M, int N, int K,
                                    const float* const* A, long lda, const float* const* W,
                                    long ldw, float* slab, size_t slab_floats, int* splits);
hipError_t gemm_nn_batched_deferred(hipStream_t st, int nb, int M, int N, int K,
                                    const float* const* A, long lda, const float* const* W,
                                    long ldw, float* slab, size_t slab_floats, int* splits);
// nb (<= 3) Linear forwards x W_i^T that share x (and K) but differ in width, one launch;
// problem i's partials at slab + off[i] as [*splits][M][N[i]]
hipError_t gemm_nt_hetero_deferred(hipStream_t st, int nb, int M, int K, const float* A, long lda,
                                   const float* const* W, long ldw, const int* N, float* slab,
                                   size_t slab_floats, int* splits, size_t* off);
// C[M,N] += A[K,M]^T * B[K,N]          (Linear weight gradient; deterministic split-K
// through `slab`, which must hold gemm_tn_slab_floats(M,N,K) floats)
size_t gemm_tn_slab_floats(int M, int N, int K);
// dbias (optional): dbias[m] += sum_k A[k, m], from the same pass over A (Linear bias gradient)
// bf16 != 0 (RAU_BF16 mode; BASELINE.json configs[2]: "bf16 MFMA gate/classifier GEMMs"): both operands
// rounded to bf16 (RNE) while staged, f32 accumulate; dbias is summed from the UNROUNDED values
hipError_t gemm_tn_acc(hipStream_t st, int M, int N, int K, const float* A, long lda,
                       const float* B, long ldb, float* C, long ldc, float* slab,
                       float* dbias = nullptr, int bf16 = 0);

// All the Linear weight gradients of a parameter group as ONE grouped launch (+ one reduction):
// C_p[M_p,N_p] += A_p[K,M_p]^T B_p[K,N_p], dbias_p[m] += sum_k A_p[k,m]; C_p dense (ldc = N_p).
struct TnProblem {
  int M, N; const float* A; long lda; const float* B; long ldb; float* C; float* dbias;
};
size_t gemm_tn_group_slab_floats(const TnProblem* pr, int np, int K);
hipError_t gemm_tn_group_acc(hipStream_t st, const TnProblem* pr, int np, int K, float* slab,
                             size_t slab_floats, int bf16 = 0);

// ----------------------------------------------------------- conv GEMMs (gemm_conv.hip)
// I[b,m,s] = tanh(sum_d Wi[m,d] * X'[b,d,s] + bi[m])   (reference SS:238-242; X' is the
// feature map with dropout already applied, see dropout_features; nB may be H*B)
// bf16 != 0 (rau_dtype RAU_BF16) on the five hop-batched conv GEMMs: operands rounded to bf16
// while staged into LDS, f32 accumulate (v_mfma_f32_32x32x16_bf16); tensors in HBM stay f32.
hipError_t conv_embed_fwd(hipStream_t st, int nB, int D, int S, int M, const float* X,
                          const float* WiT /* [D][M] */, const float* bi, float* I, int bf16 = 0,
                          int one_per_cu = 0 /* cap residency at one workgroup per CU */);
// RAU_BF16 mode with the feature maps stored as bf16 (X16 [nB][D][S] bf16; S % 4 == 0)
hipError_t conv_embed_fwd_b16(hipStream_t st, int nB, int D, int S, int M, const void* X16,
                              const void* WiT16, const float* bi, float* I);
// ifeatproj in RAU_BF16 mode with the transposed weight pre-converted (WpT16 [M][A] bf16)
hipError_t conv_att_pre_b16(hipStream_t st, int nB, int M, int S, int A, const float* I,
                            const void* WpT16, const float* bp, float* Pout);
// P[b,k,s] = sum_m Wp[k,m] I[b,m,s] + bp[k]   (hop-invariant half of SS:244-252)
hipError_t conv_att_pre(hipStream_t st, int nB, int M, int S, int A, const float* I,
                        const float* WpT /* [M][A] */, const float* bp, float* P, int bf16 = 0,
                        int one_per_cu = 0);
// Same three products with one sample per tile (gemm_sample.hip), used for 14 x 14 maps:
// C[b,m,s] = epi(sum_k Wt[k,m] X[b,k,s]); epi 0: act(. + bias[m]), epi 1: . + dj[b,m] a[b,s]
bool conv_sample_ok(int S, int which);
hipError_t conv_sample(hipStream_t st, int epi, int nB, int M, int K, int S, const float* Wt,
                       long w_rs, const float* X, long x_bs, float* C, long c_bs,
                       const float* bias, int act, const float* dj, const float* av,
                       const float* Y = nullptr, float* rs = nullptr, int c16 = 0);
// Wide tiling for 14 x 14 maps (conv_wide.hip, round 3): 64 rows x four whole samples per tile,
// operands staged by LDS-DMA; epi 0 / 2 as conv_sample's.  nB must be a multiple of 4.
bool conv_wide_ok(int M, int K, int S, long w_rs);
hipError_t conv_wide(hipStream_t st, int epi, int nB, int M, int K, int S, const float* Wt, long w_rs,
                     const float* X, long x_bs, float* C, long c_bs, const float* bias, int act,
                     const float* dj, const float* av, const float* Y, float* rs, int c16, int per_cu);
// out[c][r] = in[r][c]  (rows x cols -> cols x rows); used once per step on the two
// 1x1-conv weights so the forward conv GEMMs get a row-contiguous A operand
hipError_t transpose2d(hipStream_t st, int rows, int cols, const float* in, float* out,
                       void* out16 = nullptr);   // out16: a bf16 copy of out (RAU_BF16 mode)
// dI[b,m,s] = sum_k Wp[k,m] dS[b,k,s] + dj[b,m] a[b,s]   (gradient at i_embed's output)
hipError_t conv_att_dgrad(hipStream_t st, int nB, int M, int S, int A, const float* dS,
                          const float* Wp, const float* dj, const float* a, float* dI,
                          int bf16 = 0);
// Same product with the gradient through i_embed's tanh and the bias-gradient row sums in its
// epilogue: dZ[b,m,s] = (sum_k Wp[k,m] dS[b,k,s] + dj[b,m] a[b,s]) (1 - I[b,m,s]^2),
// rs[b,m] = sum_s dZ[b,m,s].  Only where the per-sample tiling applies (conv_dz_fused_ok): the
// i_embed weight gradient then takes dZ with its plain operand loader (dz_final below).
bool conv_dz_fused_ok(int S, int M, int bf16);
hipError_t conv_att_dgrad_dz(hipStream_t st, int nB, int M, int S, int A, const float* dS,
                             const float* Wp, const float* dj, const float* a, const float* I,
                             float* dZ, float* rs, int dz16 = 0,    // dz16: dZ stored as bf16
                             int bf16 = 0,    // RAU_BF16 mode ...
                             int ds16 = 0,    // ... and dS points at bf16 elements (dgrad16 only)
                             int light = 0);  // the caller's recurrence is the longer path: keep the f32 product
                                              // on the 176-register tile instead of dgrad_dma's 240 (two per CU)
// RAU_BF16 mode, 14 x 14 maps, M % 128 == 0, K % 32 == 0 (dgrad16.hip): the product above with both GEMM
// operands rounded to bf16 while staged, f32 accumulate and epilogue; C = dZ as f32 or bf16 elements
bool dgrad16_ok(int M, int K, int S, long w_rs);
hipError_t dgrad16(hipStream_t st, int nB, int M, int K, int S, const float* Wt, long w_rs,
                   const void* X, long x_bs, void* C, long c_bs, const float* dj, const float* av,
                   const float* Y, float* rs, int c16, int x16 = 0 /* X stored as bf16 */);
// f32 mode, 14 x 14 maps, M % 128 == 0, K % 16 == 0 (dgrad_dma.hip, round 4): the same product and epilogue on
// per-sample tiles with both operands staged by LDS-DMA; bitwise the results of conv_sample's EPI 2
bool dgrad_dma_ok(int M, int K, int S, long w_rs);
hipError_t dgrad_dma(hipStream_t st, int nB, int M, int K, int S, const float* Wt, long w_rs,
                     const float* X, long x_bs, float* C, long c_bs, const float* dj, const float* av,
                     const float* Y, float* rs);
// dWp += sum dS I^T in RAU_BF16 mode with dS stored as bf16 (att_bwd_fused's dS16), I f32
hipError_t conv_att_wgrad_ds16(hipStream_t st, int nB, int M, int S, int A, const void* dS16,
                               const float* I, float* dWp, float* slab);
// dX'[b,d,s] = sum_m Wi[m,d] dZ[b,m,s]   (dead in feval, SS:579; module-level API only)
hipError_t conv_embed_dgrad(hipStream_t st, int nB, int D, int S, int M, const float* dZ,
                            const float* Wi, float* dX);
// dWp[k,m] += sum_{b,s} dS[b,k,s] I[b,m,s]   and, with dZ = dI (1 - I^2),
// dWi[m,d] += sum_{b,s} dZ[b,m,s] X'[b,d,s]
size_t conv_wgrad_slab_floats(int nB, int rowsA, int rowsB, int S);
// bf16-operand i_embed weight gradient on 14 x 14 maps, 256 x 256 tiles fed by LDS-DMA
// (wgrad16.hip, round 3): dW[ra][rb] += sum_{b,s} A16[b][ra][s] B16[b][rb][s], operands bf16 in HBM
// (a_bs / b_bs in elements).  _ok: S == 196, ra and rb multiples of 256; RAU_WGRAD16_OFF=1 keeps the
// 128 x 128 register-staged tile (A/B knob).  The slab must hold wgrad16_slab_floats().
bool wgrad16_ok(int ra, int rb, int S);
size_t wgrad16_slab_floats(int nB, int ra, int rb, int S);
hipError_t wgrad16(hipStream_t st, int nB, int ra, int rb, int S, const void* A16, long a_bs,
                   const void* B16, long b_bs, float* dW, float* slab);
// Split-K partials of the recurrence's skinny Linear GEMMs with LDS-DMA operand staging
// (skinny_dma.hip, round 3): nprob (<= 3) problems C_p = A_p W_p^T (brc = false, W [N][K]) or
// A_p W_p (brc = true, W [K][N]) of one M and K; problem p's partials at slab + off[p] laid out
// [split][M][N[p]].  _ok: K % 32 == 0, 16-byte aligned rows, [K][N] weights a multiple of 64 wide;
// RAU_SKINNY_DMA_OFF=1 keeps the register-staged tile of gemm_core.h (A/B knob).
bool skinny_dma_ok(int M, int K, long lda, long ldb, bool brc, int nprob, const int* N,
                   const float* const* A, const float* const* B);
int skinny_dma_splits(int M, int K, int tiles_all, size_t cols_all, size_t slab_floats);
// 16-deep or 32-deep stages (one kernel template; 32-deep needs K % 64 == 0): chosen for the calling
// thread by skinny_dma_set_deep, which the step-level entry points set from chain_bound() (rau_ctx.h)
void skinny_dma_set_deep(int on);
// rau_dtype RAU_BF16 (BASELINE.json configs[2]: "bf16 MFMA gate/classifier GEMMs"): every Linear product
// of the calling thread's launches -- forward y = x W^T and input gradient dx = dy W through gemm_nt /
// gemm_nn and their batched forms, whichever kernel serves them -- rounds both operands to bf16 (round to
// nearest even) and accumulates in f32.  skinny_dma.hip does it with bf16 MFMAs; the register-staged
// fallback tiles round while staging and keep the f32 MFMA (same products, another summation order).
// Set at every step-level entry point from the ctx's dtype (set_skinny_policy, rau_ctx.h).
void lin_set_bf16(int on);
int lin_bf16();
hipError_t skinny_dma(hipStream_t st, bool brc, int nprob, int M, int K, const float* const* A,
                      long lda, const float* const* B, long ldb, const int* N, float* slab,
                      const long* off, int splits);
// the same products with both operands staged by LDS-DMA (wgrad_dma.hip, round 3): 14 x 14 maps,
// row counts multiples of 128, plain operands (dZ already final)
bool wgrad_dma_ok(int ra, int rb, int S);
hipError_t wgrad_dma(hipStream_t st, int nB, int ra, int rb, int S, const float* A, long a_bs,
                     const float* B, long b_bs, float* dW, float* slab, int splits);
hipError_t conv_att_wgrad(hipStream_t st, int nB, int M, int S, int A, const float* dS,
                          const float* I, float* dWp, float* slab, int bf16 = 0);
hipError_t conv_embed_wgrad(hipStream_t st, int nB, int D, int S, int M, const float* dI,
                            const float* I, const float* X, float* dWi, float* slab,
                            int bf16 = 0, float* dbi = nullptr /* += sum_{b,s} dZ[b,m,s] */,
                            int dz_final = 0 /* dI already holds dZ: no tanh factor, no dbi */);
// same with dZ already final and both operands stored as bf16 (conv_att_dgrad_dz with dz16)
hipError_t conv_embed_wgrad_b16(hipStream_t st, int nB, int D, int S, int M, const void* dZ16,
                                const void* X16, float* dWi, float* slab);

// --------------------------------------------------------- pointwise (kernels.hip)
enum GateOrder { GATES_ATT = 0 /* i g f o, ATTLSTM.lua:12-19 */,
                 GATES_DEEP = 1 /* i f o g, DeepLSTM.lua:46-54 */ };

// key_dev != nullptr: (seed, step) are read from key_dev[0], key_dev[1] on the device instead
hipError_t fill_masks(hipStream_t st, uint64_t seed, uint32_t site, uint32_t step, float p,
                      size_t n, uint32_t* bits, const uint64_t* key_dev = nullptr);
hipError_t embed_fwd(hipStream_t st, int rows, int E, int V, const float* emb, const int32_t* tokens,
                     const uint32_t* mask, float mscale, float* we, size_t mask_e0 = 0);
hipError_t embed_bwd_rows(hipStream_t st, int rows, int E, int V, const int32_t* tokens, const float* dwe,
                          const float* we, const uint32_t* mask, size_t mask_e0, float mscale,
                          float* gE);
// in place on g4: pre-activations (+ sum of `nsplit` split-K partials [nB,4R] in `slab`)
// -> activated gates; c = f c_prev + i g; h = o tanh c
hipError_t lstm_fwd(hipStream_t st, int order, int nB, int R, float* g4, const float* c_prev,
                    long cp_rs, float* c, long c_rs, float* h, long h_rs, float* tanhc,
                    float* drop_out, const uint32_t* mask, size_t mask_e0, float mscale,
                    const float* slab = nullptr, int nsplit = 0);
// dsum from (dh [+dh2], dc_next), dh optionally given as `nsplit` split-K partials [nB,R]
// in `slab`; optional row replacement by dq rows where lens[b]==t
hipError_t lstm_bwd(hipStream_t st, int order, int nB, int R, const float* gates,
                    const float* c_prev, long cp_rs, const float* tanhc, const float* dh,
                    long dh_rs, const float* dh2, const float* dc_next, float* dsum,
                    float* dc_prev, const int32_t* lens, int t, const float* dq_c,
                    const float* dq_h, long dq_rs, const float* slab = nullptr, int nsplit = 0);
// Up to two independent LSTM cells in one launch (blockIdx.y): the encoder's layer
// wavefront evaluates layer-1 cell t+1 and layer-2 cell t together.
struct LstmFwdCell {
  float* g4;                      // [nB,4R] out: activated gates; in: input part if has_input
  int has_input;                  // 1: pre-activation starts from g4, 0: from b1 + b2
  const float* b1; const float* b2;
  const float* slab; int nsplit;  // nsplit partials [nB,4R], contiguous, summed in order
  const float* c_prev; long cp_rs;
  float* c; long c_rs; float* h; long h_rs; float* tanhc;
  float* drop_out; const uint32_t* mask; size_t mask_e0; float mscale;
};
struct LstmFwdCells { int n; LstmFwdCell c[2]; };
hipError_t lstm_fwd_multi(hipStream_t st, int order, int nB, int R, const LstmFwdCells& cells);
// Weight-stationary persistent encoder forward (enc_ws.hip, round 3): all T token steps of both
// layers in one launch, recurrent weights held in registers, per-(layer, sample half) counters
// instead of a grid barrier.  cnt: 16 words (zeroed by the launcher), err: device error word.
struct EncWsParams {
  int B, R, TL, P;
  float *G1, *G2, *h1, *c1, *tc1, *x2, *h2, *c2, *tc2;
  const float *Wh1, *Wi2, *Wh2, *bi2, *bh2;
  const uint32_t* mask; float mscale;
  unsigned* cnt; int* err;
  int bf16 = 0;   // operands of the gate products rounded to bf16
};
bool enc_ws_ok(int B, int R);
int enc_ws_workgroups(int B);
// All workgroups of a launch must be co-resident (they wait on each other's progress counters):
// the pure predicate, and the same question asked of the current device (occupancy query).
bool enc_ws_coresident(int B, int blocks_per_cu, int n_cus);
bool enc_ws_fits_device(int B);
hipError_t enc_ws_forward(hipStream_t st, int order, EncWsParams Q);
struct LstmBwdCell {
  const float* gates; const float* c_prev; long cp_rs; const float* tanhc;
  const float* slabA; int nA;     // recurrent dh partials [nB,R]
  const float* slabB; int nBp;    // extra dh partials, passed through a dropout mask
  const uint32_t* maskB; size_t maskB_e0; float mscaleB;
  const float* dc_next; float* dsum; float* dc_prev;
  int t; const float* dq_c; const float* dq_h;   // rows with lens[b]==t take dq (SS:584-591)
};
struct LstmBwdCells { int n; const int32_t* lens; long dq_rs; LstmBwdCell c[2]; };
hipError_t lstm_bwd_multi(hipStream_t st, int order, int nB, int R, const LstmBwdCells& cells);
// One workgroup per sample: T = tanh(P + u[b,:,None]) (attbycontent, SS:250),
// e = ws . T + bs (SS:251), a = softmax(e + zm) (attbymemory, SS:288-289),
// jv = qf + sum_s I a (attselect SS:254-263 + first CAddTable SS:270).
// ap: u and/or zm may be handed over as K-split GEMM partials ([split][nB][A] / [split][nB][S],
// passed in the u / zm arguments) plus the Linear's bias, summed in order by the kernel itself --
// two reduce launches less on the recurrence's critical path.
// SL: logical number of positions when the tensors' pitch S is padded (7x7 maps); positions
// [SL, S) get zero attention.
// u_out: where to keep the finished u rows [nB][A] when T is not stored (T == nullptr).
// waves: the caller's choice of waves per sample for the forward kernel (8 | 16; 0 = the default, 8)
struct AttPartials { int u_ns = 0; const float* u_bias = nullptr; int z_ns = 0; const float* z_bias = nullptr; int SL = 0; float* u_out = nullptr; int waves = 0; };
hipError_t att_fwd_fused(hipStream_t st, int nB, int M, int A, int S, const float* P,
                         const float* u, const float* ws, const float* bs, const float* zm,
                         const float* I, const float* qf, float* T, float* a, float* jv,
                         const AttPartials& ap = AttPartials());
// One workgroup per sample, backward of the above: da = da_lin + sum_m dj I;
// dz = softmax'(da); T -> dS = dz ws (1-T^2) in place; du = sum_s dS; dwsp = sum_s dz T.
hipError_t att_bwd_fused(hipStream_t st, int nB, int M, int A, int S, const float* I,
                         const float* dj, const float* a, const float* da_lin,
                         const float* ws, float* T_to_dS, float* dz, float* du, float* dwsp,
                         const float* Psrc = nullptr /* T not kept: recompute tanh(Psrc + u) */,
                         const float* u = nullptr,
                         // da_ns > 0: da_lin holds K-split partials [split][nB][SL] of dj Wf (summed
                         // by the kernel, plus the optional pitched addend da_add [nB][S])
                         int da_ns = 0, int SL = 0, const float* da_add = nullptr,
                         // dS16 != nullptr (only where att_bwd_dma_ok): dS leaves as bf16 [nB][A][S] there
                         // and T_to_dS keeps P
                         void* dS16 = nullptr,
                         int waves_hint = 0 /* 8 | 16 waves per workgroup where RAU_ATT_WAVES_BWD is unset */);
bool att_bwd_dma_ok(int M, int A, int S, int waves_hint = 0);
// The same two passes cut into 4-wave workgroups (NC row chunks per sample, two launches per
// pass): they always fit next to resident bulk-GEMM workgroups.  `part` is scratch of
// att_split_part_floats(nB, S) floats.  T is never kept (the backward recomputes tanh(Psrc + u)
// when Psrc is given, else reads T).
size_t att_split_part_floats(int nB, int S);
hipError_t att_fwd_split(hipStream_t st, int nB, int M, int A, int S, const float* P,
                         const float* u, const float* ws, const float* bs, const float* zm,
                         const float* I, const float* qf, float* a, float* jv, float* part,
                         const AttPartials& ap = AttPartials());
hipError_t att_bwd_split(hipStream_t st, int nB, int M, int A, int S, const float* I,
                         const float* dj, const float* a, const float* da_lin, const float* ws,
                         float* T_to_dS, float* dz, float* du, float* dwsp, const float* Psrc,
                         const float* u, float* part, int da_ns = 0, int SL = 0,
                         const float* da_add = nullptr);
// xd[h][i] = X[i] * keep(h, i) * scale for h < H, i < per_hop (feature-map dropout, SS:239)
// SL != Sp: rows of SL logical positions at pitch Sp (mask indexed logically, pad columns zeroed)
hipError_t dropout_features(hipStream_t st, int H, size_t per_hop, const float* X,
                            const uint32_t* mask, float mscale, float* xd, size_t mask_e0 = 0,
                            int SL = 0, int Sp = 0);
// The same pass (f32 or, b16 != 0, bf16 output) drawing the site's keep bits itself instead of reading
// them: per_hop % 16 == 0, logical == physical layout; bit-identical to fill_masks + dropout_features
hipError_t dropout_features_gen(hipStream_t st, uint64_t seed, uint32_t site, uint32_t step, float p,
                                const uint64_t* key_dev, int H, size_t per_hop, const float* X,
                                float mscale, void* xd, int b16);
// RAU_BF16 mode (S % 4 == 0): the same values stored as bf16, [h][i], the form conv_embed_fwd_b16 /
// conv_embed_wgrad_b16 read
hipError_t dropout_features_b16(hipStream_t st, int H, size_t per_hop, const float* X,
                                const uint32_t* mask, float mscale, void* xd16);
// dst[n] += sum_rows X[row*ld + n]   (two-stage, deterministic; tmp >= 32*N floats)
hipError_t colsum_acc(hipStream_t st, int rows, int N, const float* X, long ld, float* dst,
                      float* tmp);
// per-row CE: argmax (1-based, first max), loss row, dl = (softmax - onehot)/nB, do_pred
// (labels == nullptr: no loss/dl; mf == nullptr: no do_pred)
// nsplit > 0: the logits arrive as K-split partials `part` [split][nB][K] (+ bias) and are
// finished here into logits_out
hipError_t ce_fwd(hipStream_t st, int nB, int K, int M, const float* logits,
                  const int32_t* labels, const float* mf, const float* wd, const float* bd,
                  float* dl, float* lossrow, int32_t* argmax, float* dopred,
                  const float* part = nullptr, int nsplit = 0, const float* bias = nullptr,
                  float* logits_out = nullptr,
                  int Bper = 0 /* rows are [hop][sample]: samples per hop (label period, 1/B) */);
hipError_t loss_reduce(hipStream_t st, int H, int nB, const float* lossrow, float* losses);
hipError_t scale_hops(hipStream_t st, int H, size_t per_hop, const float* w_dev, float* x);
// x_i[h][0 .. p_i) *= w[h] for three hop-major tensors in one launch
hipError_t scale_hops3(hipStream_t st, int H, const float* w_dev, size_t p0, float* x0, size_t p1, float* x1,
                       size_t p2, float* x2);
hipError_t gather_q(hipStream_t st, int nB, int Rq, int T, const int32_t* lens, const float* c1,
                    const float* h1, const float* c2, const float* h2, float* q);
hipError_t apply_mask(hipStream_t st, size_t n, size_t period, const float* x,
                      const uint32_t* mask, float mscale, float* y, size_t mask_e0 = 0);
// module-level helpers: out[r,n] = s[r] (X ? X[r,n] : 1) (v ? v[n] : 1); sigmoid / tanh backward
hipError_t row_scale(hipStream_t st, int rows, int N, const float* s, const float* X,
                     const float* v, float* out);
hipError_t sigmoid_bwd(hipStream_t st, int n, const float* dy, const float* y, float* dx);
hipError_t mul_dtanh(hipStream_t st, size_t n, const float* d, const float* y, float* out);
hipError_t scale_inplace(hipStream_t st, size_t n, float s, float* x);
// dq[i] = sum_h dQD[h][i] * mask_h[i] * scale
hipError_t dq_reduce(hipStream_t st, int H, size_t n, const float* dQD, const uint32_t* mask,
                     float mscale, float* dq);
// dx2 -> dh1 extra: y = x * mask * scale (mask may be null)
hipError_t embed_bwd(hipStream_t st, int nuniq, int E, const int32_t* utok, const int32_t* ustart,
                     const int32_t* upos, const float* dwe, const float* we,
                     const uint32_t* mask, float mscale, float* gE);
hipError_t splitk_reduce_acc(hipStream_t st, size_t n, int splits, const float* slab,
                             size_t slab_stride, float* dst);
// C = epilogue(sum_s slab[s]) with the LinOpts epilogue (bias, addend, tanh, ...)
hipError_t lin_reduce_epilogue(hipStream_t st, int M, int N, int splits, const float* slab,
                               float* C, long ldc, const LinOpts& o);
hipError_t uniform_fill(hipStream_t st, uint64_t seed, uint32_t stream, size_t n, float lo,
                        float hi, float* x);
// noise + norm + clip + adam (SS:597-630, optim_updates.lua:59-87)
hipError_t add_noise_sqnorm(hipStream_t st, size_t n, float* g, float nstd, uint64_t seed,
                            uint32_t stream, float* partial /* >= 1024 */);
hipError_t finish_norm(hipStream_t st, int nparts, const float* partial, float* norm_out);
hipError_t clip_adam(hipStream_t st, size_t n, float* x, float* g, float* m, float* v,
                     const float* norm, float clip, float stepsize, float beta1, float beta2,
                     float eps);

}  // namespace rau
