// rau_modules.hip -- module-level entry points of include/rau.h: one call per
// nn.Module :forward / :backward of the reference's clones, so that the loops of
// feval (experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua:443-596) can run
// unchanged above the C ABI:
//   embed_clones[t]   word_embed   SS:203-206   -> rau_embed_forward / _backward
//   lstm_clones[t]    DeepLSTM     SS:213       -> rau_deeplstm_forward / _backward
//   multimodal_clones[h]           SS:292-307   -> rau_multimodal_forward / _backward
//   criteria[h]       CrossEntropyCriterion SS:310 -> rau_criterion_forward / _backward
// Same kernels as the step-level path (rau_forward / rau_backward), but executed at the
// reference's granularity on the ctx stream alone: nothing is hoisted across clones, every
// weight gradient is accumulated by the clone's own :backward (accGradParameters).  The
// step-level path is the fast one; this one exists for drop-in compatibility and as a
// second, independently scheduled route to the same numbers (tests/test_gpu_modules.py).
#include "rau_ctx.h"

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int mod_alloc(rau_ctx* ctx) {
  if (ctx->mod_ready) return 0;
  // Callers hand in / get back DENSE [.., S] tensors.  Internally every [.., S] tensor uses the
  // position pitch Sp (S rounded up to a multiple of 4: 7x7 maps 49 -> 52), so when Sp != S the
  // feature map / d_attprob are re-pitched on the way in and attprob / d_X on the way out.
  const rau_config& c = ctx->cfg;
  const size_t B = c.B, Q = ctx->Q;
  const size_t wide = std::max<size_t>({(size_t)c.Rq, (size_t)c.R, (size_t)c.M});
#define CK(x) do { if (int rc_ = (x)) return rc_; } while (0)
  CK(dalloc(ctx, &ctx->m_state, (size_t)c.T * B * Q));
  CK(dalloc(ctx, &ctx->m_dstate, (size_t)c.T * B * Q));
  for (int i = 0; i < 4; ++i) CK(dalloc(ctx, &ctx->m_tmp[i], B * wide));
  CK(dalloc(ctx, &ctx->m_dq, (size_t)c.H * B * Q));
  CK(dalloc(ctx, &ctx->m_dc, (size_t)c.H * B * c.R));
  CK(dalloc(ctx, &ctx->m_dh, (size_t)c.H * B * c.R));
  CK(dalloc(ctx, &ctx->m_add, B * c.M));
  CK(dalloc(ctx, &ctx->m_s, B));
  CK(dalloc(ctx, &ctx->m_zero, B * std::max<size_t>(Q, (size_t)c.R)));
  CK(dalloc(ctx, &ctx->m_loss, (size_t)c.H));
  if (ctx->Sp != c.S) {
    CK(dalloc(ctx, &ctx->m_Xp, B * c.D * ctx->Sp));
    CK(dalloc(ctx, &ctx->m_a, (size_t)c.H * B * c.S));
    CK(dalloc(ctx, &ctx->m_da, B * ctx->Sp));
  }
#undef CK
  ctx->mod_ready = true;
  return 0;
}

// strided [B, w] block copy on the ctx stream (packs / unpacks the [c1 h1 c2 h2] state)
int copy2d(rau_ctx* ctx, float* dst, size_t dst_rs, const float* src, size_t src_rs, size_t w) {
  HIPC(hipMemcpy2DAsync(dst, dst_rs * 4, src, src_rs * 4, w * 4, ctx->cfg.B,
                        hipMemcpyDeviceToDevice, ctx->st));
  return 0;
}

struct Masks {
  const uint32_t *we, *rnn, *q, *x, *mf;
  float s_we, s_rnn, s_q, s_x, s_mf;
};
Masks masks_of(rau_ctx* ctx) {
  const bool tr = ctx->mode == RAU_MODE_TRAIN;
  auto mk = [&](int site) -> const uint32_t* {
    return (tr && mask_p(ctx, site) > 0.f) ? ctx->mbits[site] : nullptr;
  };
  auto sc = [&](int site) { return 1.f / (1.f - mask_p(ctx, site)); };
  return Masks{mk(RAU_MASK_WE), mk(RAU_MASK_RNN), mk(RAU_MASK_Q), mk(RAU_MASK_X), mk(RAU_MASK_MF),
               sc(RAU_MASK_WE), sc(RAU_MASK_RNN), sc(RAU_MASK_Q), sc(RAU_MASK_X), sc(RAU_MASK_MF)};
}

// dW += dY^T X and db += column sums of dY for one Linear, rows = one clone's batch
int lin_wgrad(rau_ctx* ctx, Lin& l, const float* dY, const float* X, long ldx, bool bias = true,
              long ldy = 0) {
  const int B = ctx->cfg.B;
  RUN("wgrad_gemm", 2.0 * l.out * l.in * B, 0,
      gemm_tn_acc(ctx->st, l.out, l.in, B, dY, ldy ? ldy : l.out, X, ldx, l.dW, l.in, ctx->slab3,
                  bias ? l.db : nullptr, ctx->bf16 == 1));
  return 0;
}
// [rows][w_src] -> [rows][w_dst] on the ctx stream (re-pitching of [.., S] tensors)
int repitch(rau_ctx* ctx, float* dst, size_t w_dst, const float* src, size_t w_src, size_t rows) {
  const size_t w = std::min(w_dst, w_src);
  HIPC(hipMemcpy2DAsync(dst, w_dst * 4, src, w_src * 4, w * 4, rows, hipMemcpyDeviceToDevice,
                        ctx->st));
  return 0;
}

}  // namespace

// Philox masks for the current (seed, step) if the caller has not uploaded explicit ones;
// module-level forwards of one step all see the same mask tensors, each its own slice.
static int ensure_masks(rau_ctx* ctx) {
  if (ctx->mode != RAU_MODE_TRAIN) return 0;
  if (ctx->mod_masks_seed == ctx->seed && ctx->mod_masks_step == ctx->step && ctx->mod_masks_valid)
    return 0;
  for (int i = 0; i < 5; ++i)
    if (!ctx->mexplicit[i] && ctx->mp[i] > 0.f)
      RUN("fill_masks", 0, ctx->mcount[i] / 8.0,
          fill_masks(ctx->st, ctx->seed, (uint32_t)i, ctx->step, ctx->mp[i], ctx->mcount[i],
                     ctx->mbits[i], ctx->dkey));
  ctx->mod_masks_seed = ctx->seed;
  ctx->mod_masks_step = ctx->step;
  ctx->mod_masks_valid = true;
  return 0;
}

extern "C" {

// ------------------------------------------------------------ word_embed clone t
int rau_embed_forward(rau_ctx* ctx, int t, const int32_t* tokens_dev, float** we) {
  NEED(ctx && we, "null argument");
  const rau_config& c = ctx->cfg;
  NEED(t >= 0 && t < c.T, "rau_embed_forward: t=%d out of [0,%d)", t, c.T);
  if (int rc = mod_alloc(ctx)) return rc;
  if (int rc = ensure_masks(ctx)) return rc;
  if (!tokens_dev) {
    if (!ctx->have_batch) return fail(RAU_ERR_STATE, "rau_embed_forward: no tokens and no batch");
    tokens_dev = ctx->tokens + (size_t)t * c.B;
  }
  const Masks m = masks_of(ctx);
  float* out = ctx->we + (size_t)t * c.B * c.E;
  RUN("embed_fwd", 0, c.B * c.E * 8.0,
      embed_fwd(ctx->st, c.B, c.E, c.V, ctx->grp[RAU_GROUP_EMBED].w, tokens_dev, m.we, m.s_we, out,
                (size_t)t * c.B * c.E));
  *we = out;
  return RAU_OK;
}

int rau_embed_backward(rau_ctx* ctx, int t, const int32_t* tokens_dev, const float* d_we) {
  NEED(ctx && d_we, "null argument");
  const rau_config& c = ctx->cfg;
  NEED(t >= 0 && t < c.T, "rau_embed_backward: t=%d out of [0,%d)", t, c.T);
  if (int rc = mod_alloc(ctx)) return rc;
  if (!tokens_dev) {
    if (!ctx->have_batch) return fail(RAU_ERR_STATE, "rau_embed_backward: no tokens and no batch");
    tokens_dev = ctx->tokens + (size_t)t * c.B;
  }
  const Masks m = masks_of(ctx);
  RUN("embed_bwd", 0, c.B * c.E * 12.0,
      embed_bwd_rows(ctx->st, c.B, c.E, c.V, tokens_dev, d_we, ctx->we + (size_t)t * c.B * c.E, m.we,
                     (size_t)t * c.B * c.E, m.s_we, ctx->grp[RAU_GROUP_EMBED].g));
  return RAU_OK;
}

// ------------------------------------------------------------ DeepLSTM clone t
int rau_deeplstm_forward(rau_ctx* ctx, int t, const float* x, const float* state,
                         float** state_out) {
  if (ctx) set_skinny_policy(ctx);
  NEED(ctx && x && state_out, "null argument");
  const rau_config& c = ctx->cfg;
  NEED(t >= 0 && t < c.T, "rau_deeplstm_forward: t=%d out of [0,%d)", t, c.T);
  NEED(aligned16(x) && aligned16(state), "rau_deeplstm_forward: pointers must be 16-byte aligned");
  if (int rc = mod_alloc(ctx)) return rc;
  if (int rc = ensure_masks(ctx)) return rc;
  const int B = c.B, E = c.E, Rq = c.Rq, Q = ctx->Q;
  hipStream_t st = ctx->st;
  const Masks m = masks_of(ctx);
  if (!state) state = ctx->m_zero;   // init_state zeros, SS:358
  const size_t BRq = (size_t)B * Rq, G4 = (size_t)B * 4 * Rq;
  float* out = ctx->m_state + (size_t)t * B * Q;
  float* G1 = ctx->G1 + (size_t)t * G4;
  float* G2 = ctx->G2 + (size_t)t * G4;
  float* x2 = ctx->x2 + (size_t)t * BRq;
  auto gflop = [](double mm, double n, double k) { return 2.0 * mm * n * k; };
  {  // layer 1: i2h(x) + h2h(prev_h), DeepLSTM.lua:42-44
    LINOPTS(o);
    o.bias = ctx->i2h[0].b;
    o.bias2 = ctx->h2h[0].b;
    RUN("enc_i2h_gemm", gflop(B, 4 * Rq, E), 0, gemm_nt(st, B, 4 * Rq, E, x, E, ctx->i2h[0].W, E, G1, 4 * Rq, o));
    LINOPTS(oa);
    oa.accumulate = 1;
    RUN("enc_h2h_gemm", gflop(B, 4 * Rq, Rq), 0,
        gemm_nt(st, B, 4 * Rq, Rq, state + Rq, Q, ctx->h2h[0].W, Rq, G1, 4 * Rq, oa));
    RUN("lstm_fwd", 0, BRq * 40.0,
        lstm_fwd(st, GATES_DEEP, B, Rq, G1, state, Q, out, Q, out + Rq, Q,
                 ctx->tc1 + (size_t)t * BRq, x2, m.rnn, (size_t)t * BRq, m.s_rnn));
  }
  {  // layer 2 on dropout(h1), DeepLSTM.lua:39
    LINOPTS(o);
    o.bias = ctx->i2h[1].b;
    o.bias2 = ctx->h2h[1].b;
    RUN("enc_h2h_gemm", gflop(B, 4 * Rq, Rq), 0, gemm_nt(st, B, 4 * Rq, Rq, x2, Rq, ctx->i2h[1].W, Rq, G2, 4 * Rq, o));
    LINOPTS(oa);
    oa.accumulate = 1;
    RUN("enc_h2h_gemm", gflop(B, 4 * Rq, Rq), 0,
        gemm_nt(st, B, 4 * Rq, Rq, state + 3 * Rq, Q, ctx->h2h[1].W, Rq, G2, 4 * Rq, oa));
    RUN("lstm_fwd", 0, BRq * 40.0,
        lstm_fwd(st, GATES_DEEP, B, Rq, G2, state + 2 * Rq, Q, out + 2 * Rq, Q, out + 3 * Rq, Q,
                 ctx->tc2 + (size_t)t * BRq, nullptr, nullptr, 0, 1.f));
  }
  *state_out = out;
  return RAU_OK;
}

int rau_deeplstm_backward(rau_ctx* ctx, int t, const float* x, const float* state,
                          const float* d_state_out, float** d_x, float** d_state) {
  if (ctx) set_skinny_policy(ctx);
  NEED(ctx && x && d_state_out && d_x && d_state, "null argument");
  const rau_config& c = ctx->cfg;
  NEED(t >= 0 && t < c.T, "rau_deeplstm_backward: t=%d out of [0,%d)", t, c.T);
  NEED(aligned16(x) && aligned16(state) && aligned16(d_state_out),
       "rau_deeplstm_backward: pointers must be 16-byte aligned");
  if (int rc = mod_alloc(ctx)) return rc;
  const int B = c.B, E = c.E, Rq = c.Rq, Q = ctx->Q;
  hipStream_t st = ctx->st;
  const Masks m = masks_of(ctx);
  if (!state) state = ctx->m_zero;
  const size_t BRq = (size_t)B * Rq, G4 = (size_t)B * 4 * Rq;
  float* dst = ctx->m_dstate + (size_t)t * B * Q;
  float* dG1 = ctx->dG1 + (size_t)t * G4;
  float* dG2 = ctx->dG2 + (size_t)t * G4;
  float* dxo = ctx->dwe + (size_t)t * B * E;
  float *dc_in = ctx->m_tmp[0], *dc_prev = ctx->m_tmp[1], *dx2 = ctx->m_tmp[2];
  auto gflop = [](double mm, double n, double k) { return 2.0 * mm * n * k; };
  // ---- layer 2
  if (int rc = copy2d(ctx, dc_in, Rq, d_state_out + 2 * Rq, Q, Rq)) return rc;
  RUN("lstm_bwd", 0, BRq * 48.0,
      lstm_bwd(st, GATES_DEEP, B, Rq, ctx->G2 + (size_t)t * G4, state + 2 * Rq, Q,
               ctx->tc2 + (size_t)t * BRq, d_state_out + 3 * Rq, Q, nullptr, dc_in, dG2, dc_prev,
               nullptr, 0, nullptr, nullptr, 0));
  if (int rc = copy2d(ctx, dst + 2 * Rq, Q, dc_prev, Rq, Rq)) return rc;
  {
    LINOPTS(o);   // grad at prev_h of layer 2
    RUN("enc_h2h_dgrad", gflop(B, Rq, 4 * Rq), 0,
        gemm_nn(st, B, Rq, 4 * Rq, dG2, 4 * Rq, ctx->h2h[1].W, Rq, dst + 3 * Rq, Q, o));
    LINOPTS(o2);  // grad at layer-2 input, back through the inter-layer dropout
    o2.emask = m.rnn;
    o2.emask_e0 = (size_t)t * BRq;
    o2.emscale = m.s_rnn;
    RUN("enc_h2h_dgrad", gflop(B, Rq, 4 * Rq), 0,
        gemm_nn(st, B, Rq, 4 * Rq, dG2, 4 * Rq, ctx->i2h[1].W, Rq, dx2, Rq, o2));
  }
  // ---- layer 1: next_h of layer 1 fans out to the output state and to layer 2
  if (int rc = copy2d(ctx, dc_in, Rq, d_state_out, Q, Rq)) return rc;
  RUN("lstm_bwd", 0, BRq * 48.0,
      lstm_bwd(st, GATES_DEEP, B, Rq, ctx->G1 + (size_t)t * G4, state, Q,
               ctx->tc1 + (size_t)t * BRq, d_state_out + Rq, Q, dx2, dc_in, dG1, dc_prev, nullptr,
               0, nullptr, nullptr, 0));
  if (int rc = copy2d(ctx, dst, Q, dc_prev, Rq, Rq)) return rc;
  {
    LINOPTS(o);
    RUN("enc_h2h_dgrad", gflop(B, Rq, 4 * Rq), 0,
        gemm_nn(st, B, Rq, 4 * Rq, dG1, 4 * Rq, ctx->h2h[0].W, Rq, dst + Rq, Q, o));
    LINOPTS(o2);
    RUN("enc_i2h_dgrad", gflop(B, E, 4 * Rq), 0,
        gemm_nn(st, B, E, 4 * Rq, dG1, 4 * Rq, ctx->i2h[0].W, E, dxo, E, o2));
  }
  // ---- accGradParameters
  if (int rc = lin_wgrad(ctx, ctx->i2h[0], dG1, x, E)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->h2h[0], dG1, state + Rq, Q)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->i2h[1], dG2, ctx->x2 + (size_t)t * BRq, Rq)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->h2h[1], dG2, state + 3 * Rq, Q)) return rc;
  *d_x = dxo;
  *d_state = dst;
  return RAU_OK;
}

// ------------------------------------------------------------ multimodal clone h
int rau_multimodal_forward(rau_ctx* ctx, int h, const float* q, const float* X, const float* c_prev,
                           const float* h_prev, float** logits, float** do_pred, float** attprob,
                           float** c_out, float** h_out) {
  if (ctx) set_skinny_policy(ctx);
  NEED(ctx && q, "null argument");
  const rau_config& c = ctx->cfg;
  NEED(h >= 0 && h < c.H, "rau_multimodal_forward: h=%d out of [0,%d)", h, c.H);
  if (int rc = mod_alloc(ctx)) return rc;
  if (int rc = ensure_masks(ctx)) return rc;
  if (!X) {
    if (!ctx->have_batch) return fail(RAU_ERR_STATE, "rau_multimodal_forward: no X and no batch");
    X = ctx->feats;
  }
  if (!c_prev) c_prev = ctx->m_zero;   // att_c / att_h zeros, SS:362-365
  if (!h_prev) h_prev = ctx->m_zero;
  NEED(aligned16(q) && aligned16(X) && aligned16(c_prev) && aligned16(h_prev),
       "rau_multimodal_forward: pointers must be 16-byte aligned");
  const int B = c.B, D = c.D, S = ctx->Sp, SL = c.S, M = c.M, A = c.A, R = c.R, K = c.K, Q = ctx->Q;
  hipStream_t st = ctx->st;
  const Masks m = masks_of(ctx);
  auto gflop = [](double mm, double n, double k) { return 2.0 * mm * n * k; };
  const size_t BM_ = (size_t)B * M, BR_ = (size_t)B * R;
  ctx->I_shared = false;
  ctx->fwd_done = false;   // the step-level slots are being overwritten
  if (S != SL && X != ctx->feats) {   // dense [B,D,SL] from the caller -> pitched (pad columns stay 0)
    if (int rc = repitch(ctx, ctx->m_Xp, S, X, SL, (size_t)B * D)) return rc;
    X = ctx->m_Xp;
  }
  // q_embed's input half: dropout(q) Wq^T + bq + bh   (SS:233-235)
  float* qd = ctx->qd + (size_t)h * B * Q;
  RUN("apply_mask", 0, (double)B * Q * 8,
      apply_mask(st, (size_t)B * Q, (size_t)B * Q, q, m.q, m.s_q, qd, (size_t)h * B * Q));
  ctx->yq_shared = false;   // this path fills the hop's own rows
  {
    LINOPTS(o);
    o.bias = ctx->q_proj.b;
    o.bias2 = ctx->h_proj.b;
    RUN("q_proj_gemm", gflop(B, M, Q), 0,
        gemm_nt(st, B, M, Q, qd, Q, ctx->q_proj.W, Q, ctx->Yq + (size_t)h * BM_, M, o));
  }
  // i_embed (SS:238-242) and attbycontent's ifeatproj (SS:247-249) for this clone
  const float* xin = X;
  if (m.x) {
    float* xd = ctx->xd + (size_t)h * B * D * S;
    RUN("dropout_features", 0, 2.0 * B * D * S * 4,
        dropout_features(st, 1, (size_t)B * D * S, X, m.x, m.s_x, xd, (size_t)h * B * D * SL, SL, S));
    xin = xd;
  }
  RUN("transpose", 0, (double)M * D * 8, transpose2d(st, M, D, ctx->i_embed.W, ctx->WiT));
  RUN("transpose", 0, (double)A * M * 8, transpose2d(st, A, M, ctx->att_i.W, ctx->WpT));
  float* Ih = ctx->I + (size_t)h * BM_ * S;
  float* Th = ctx->T + (size_t)h * B * A * S;
  RUN("conv_embed_fwd", gflop(M, (double)B * S, D), ((double)B * D * S + BM_ * S) * 4,
      conv_embed_fwd(st, B, D, S, M, xin, ctx->WiT, ctx->i_embed.b, Ih, ctx->bf16));
  RUN("conv_att_pre", gflop(A, (double)B * S, M), (BM_ * S + (double)B * A * S) * 4,
      conv_att_pre(st, B, M, S, A, Ih, ctx->WpT, ctx->att_i.b, Th, ctx->bf16));
  float* co = ctx->cc + (size_t)(h + 1) * BR_;
  float* ho = ctx->hh + (size_t)(h + 1) * BR_;
  if (int rc = hop_forward(ctx, h, c_prev, h_prev, co, ho, Ih, Th, nullptr)) return rc;
  if (logits) *logits = ctx->logits + (size_t)h * B * K;
  if (do_pred) *do_pred = ctx->dopred + (size_t)h * B;
  if (attprob) {
    *attprob = ctx->a + (size_t)h * B * S;
    if (S != SL) {   // hand back a dense [B,SL] copy
      float* ad = ctx->m_a + (size_t)h * B * SL;
      if (int rc = repitch(ctx, ad, SL, ctx->a + (size_t)h * B * S, S, B)) return rc;
      *attprob = ad;
    }
  }
  if (c_out) *c_out = co;
  if (h_out) *h_out = ho;
  return RAU_OK;
}

int rau_multimodal_backward(rau_ctx* ctx, int h, const float* q, const float* X,
                            const float* c_prev, const float* h_prev, const float* d_logits,
                            const float* d_do_pred, const float* d_attprob, const float* d_c,
                            const float* d_h, float** d_q, float** d_X, float** d_c_prev,
                            float** d_h_prev) {
  if (ctx) set_skinny_policy(ctx);
  NEED(ctx && q && d_logits, "null argument");
  const rau_config& c = ctx->cfg;
  NEED(h >= 0 && h < c.H, "rau_multimodal_backward: h=%d out of [0,%d)", h, c.H);
  if (int rc = mod_alloc(ctx)) return rc;
  if (!X) {
    if (!ctx->have_batch) return fail(RAU_ERR_STATE, "rau_multimodal_backward: no X and no batch");
    X = ctx->feats;
  }
  if (!c_prev) c_prev = ctx->m_zero;
  if (!h_prev) h_prev = ctx->m_zero;
  NEED(aligned16(q) && aligned16(X) && aligned16(c_prev) && aligned16(h_prev) &&
           aligned16(d_logits) && aligned16(d_attprob) && aligned16(d_c) && aligned16(d_h),
       "rau_multimodal_backward: pointers must be 16-byte aligned");
  const int B = c.B, D = c.D, S = ctx->Sp, SL = c.S, M = c.M, A = c.A, R = c.R, Q = ctx->Q;
  hipStream_t st = ctx->st;
  const Masks m = masks_of(ctx);
  auto gflop = [](double mm, double n, double k) { return 2.0 * mm * n * k; };
  const size_t BM_ = (size_t)B * M, BS_ = (size_t)B * S, BR_ = (size_t)B * R;
  if (S != SL) {
    if (X != ctx->feats) {   // the forward of this clone re-pitched the same X
      if (int rc = repitch(ctx, ctx->m_Xp, S, X, SL, (size_t)B * D)) return rc;
      X = ctx->m_Xp;
    }
    if (d_attprob) {
      HIPC(hipMemsetAsync(ctx->m_da, 0, BS_ * sizeof(float), st));
      if (int rc = repitch(ctx, ctx->m_da, S, d_attprob, SL, B)) return rc;
      d_attprob = ctx->m_da;
    }
  }
  const float* Ih = ctx->I + (size_t)h * BM_ * S;
  const float* mfh = ctx->mf + (size_t)h * BM_;
  float* Th = ctx->T + (size_t)h * B * A * S;
  float* dZh = ctx->dZ + (size_t)h * BM_ * S;
  const float* xin = m.x ? ctx->xd + (size_t)h * B * D * S : X;

  HopGrad g{};
  g.dl = d_logits;
  g.dc_next = d_c;
  g.dh_next = d_h;
  g.da_out = d_attprob;
  g.dc_out = ctx->m_dc + (size_t)h * BR_;
  g.dh_out = ctx->m_dh + (size_t)h * BR_;
  if (d_do_pred) {
    // out_do_pred = Sigmoid(Linear(merge_feat)) (SS:281-282): s = ddp dp (1-dp);
    // dmf += s (x) wd; dwd += s^T mf; dbd += sum s.   (feval always passes zeros, SS:566)
    RUN("dopred_bwd", 0, 0, sigmoid_bwd(st, B, d_do_pred, ctx->dopred + (size_t)h * B, ctx->m_s));
    RUN("dopred_bwd", 0, 0, row_scale(st, B, M, ctx->m_s, nullptr, ctx->do_pred.W, ctx->m_add));
    g.dmf_add = ctx->m_add;
  }
  if (int rc = hop_backward(ctx, h, c_prev, Ih, g)) return rc;
  if (d_do_pred) {
    // m_add is free again once dpre has been formed: reuse it for s (.) mf rows
    RUN("dopred_bwd", 0, 0, row_scale(st, B, M, ctx->m_s, mfh, nullptr, ctx->m_add));
    RUN("colsum", 0, 0, colsum_acc(st, B, M, ctx->m_add, M, ctx->do_pred.dW, ctx->coltmp3));
    RUN("colsum", 0, 0, colsum_acc(st, B, 1, ctx->m_s, 1, ctx->do_pred.db, ctx->coltmp3));
  }
  // ---- 1x1-conv gradients of this clone: dI = Wp^T dS + dj (x) a; dWp += dS I^T;
  // dWi += (dI (1-I^2)) X'^T; bias gradients
  RUN("conv_att_dgrad", gflop(M, (double)B * S, A), ((double)B * A * S + 2.0 * BM_ * S) * 4,
      conv_att_dgrad(st, B, M, S, A, Th, ctx->att_i.W, ctx->dj + (size_t)h * BM_,
                     ctx->a + (size_t)h * BS_, dZh, ctx->bf16));
  RUN("conv_att_wgrad", gflop(A, M, (double)B * S), ((double)B * A * S + BM_ * S) * 4,
      conv_att_wgrad(st, B, M, S, A, Th, Ih, ctx->att_i.dW, ctx->slab2, ctx->bf16));
  RUN("conv_embed_wgrad", gflop(M, D, (double)B * S), (BM_ * S + (double)B * D * S) * 4,
      conv_embed_wgrad(st, B, D, S, M, dZh, Ih, xin, ctx->i_embed.dW, ctx->slab2, ctx->bf16,
                       ctx->i_embed.db));
  // ---- gradient w.r.t. q through q_embed's dropout
  float* dqo = ctx->m_dq + (size_t)h * B * Q;
  {
    LINOPTS(o);
    o.emask = m.q;
    o.emask_e0 = (size_t)h * B * Q;
    o.emscale = m.s_q;
    RUN("q_proj_dgrad", gflop(B, Q, M), 0,
        gemm_nn(st, B, Q, M, ctx->dqt + (size_t)h * BM_, M, ctx->q_proj.W, Q, dqo, Q, o));
  }
  // ---- feature-map gradient (the reference computes it and SS:579 throws it away): on request
  if (d_X) {
    if (!ctx->m_dX) {
      if (int rc = dalloc(ctx, &ctx->m_dX, (size_t)B * D * S)) return rc;
      if (int rc = dalloc(ctx, &ctx->m_dZ, BM_ * S)) return rc;
    }
    RUN("mul_dtanh", 0, BM_ * S * 12.0, mul_dtanh(st, BM_ * S, dZh, Ih, ctx->m_dZ));
    RUN("conv_embed_dgrad", gflop(D, (double)B * S, M), (BM_ * S + (double)B * D * S) * 4,
        conv_embed_dgrad(st, B, D, S, M, ctx->m_dZ, ctx->i_embed.W, ctx->m_dX));
    float* dXo = ctx->m_dX;
    if (S != SL) {   // dense [B,D,SL] for the caller (reuses the re-pitch buffer's sibling)
      if (!ctx->m_dXd)
        if (int rc = dalloc(ctx, &ctx->m_dXd, (size_t)B * D * SL)) return rc;
      if (int rc = repitch(ctx, ctx->m_dXd, SL, ctx->m_dX, S, (size_t)B * D)) return rc;
      dXo = ctx->m_dXd;
    }
    if (m.x)   // the mask is indexed logically: applied on the dense tensor
      RUN("apply_mask", 0, (double)B * D * SL * 8,
          apply_mask(st, (size_t)B * D * SL, (size_t)B * D * SL, dXo, m.x, m.s_x, dXo,
                     (size_t)h * B * D * SL));
    *d_X = dXo;
  }
  // ---- accGradParameters of the clone's Linears
  const size_t h4 = (size_t)h * B * 4 * R;
  if (int rc = lin_wgrad(ctx, ctx->cls, d_logits, mfh, M)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->lstm_out, ctx->dpre + (size_t)h * BM_, ctx->hh + (size_t)(h + 1) * BR_, R)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->lstm_i2h, ctx->dg4 + h4, ctx->j + (size_t)h * BM_, M)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->lstm_h2h, ctx->dg4 + h4, h_prev, R)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->feat_attprob, ctx->dj + (size_t)h * BM_, ctx->a + (size_t)h * BS_, S)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->att_mem, ctx->dz + (size_t)h * BS_, h_prev, R, true, S)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->att_q, ctx->du + (size_t)h * B * A, ctx->qf + (size_t)h * BM_, M)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->q_proj, ctx->dqt + (size_t)h * BM_, ctx->qd + (size_t)h * B * Q, Q)) return rc;
  if (int rc = lin_wgrad(ctx, ctx->h_proj, ctx->dqt + (size_t)h * BM_, h_prev, R)) return rc;
  // attscore: dws = sum_b dwsp[b]; dbs = sum dz.  ifeatproj bias: sum_b du[b]
  RUN("colsum", 0, (double)B * A * 4,
      colsum_acc(st, B, A, ctx->dwsp + (size_t)h * B * A, A, ctx->att_score.dW, ctx->coltmp3));
  HIPC(hipMemsetAsync(ctx->tmpS, 0, S * sizeof(float), st));
  RUN("colsum", 0, (double)B * S * 4, colsum_acc(st, B, SL, ctx->dz + (size_t)h * BS_, S, ctx->tmpS, ctx->coltmp3));
  RUN("colsum", 0, S * 4.0, colsum_acc(st, SL, 1, ctx->tmpS, 1, ctx->att_score.db, ctx->coltmp3));
  RUN("colsum", 0, (double)B * A * 4,
      colsum_acc(st, B, A, ctx->du + (size_t)h * B * A, A, ctx->att_i.db, ctx->coltmp3));
  if (d_q) *d_q = dqo;
  if (d_c_prev) *d_c_prev = g.dc_out;
  if (d_h_prev) *d_h_prev = g.dh_out;
  return RAU_OK;
}

// ------------------------------------------------------------ criteria[h]
int rau_criterion_forward(rau_ctx* ctx, int h, const float* logits, const int32_t* labels_dev,
                          float* loss) {
  NEED(ctx && logits, "null argument");
  const rau_config& c = ctx->cfg;
  NEED(h >= 0 && h < c.H, "rau_criterion_forward: h=%d out of [0,%d)", h, c.H);
  if (int rc = mod_alloc(ctx)) return rc;
  if (!labels_dev) {
    if (!ctx->have_labels) return fail(RAU_ERR_STATE, "rau_criterion_forward: no labels");
    labels_dev = ctx->labels_d;
  }
  RUN("ce_fwd", 0, (double)c.B * c.K * 12,
      ce_fwd(ctx->st, c.B, c.K, c.M, logits, labels_dev, nullptr, nullptr, nullptr,
             ctx->dl + (size_t)h * c.B * c.K, ctx->lossrow + (size_t)h * c.B,
             ctx->argmax_d + (size_t)h * c.B, nullptr));
  RUN("loss_reduce", 0, 0,
      loss_reduce(ctx->st, 1, c.B, ctx->lossrow + (size_t)h * c.B, ctx->m_loss + h));
  if (loss) {
    HIPC(hipMemcpyAsync(loss, ctx->m_loss + h, sizeof(float), hipMemcpyDeviceToHost, ctx->st));
    HIPC(hipStreamSynchronize(ctx->st));
  }
  return RAU_OK;
}

int rau_criterion_backward(rau_ctx* ctx, int h, const float* logits, const int32_t* labels_dev,
                           float scale, float** d_logits) {
  NEED(ctx && logits && d_logits, "null argument");
  const rau_config& c = ctx->cfg;
  NEED(h >= 0 && h < c.H, "rau_criterion_backward: h=%d out of [0,%d)", h, c.H);
  if (int rc = mod_alloc(ctx)) return rc;
  if (!labels_dev) {
    if (!ctx->have_labels) return fail(RAU_ERR_STATE, "rau_criterion_backward: no labels");
    labels_dev = ctx->labels_d;
  }
  float* dl = ctx->dl + (size_t)h * c.B * c.K;
  RUN("ce_fwd", 0, (double)c.B * c.K * 12,
      ce_fwd(ctx->st, c.B, c.K, c.M, logits, labels_dev, nullptr, nullptr, nullptr, dl,
             ctx->lossrow + (size_t)h * c.B, ctx->argmax_d + (size_t)h * c.B, nullptr));
  if (scale != 1.f)   // dpred:mul(nHop), SS:569
    RUN("scale_hops", 0, (double)c.B * c.K * 8, scale_inplace(ctx->st, (size_t)c.B * c.K, scale, dl));
  *d_logits = dl;
  return RAU_OK;
}

}  // extern "C"
