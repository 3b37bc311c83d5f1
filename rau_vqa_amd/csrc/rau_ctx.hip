// rau_ctx.hip -- the rau_ctx object and the C ABI of include/rau.h.
//
// Orchestrates one training step of the Recurrent Answering Unit on one
// MI355X: question encoder (word embedding + 2-layer LSTM over <=T tokens),
// H weight-shared answering hops (attention over the S-position feature map,
// attention LSTM, classifier, cross-entropy) and the full backward pass into
// the three flat gradient buffers.  Reference being replaced: feval in
// experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua:428-596 and the graph
// it drives (SS:198-316, model/ATTLSTM.lua, model/DeepLSTM.lua).
//
// Restructuring relative to the reference's module-by-module execution (all of
// it value-preserving; see DESIGN.md "Schedule"):
//   * work that does not depend on the recurrence is hoisted out of the loops
//     and batched: layer input projections over all tokens, the q projection
//     over all hops, dq = sum_h over hops, and every Linear weight gradient
//     (one [H*B]- or [T*B]-row GEMM per weight instead of one per clone);
//   * dropout masks are bit-packed and applied while staging GEMM operands;
//   * the dead gradient w.r.t. the feature map (SS:579 discards it) is skipped.
#include "rau_ctx.h"

static thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static int prof_collect(rau_ctx* ctx) {
  if (ctx->precs.empty()) return 0;
  HIPC(hipStreamSynchronize(ctx->st));
  HIPC(hipStreamSynchronize(ctx->st2));
  HIPC(hipStreamSynchronize(ctx->st3));
  FILE* tl = nullptr;   // RAU_PROF_TIMELINE=path: per-launch (class, stream, start, end) in ms
  if (const char* p = std::getenv("RAU_PROF_TIMELINE")) tl = std::fopen(p, "a");
  for (auto& r : ctx->precs) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    if (tl) {
      float t0 = 0.f;
      hipEventElapsedTime(&t0, ctx->precs[0].a, r.a);
      std::fprintf(tl, "%s,%d,%.4f,%.4f\n", ctx->pcls[r.cls].name.c_str(), r.sid, t0, t0 + ms);
    }
    ctx->pcls[r.cls].ms += ms;
    ctx->evpool.push_back(r.a);
    ctx->evpool.push_back(r.b);
  }
  ctx->precs.clear();
  if (tl) { std::fprintf(tl, "#\n"); std::fclose(tl); }
  return 0;
}
// The Linear weight-gradient problems of the mult group (over hop-major rows) and of the encoder
// (over token-major rows): dW += dY^T X, db += column sums of dY.  dz and a are pitched (leading
// dimension = position pitch, Linear size = logical S).
static int mult_wgrad_problems(rau_ctx* ctx, TnProblem* pr) {
  const rau_config& c = ctx->cfg;
  const int S = ctx->Sp, SL = c.S, M = c.M, A = c.A, R = c.R, K = c.K, Q = ctx->Q;
  const size_t BR_ = (size_t)c.B * R;
  const float* hprev = ctx->hh;            // h_{0..H-1}
  const float* hnew = ctx->hh + BR_;       // h_{1..H}
  struct WG { Lin* l; const float* dY; long ldy; const float* X; long ldx; };
  const WG wgs[] = {
      {&ctx->cls, ctx->dl, K, ctx->mf, M},             {&ctx->lstm_out, ctx->dpre, M, hnew, R},
      {&ctx->lstm_i2h, ctx->dg4, 4 * R, ctx->j, M},    {&ctx->lstm_h2h, ctx->dg4, 4 * R, hprev, R},
      {&ctx->feat_attprob, ctx->dj, M, ctx->a, S},     {&ctx->att_mem, ctx->dz, S, hprev, R},
      {&ctx->att_q, ctx->du, A, ctx->qf, M},           {&ctx->q_proj, ctx->dqt, M, ctx->qd, Q},
      {&ctx->h_proj, ctx->dqt, M, hprev, R}};
  int n = 0;
  for (const WG& w : wgs) {
    const int out = w.l == &ctx->att_mem ? SL : w.l->out;
    const int in = w.l == &ctx->feat_attprob ? SL : w.l->in;
    pr[n++] = TnProblem{out, in, w.dY, w.ldy, w.X, w.ldx, w.l->dW, w.l->db};
  }
  return n;
}
static int enc_wgrad_problems(rau_ctx* ctx, size_t row0, TnProblem* pr) {
  struct WG { Lin* l; const float* dY; const float* X; };
  const WG wgs[] = {{&ctx->i2h[0], ctx->dG1, ctx->we}, {&ctx->h2h[0], ctx->dG1, ctx->h1},
                    {&ctx->i2h[1], ctx->dG2, ctx->x2}, {&ctx->h2h[1], ctx->dG2, ctx->h2}};
  int n = 0;
  for (const WG& w : wgs)
    pr[n++] = TnProblem{w.l->out, w.l->in, w.dY + row0 * w.l->out, w.l->out,
                        w.X + row0 * w.l->in, w.l->in, w.l->dW, w.l->db};
  return n;
}

// ================================================================== C ABI
extern "C" {

const char* rau_last_error(void) { return g_err; }
int rau_abi_version(void) { return RAU_ABI_VERSION; }

// ---- diagnostics: host-side predicates, no device touched (include/rau.h, last section)
int rau_split_guard_check(size_t ws_floats, size_t offset, int nsplit, size_t per_split_floats) {
  // a stand-in workspace in host memory: the check is pointer arithmetic only, nothing is dereferenced
  static thread_local float anchor[1];
  const float* base = anchor;
  split_ws_register(base, ws_floats);
  const bool ok = split_span_ok(base + offset, nsplit, per_split_floats);
  split_ws_unregister(base);
  if (!ok)
    return fail(RAU_ERR_STATE, "split-K partials [%zu + %d x %zu) outside a workspace of %zu floats: "
                               "nothing would be launched", offset, nsplit, per_split_floats, ws_floats);
  return RAU_OK;
}
int rau_enc_ws_coresident(int batch, int blocks_per_cu, int n_cus) {
  return (enc_ws_ok(batch, 512) && enc_ws_coresident(batch, blocks_per_cu, n_cus)) ? 1 : 0;
}

void rau_default_config(rau_config* cfg) {
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->B = 100;   // opt.batch_size default, SS:48
  cfg->T = 26;
  cfg->V = 14000;
  cfg->E = 200;
  cfg->Rq = 512;
  cfg->D = 512;
  cfg->S = 196;
  cfg->M = 512;
  cfg->A = 256;
  cfg->R = 512;
  cfg->K = 1000;
  cfg->H = 8;
  cfg->p_we = cfg->p_rnn = cfg->p_q = cfg->p_x = cfg->p_mf = 0.5f;
  cfg->dtype = RAU_F32;
  cfg->device_id = 0;
}

int rau_create(const rau_config* cfg, rau_ctx** out) {
  NEED(cfg && out, "rau_create: null argument");
  const rau_config& c = *cfg;
  NEED(c.B > 0 && c.T > 0 && c.V > 1 && c.H > 0, "rau_create: B,T,H must be > 0 and V > 1");
  NEED(c.E > 0 && c.E % 4 == 0, "rau_create: E=%d must be a positive multiple of 4", c.E);
  NEED(c.S > 0, "rau_create: S=%d must be positive", c.S);
  NEED(c.K > 0 && c.K % 4 == 0, "rau_create: K=%d must be a positive multiple of 4", c.K);
  NEED(c.Rq > 0 && c.Rq % 4 == 0 && c.R > 0 && c.R % 4 == 0 && c.M > 0 && c.M % 4 == 0 &&
           c.A > 0 && c.A % 4 == 0 && c.D > 0 && c.D % 4 == 0,
       "rau_create: Rq,R,M,A,D must be positive multiples of 4");
  NEED(c.dtype == RAU_F32 || c.dtype == RAU_BF16 || c.dtype == RAU_F32S,
       "rau_create: dtype %d not supported", c.dtype);
  const float ps[5] = {c.p_we, c.p_rnn, c.p_q, c.p_x, c.p_mf};
  for (float p : ps) NEED(p >= 0.f && p < 1.f, "rau_create: dropout p=%f out of [0,1)", p);

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(RAU_ERR_DEVICE, "no HIP device available (%s); librau has no CPU fallback",
                hipGetErrorString(e));
  NEED(c.device_id >= 0 && c.device_id < ndev, "rau_create: device_id %d of %d", c.device_id, ndev);
  HIPC(hipSetDevice(c.device_id));
  hipDeviceProp_t prop;
  HIPC(hipGetDeviceProperties(&prop, c.device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RAU_ERR_DEVICE, "device %d is %s; librau is built for gfx950 only", c.device_id,
                prop.gcnArchName);

  rau_ctx* ctx = new rau_ctx();
  ctx->cfg = c;
  ctx->Q = 4 * c.Rq;
  ctx->Sp = (c.S + 3) & ~3;
  ctx->bf16 = c.dtype;   // 0 exact f32, 1 bf16-rounded operands, 2 split operands (conv GEMMs)
  // The device Philox masks keep an element when an 8-bit draw >= round(p * 256) (fill_masks): the
  // effective drop probability is p quantised to 1/256, and the inverted-dropout scale 1/(1-p)
  // uses THAT value so that E[mask * scale] = 1 exactly (the reference's 0.5 is representable).
  for (int i = 0; i < 5; ++i) {
    const float pq = std::lround(ps[i] * 256.0f) / 256.0f;
    if (pq >= 1.f) {
      delete ctx;
      return fail(RAU_ERR_INVALID, "rau_create: dropout p=%f rounds to 1 at 1/256 resolution", ps[i]);
    }
    ctx->mp[i] = pq;
    ctx->mp_exact[i] = ps[i];
  }
  *out = nullptr;
#define CK(x)                \
  do {                       \
    int rc_ = (x);           \
    if (rc_ != 0) {          \
      rau_destroy(ctx);      \
      return rc_;            \
    }                        \
  } while (0)
  {
    // the chain stream outranks the bulk stream: its short kernels are the critical
    // path and must not queue behind the bulk GEMMs' workgroups
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipError_t es = hipStreamCreateWithPriority(&ctx->st, hipStreamNonBlocking, prio_hi);
    if (es == hipSuccess) es = hipStreamCreateWithPriority(&ctx->st2, hipStreamNonBlocking, prio_lo);
    if (es != hipSuccess) {
      delete ctx;
      return fail(RAU_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(es));
    }
  }
  hipEventCreate(&ctx->ev0);
  hipEventCreate(&ctx->ev1);
  const unsigned evflags = hipEventDisableTiming;
  for (hipEvent_t* e : {&ctx->evA, &ctx->evD, &ctx->evW, &ctx->evE, &ctx->evW3, &ctx->evM3, &ctx->evEnd, &ctx->evE1,
                        &ctx->evG0, &ctx->evG, &ctx->evQ0, &ctx->evQ, &ctx->evDq})
    hipEventCreateWithFlags(e, evflags);
  for (hipEvent_t& e : ctx->evGc) hipEventCreateWithFlags(&e, evflags);
  {
    int plo = 0, phi = 0;
    hipDeviceGetStreamPriorityRange(&plo, &phi);
    hipStreamCreateWithPriority(&ctx->st3, hipStreamNonBlocking, plo);
  }
  // Default partition: pairs, then the last two hops alone.  The forward phase ends one hop after
  // the last group's GEMMs and the backward bulk work can start one hop into the backward chain
  // (measured on H = 8: 2,2,2,1,1 vs 2,2,2,2); pairs elsewhere keep the launches large.
  // RAU_HOP_GROUPS="4,2,1,1" overrides (sizes must sum to H).
  {
    std::vector<int> sizes;
    if (const char* eg = std::getenv("RAU_HOP_GROUPS")) {
      int sum = 0;
      for (const char* p = eg; *p;) {
        const int v = std::atoi(p);
        if (v < 1) { sizes.clear(); break; }
        sizes.push_back(v);
        sum += v;
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
      if (sum != c.H) sizes.clear();
    }
    if (sizes.empty() && c.B <= 64) {
      // small batches are bound by the recurrence's launches, not by the conv GEMMs: one group
      // (B = 32 / 64: 3.64 / 4.43 -> 3.55 / 4.33 ms; B = 128: 6.09 -> 6.49, so not there)
      sizes.push_back(c.H);
    }
    if (sizes.empty()) {
      int left = c.H;
      while (left > 3) { sizes.push_back(2); left -= 2; }
      while (left > 0) { sizes.push_back(1); left -= 1; }
      if (c.H == 2) sizes = {1, 1};
    }
    ctx->groups.assign(c.H, 0);
    int h0 = 0;
    for (int n : sizes) { ctx->groups[h0] = n; h0 += n; }
    if (const char* eg = std::getenv("RAU_BWD_GROUPS")) {
      std::vector<int> bs;
      int sum = 0;
      for (const char* p = eg; *p;) {
        const int v = std::atoi(p);
        if (v < 1) { bs.clear(); sum = -1; break; }
        bs.push_back(v);
        sum += v;
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
      if (sum == c.H) {
        ctx->bgroups.assign(c.H, 0);
        int b0 = 0;
        for (int n : bs) { ctx->bgroups[b0] = n; b0 += n; }
      }
    }
  }
  ctx->evF.resize(c.H);
  ctx->evK.resize(c.H);

  ctx->evH.resize(c.H);
  for (int i = 0; i < c.H; ++i) {
    hipEventCreateWithFlags(&ctx->evF[i], evflags);
    hipEventCreateWithFlags(&ctx->evK[i], evflags);

    hipEventCreateWithFlags(&ctx->evH[i], evflags);
  }
  hipEventCreateWithFlags(&ctx->evHd, evflags);

  // S below is the position PITCH of the device tensors; SL the logical number of positions
  // (they differ only for maps like 7x7 = 49 -> 52: pad columns carry zero features, zero
  // attention and zero gradients, see att_fwd_fused)
  const int B = c.B, T = c.T, E = c.E, Rq = c.Rq, D = c.D, S = ctx->Sp, SL = c.S, M = c.M, A = c.A,
            R = c.R, K = c.K, H = c.H, Q = ctx->Q;
  // ---- parameter layouts (weight [out,in] then bias [out])
  {
    Group& g = ctx->grp[RAU_GROUP_EMBED];
    g.layout.push_back({"word_embed.weight", 0, c.V, E});  // LookupTable, SS:204
    g.n = (size_t)c.V * E;
  }
  {
    LayoutBuilder lb{&ctx->grp[RAU_GROUP_RNN]};
    ctx->i2h[0] = lb.take("rnn.l1.i2h", 4 * Rq, E);   // DeepLSTM.lua:42
    ctx->h2h[0] = lb.take("rnn.l1.h2h", 4 * Rq, Rq);  // DeepLSTM.lua:43
    ctx->i2h[1] = lb.take("rnn.l2.i2h", 4 * Rq, Rq);
    ctx->h2h[1] = lb.take("rnn.l2.h2h", 4 * Rq, Rq);
    ctx->grp[RAU_GROUP_RNN].n = lb.off;
  }
  {
    LayoutBuilder lb{&ctx->grp[RAU_GROUP_MULT]};
    ctx->q_proj = lb.take("q_embed.q_proj", M, Q);          // SS:233
    ctx->h_proj = lb.take("q_embed.h_proj", M, R);          // SS:234
    ctx->i_embed = lb.take("i_embed.conv", M, D);           // SS:240
    ctx->att_q = lb.take("attbycontent.qfeatatt", A, M);    // SS:246
    ctx->att_i = lb.take("attbycontent.ifeatproj", A, M);   // SS:247
    ctx->att_score = lb.take("attbycontent.attscore", 1, A);  // SS:251
    ctx->att_mem = lb.take("attbymemory.linear", SL, R);     // SS:287
    ctx->feat_attprob = lb.take("classifier.feat_attprob", M, SL);  // SS:271
    ctx->lstm_i2h = lb.take("classifier.attlstm.i2h", 4 * R, M);   // ATTLSTM.lua:6
    ctx->lstm_h2h = lb.take("classifier.attlstm.h2h", 4 * R, R);   // ATTLSTM.lua:7
    ctx->lstm_out = lb.take("classifier.lstm_out", M, R);          // SS:279
    ctx->cls = lb.take("classifier.out_score", K, M);              // SS:280
    ctx->do_pred = lb.take("classifier.out_do_pred", 1, M);        // SS:281
    ctx->grp[RAU_GROUP_MULT].n = lb.off;
  }
  for (int gi = 0; gi < 3; ++gi) {
    CK(dalloc(ctx, &ctx->grp[gi].w, ctx->grp[gi].n + 4));
    CK(dalloc(ctx, &ctx->grp[gi].g, ctx->grp[gi].n + 4));
  }
  for (int L = 0; L < 2; ++L) {
    bind(ctx->i2h[L], ctx->grp[RAU_GROUP_RNN]);
    bind(ctx->h2h[L], ctx->grp[RAU_GROUP_RNN]);
  }
  Lin* ml[] = {&ctx->q_proj, &ctx->h_proj, &ctx->i_embed, &ctx->att_q, &ctx->att_i,
               &ctx->att_score, &ctx->att_mem, &ctx->feat_attprob, &ctx->lstm_i2h,
               &ctx->lstm_h2h, &ctx->lstm_out, &ctx->cls, &ctx->do_pred};
  for (Lin* l : ml) bind(*l, ctx->grp[RAU_GROUP_MULT]);

  // ---- batch
  CK(dalloc(ctx, &ctx->feats, (size_t)B * D * S));
  CK(dalloc(ctx, &ctx->tokens, (size_t)T * B));
  CK(dalloc(ctx, &ctx->lens_d, (size_t)B));
  CK(dalloc(ctx, &ctx->labels_d, (size_t)B));
  CK(dalloc(ctx, &ctx->utok, (size_t)T * B));
  CK(dalloc(ctx, &ctx->ustart, (size_t)T * B + 1));
  CK(dalloc(ctx, &ctx->upos, (size_t)T * B));
  // ---- masks (bit-packed, +1 word slack)
  ctx->mcount[RAU_MASK_WE] = (size_t)T * B * E;
  ctx->mcount[RAU_MASK_RNN] = (size_t)T * B * Rq;
  ctx->mcount[RAU_MASK_Q] = (size_t)H * B * Q;
  ctx->mcount[RAU_MASK_X] = (size_t)H * B * D * SL;
  ctx->mcount[RAU_MASK_MF] = (size_t)H * B * M;
  for (int i = 0; i < 5; ++i) CK(dalloc(ctx, &ctx->mbits[i], (ctx->mcount[i] + 31) / 32 + 1));
  // ---- encoder
  const size_t TB = (size_t)T * B;
  CK(dalloc(ctx, &ctx->we, TB * E));
  CK(dalloc(ctx, &ctx->G1, TB * 4 * Rq));
  CK(dalloc(ctx, &ctx->G2, TB * 4 * Rq));
  CK(dalloc(ctx, &ctx->c1, (TB + B) * Rq));
  CK(dalloc(ctx, &ctx->h1, (TB + B) * Rq));
  CK(dalloc(ctx, &ctx->c2, (TB + B) * Rq));
  CK(dalloc(ctx, &ctx->h2, (TB + B) * Rq));
  CK(dalloc(ctx, &ctx->tc1, TB * Rq));
  CK(dalloc(ctx, &ctx->tc2, TB * Rq));
  CK(dalloc(ctx, &ctx->x2, TB * Rq));
  CK(dalloc(ctx, &ctx->q, (size_t)B * Q));
  // ---- RAU
  const size_t HB = (size_t)H * B;
  CK(dalloc(ctx, &ctx->qd, HB * Q));
  CK(dalloc(ctx, &ctx->Yq, HB * M));
  CK(dalloc(ctx, &ctx->qf, HB * M));
  CK(dalloc(ctx, &ctx->xd, HB * D * S));
  // bf16 mode on maps the fused-dZ backward handles (the only backward that reads the bf16 form)
  if (ctx->bf16 == 1 && ctx->Sp == c.S && conv_dz_fused_ok(S, M, 1) && !std::getenv("RAU_XD_F32")) {
    float* x16 = nullptr;
    CK(dalloc(ctx, &x16, (HB * D * S + 1) / 2));
    ctx->xd16 = x16;
    float *w16 = nullptr, *p16 = nullptr;
    CK(dalloc(ctx, &w16, ((size_t)M * D + 1) / 2));
    CK(dalloc(ctx, &p16, ((size_t)A * M + 1) / 2));
    ctx->WiT16 = w16;
    ctx->WpT16 = p16;
    if (dgrad16_ok(M, A, S, M) && att_bwd_dma_ok(M, A, S, 8)) {
      float* d16 = nullptr;
      CK(dalloc(ctx, &d16, (HB * A * S + 1) / 2));
      ctx->dS16 = d16;
    }
  }
  CK(dalloc(ctx, &ctx->I, HB * M * S));
  CK(dalloc(ctx, &ctx->T, HB * A * S));
  CK(dalloc(ctx, &ctx->u, HB * A));   // finished u = qf Wa^T + ba per hop (the backward's tanh(P + u))
  CK(dalloc(ctx, &ctx->P0, (size_t)B * A * S));
  CK(dalloc(ctx, &ctx->WiT, (size_t)M * D));
  CK(dalloc(ctx, &ctx->WpT, (size_t)A * M));
  CK(dalloc(ctx, &ctx->zm, (size_t)B * S));
  CK(dalloc(ctx, &ctx->att_part, att_split_part_floats(B, S)));
  // Same-box A/B (B=256, D=512, ms/step): fused 16-wave attention + split-K encoder 10.54-10.59,
  // split attention + fused encoder step 10.92-10.95, round-1 library 10.75-10.82.  The split
  // attention kernels and the fused LSTM step are faster ALONE (no bulk GEMM beside them): the
  // evaluate-mode forward uses the fused LSTM step; RAU_ATT_SPLIT forces the split attention kernels.
  // Small batches (one 16-wave workgroup per sample leaves most CUs idle): B = 32 / 64 / 128 fused
  // 3.83 / 4.57 / 6.06 ms, split in 8 row chunks 3.63 / 4.39 / 6.05 ms -> split up to B = 64
  // (RAU_ATT_FUSED keeps the fused kernels).
  ctx->att_split_env = std::getenv("RAU_ATT_SPLIT") != nullptr ||
                       (c.B <= 64 && std::getenv("RAU_ATT_FUSED") == nullptr);
  {
    // Weight-stationary persistent encoder (enc_ws.hip): chosen by shape, never by the environment in
    // normal use -- contexts of up to 64 samples (the strong-scaling shards of configs[3]) are bound
    // by the recurrence's launches.  RAU_ENC_WS=0|1 is the A/B override.
    const char* e = std::getenv("RAU_ENC_WS");
    // measured in the step (MS weights, same box): B = 16 / 32 / 64 -> 3.00 / 3.49 / 4.46 ms with it,
    // 3.16 / 3.52 / 4.26 without; evaluate-mode forward 14.3 / 26.8 / 45.9 k vs 10.8 / 20.8 / 40.4 k QA/s.
    // Every workgroup reads all h rows of its sample half each step, so the traffic grows with B while
    // the weights it avoids re-reading do not: training contexts up to 32 samples, inference up to 64.
    // Its workgroups wait on each other's progress counters, so the whole grid must be resident at
    // once: asked of THIS device (a CPX partition or a reduced-CU device says no and keeps the
    // launch-per-step path).  If a bounded wait still gives up at run time (several contexts
    // competing for the CUs), persist_check() below turns the path off for the ctx.
    const bool fits = enc_ws_ok(B, Rq) && enc_ws_fits_device(B);
    ctx->enc_ws_train = fits && (e ? std::atoi(e) != 0 : B <= 32);
    if (const char* ss = std::getenv("RAU_SIDE_SPLIT")) ctx->side_split_env = std::atoi(ss) != 0;
    ctx->enc_ws = fits && (e ? std::atoi(e) != 0 : B <= 64);
    float* f = nullptr;
    CK(dalloc(ctx, &f, 16));
    ctx->ws_cnt = reinterpret_cast<unsigned*>(f);
  }
  {
    float* f = nullptr;
    CK(dalloc(ctx, &f, 4));   // device error word of the persistent encoder (a bounded spin gave up)
    ctx->perr_d = reinterpret_cast<int*>(f);
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->perr_h), sizeof(int), hipHostMallocDefault) != hipSuccess)
      ctx->perr_h = nullptr;
    else
      *ctx->perr_h = 0;
  }
  CK(dalloc(ctx, &ctx->a, HB * S));
  CK(dalloc(ctx, &ctx->jv, (size_t)B * M));
  CK(dalloc(ctx, &ctx->j, HB * M));
  CK(dalloc(ctx, &ctx->g4, HB * 4 * R));
  CK(dalloc(ctx, &ctx->cc, (HB + B) * R));
  CK(dalloc(ctx, &ctx->hh, (HB + B) * R));
  CK(dalloc(ctx, &ctx->tc, HB * R));
  CK(dalloc(ctx, &ctx->mf, HB * M));
  CK(dalloc(ctx, &ctx->logits, HB * K));
  CK(dalloc(ctx, &ctx->dl, HB * K));
  CK(dalloc(ctx, &ctx->lossrow, HB));
  CK(dalloc(ctx, &ctx->dopred, HB));
  CK(dalloc(ctx, &ctx->losses_d, (size_t)H));
  CK(dalloc(ctx, &ctx->hopw_d, (size_t)H));
  {  // ctx-owned pinned staging for the hop weights: the async upload never reads caller memory
    hipError_t eh = hipHostMalloc((void**)&ctx->hopw_h, (size_t)2 * H * sizeof(float), hipHostMallocDefault);
    if (eh != hipSuccess) {
      rau_destroy(ctx);
      return fail(RAU_ERR_NOMEM, "hipHostMalloc(hop weights): %s", hipGetErrorString(eh));
    }
  }
  CK(dalloc(ctx, &ctx->argmax_d, HB));
  // ---- backward
  CK(dalloc(ctx, &ctx->dpre, HB * M));
  CK(dalloc(ctx, &ctx->dhn, HB * R));   // dpre Wo for all hops
  CK(dalloc(ctx, &ctx->dg4, HB * 4 * R));
  for (int i = 0; i < 2; ++i) {
    CK(dalloc(ctx, &ctx->dcn[i], (size_t)B * R));
    CK(dalloc(ctx, &ctx->dhp[i], (size_t)B * R));
  }
  CK(dalloc(ctx, &ctx->dj, HB * M));
  CK(dalloc(ctx, &ctx->da_lin, (size_t)B * S));
  CK(dalloc(ctx, &ctx->dz, HB * S));
  CK(dalloc(ctx, &ctx->du, HB * A));
  CK(dalloc(ctx, &ctx->dwsp, HB * A));
  CK(dalloc(ctx, &ctx->dZ, HB * M * S));
  CK(dalloc(ctx, &ctx->dqt, HB * M));
  CK(dalloc(ctx, &ctx->dQD, HB * Q));
  CK(dalloc(ctx, &ctx->dq, (size_t)B * Q));
  {
    // split-K workspaces: one per stream (the two streams run concurrently)
    size_t sl2 = std::max(conv_wgrad_slab_floats(H * B, A, M, S),
                          conv_wgrad_slab_floats(H * B, M, D, S));
    if (ctx->bf16) sl2 = std::max(sl2, wgrad16_slab_floats(H * B, M, D, S));
    // The grouped Linear weight-gradient GEMMs' partial slabs.  They run on the weight-gradient stream
    // (slab3) or, where the bulk stream is the longer path, at the END of the bulk stream (rau_backward):
    // there they take the BULK stream's workspace -- each stream owns its workspace, and two launches that
    // share one must be ordered by their stream.  (Round 4 first moved the mult group's GEMM to the bulk
    // stream with slab3 still in its hands while the encoder group's GEMM used slab3 on the third stream:
    // the two ran concurrently and overwrote each other's partials; tests/test_gpu_att_variants.py caught it.)
    size_t grp_floats = 0;
    {
      TnProblem pr[13];
      int n = mult_wgrad_problems(ctx, pr);
      for (int hh = 1; hh <= H; ++hh)
        grp_floats = std::max(grp_floats, gemm_tn_group_slab_floats(pr, n, hh * B));
      n = enc_wgrad_problems(ctx, 0, pr);
      for (int tt = 1; tt <= T; ++tt)
        grp_floats = std::max(grp_floats, gemm_tn_group_slab_floats(pr, n, tt * B));
    }
    sl2 = std::max(sl2, grp_floats);
    CK(dalloc(ctx, &ctx->slab2, sl2));
    ctx->slab2_floats = sl2;
    const int rowsH = H * B, rowsT = T * B;
    const int shapes[][3] = {{K, M, rowsH},      {M, R, rowsH},      {4 * R, M, rowsH},
                             {4 * R, R, rowsH},  {M, S, rowsH},      {S, R, rowsH},
                             {A, M, rowsH},      {M, Q, rowsH},      {4 * Rq, E, rowsT},
                             {4 * Rq, Rq, rowsT}};
    // skinny split-K partials: every deferred GEMM of the hop loop must fit un-split in its share of
    // the slab (a quarter to a sixth), whatever its output width (4R, 4Rq, K, Q, M, A or S)
    size_t sl = (size_t)16 * B * std::max({4 * R, 4 * Rq, K, Q, M, A, S});
    for (auto& s : shapes) sl = std::max(sl, gemm_tn_slab_floats(s[0], s[1], s[2]));
    ctx->slab_floats = sl;
    CK(dalloc(ctx, &ctx->slab, sl));
    {  // the weight-gradient stream's workspace also holds the grouped launches' partial slabs
      const size_t sl3 = std::max(sl, grp_floats);
      ctx->slab3_floats = sl3;
      CK(dalloc(ctx, &ctx->slab3, sl3));
    }
    // every consumer of K-split partials checks the span it is about to read against these (split_guard.hip)
    split_ws_register(ctx->slab, ctx->slab_floats);
    split_ws_register(ctx->slab2, ctx->slab2_floats);
    split_ws_register(ctx->slab3, ctx->slab3_floats);
  }
  {
    const int widest = std::max({4 * R, 4 * Rq, K, M, S, A, Q});
    CK(dalloc(ctx, &ctx->coltmp3, (size_t)32 * widest));
    CK(dalloc(ctx, &ctx->coltmp2, (size_t)32 * widest));      // the bulk stream's column-sum scratch
    CK(dalloc(ctx, &ctx->dbi_part, (size_t)H * B * M));       // per-sample row sums of dZ
    CK(dalloc(ctx, &ctx->tmpS, (size_t)S));
  }
  CK(dalloc(ctx, &ctx->dG1, TB * 4 * Rq));
  CK(dalloc(ctx, &ctx->dG2, TB * 4 * Rq));
  CK(dalloc(ctx, &ctx->dwe, TB * E));
  for (int L = 0; L < 2; ++L) {
    CK(dalloc(ctx, &ctx->edc[L][0], (size_t)B * Rq));
    CK(dalloc(ctx, &ctx->edc[L][1], (size_t)B * Rq));
  }
  CK(dalloc(ctx, &ctx->dkey, (size_t)2));
  CK(dalloc(ctx, &ctx->npart, (size_t)1024));
  CK(dalloc(ctx, &ctx->norms_d, (size_t)4));
#undef CK
  {
    hipError_t es = hipStreamSynchronize(ctx->st);
    if (es != hipSuccess) {
      rau_destroy(ctx);
      return fail(RAU_ERR_DEVICE, "rau_create sync: %s", hipGetErrorString(es));
    }
  }
  *out = ctx;
  return RAU_OK;
}

void rau_destroy(rau_ctx* ctx) {
  if (!ctx) return;
  rau_comm_destroy(ctx);
  if (ctx->st_comm) hipStreamDestroy(ctx->st_comm);
  if (ctx->evC) hipEventDestroy(ctx->evC);
  if (ctx->st) hipStreamSynchronize(ctx->st);
  if (ctx->st2) hipStreamSynchronize(ctx->st2);
  if (ctx->st3) hipStreamSynchronize(ctx->st3);
  for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second);
  split_ws_unregister(ctx->slab);
  split_ws_unregister(ctx->slab2);
  split_ws_unregister(ctx->slab3);
  for (void* p : ctx->allocs) hipFree(p);
  if (ctx->hopw_h) hipHostFree(ctx->hopw_h);
  for (auto& r : ctx->precs) {
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  for (auto e : ctx->evpool) hipEventDestroy(e);
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  for (hipEvent_t e : {ctx->evA, ctx->evD, ctx->evW, ctx->evE, ctx->evW3, ctx->evM3, ctx->evEnd, ctx->evE1,
                       ctx->evG0, ctx->evG, ctx->evQ0, ctx->evQ, ctx->evDq})
    if (e) hipEventDestroy(e);
  for (hipEvent_t e : ctx->evGc)
    if (e) hipEventDestroy(e);
  if (ctx->st3) hipStreamDestroy(ctx->st3);
  if (ctx->perr_h) hipHostFree(ctx->perr_h);
  if (ctx->stc) { hipStreamSynchronize(ctx->stc); hipStreamDestroy(ctx->stc); }
  for (BatchSlot& s : ctx->slot) {
    if (s.feats_h) hipHostFree(s.feats_h);
    if (s.uploaded) hipEventDestroy(s.uploaded);
    if (s.consumed) hipEventDestroy(s.consumed);
  }
  for (hipEvent_t e : ctx->hopw_ev) if (e) hipEventDestroy(e);
  for (hipEvent_t e : ctx->evF) hipEventDestroy(e);
  for (hipEvent_t e : ctx->evK) hipEventDestroy(e);

  for (hipEvent_t e : ctx->evH) hipEventDestroy(e);
  if (ctx->evHd) hipEventDestroy(ctx->evHd);
  if (ctx->st2) hipStreamDestroy(ctx->st2);
  if (ctx->st) hipStreamDestroy(ctx->st);
  delete ctx;
}

// ------------------------------------------------------------- parameters
static int check_group(rau_ctx* ctx, int group) {
  NEED(ctx, "null ctx");
  NEED(group >= 0 && group < 3, "bad group %d", group);
  return 0;
}
int rau_params(rau_ctx* ctx, int group, float** weights, float** grads, size_t* n) {
  if (int rc = check_group(ctx, group)) return rc;
  if (weights) *weights = ctx->grp[group].w;
  if (grads) *grads = ctx->grp[group].g;
  if (n) *n = ctx->grp[group].n;
  return RAU_OK;
}
int rau_layout_count(const rau_ctx* ctx, int group) {
  if (!ctx || group < 0 || group > 2) return fail(RAU_ERR_INVALID, "bad ctx/group");
  return (int)ctx->grp[group].layout.size();
}
int rau_layout_entry(const rau_ctx* ctx, int group, int index, const char** name, size_t* offset,
                     int32_t* rows, int32_t* cols) {
  if (!ctx || group < 0 || group > 2) return fail(RAU_ERR_INVALID, "bad ctx/group");
  const auto& l = ctx->grp[group].layout;
  NEED(index >= 0 && index < (int)l.size(), "layout index %d out of range", index);
  if (name) *name = l[index].name.c_str();
  if (offset) *offset = l[index].off;
  if (rows) *rows = l[index].rows;
  if (cols) *cols = l[index].cols;
  return RAU_OK;
}
static int copy_group(rau_ctx* ctx, int group, bool grads, float* host, const float* chost,
                      size_t n) {
  if (int rc = check_group(ctx, group)) return rc;
  Group& g = ctx->grp[group];
  NEED(n == g.n, "group %d has %zu floats, caller passed %zu", group, g.n, n);
  float* dev = grads ? g.g : g.w;
  if (chost)
    HIPC(hipMemcpyAsync(dev, chost, n * sizeof(float), hipMemcpyHostToDevice, ctx->st));
  else
    HIPC(hipMemcpyAsync(host, dev, n * sizeof(float), hipMemcpyDeviceToHost, ctx->st));
  HIPC(hipStreamSynchronize(ctx->st));
  return RAU_OK;
}
int rau_set_params(rau_ctx* ctx, int group, const float* host, size_t n) {
  NEED(host, "null host pointer");
  return copy_group(ctx, group, false, nullptr, host, n);
}
int rau_get_params(rau_ctx* ctx, int group, float* host, size_t n) {
  NEED(host, "null host pointer");
  return copy_group(ctx, group, false, host, nullptr, n);
}
int rau_get_grads(rau_ctx* ctx, int group, float* host, size_t n) {
  NEED(host, "null host pointer");
  return copy_group(ctx, group, true, host, nullptr, n);
}
int rau_set_grads(rau_ctx* ctx, int group, const float* host, size_t n) {
  NEED(host, "null host pointer");
  return copy_group(ctx, group, true, nullptr, host, n);
}
int rau_init_uniform(rau_ctx* ctx, uint64_t seed, float lo, float hi) {
  NEED(ctx, "null ctx");
  for (int gi = 0; gi < 3; ++gi)
    RUN("uniform_fill", 0, ctx->grp[gi].n * 4.0,
        uniform_fill(ctx->st, seed, (uint32_t)gi, ctx->grp[gi].n, lo, hi, ctx->grp[gi].w));
  return RAU_OK;
}
int rau_zero_grads(rau_ctx* ctx) {
  NEED(ctx, "null ctx");
  for (int gi = 0; gi < 3; ++gi)
    HIPC(hipMemsetAsync(ctx->grp[gi].g, 0, ctx->grp[gi].n * sizeof(float), ctx->st));
  return RAU_OK;
}

// ---------------------------------------------------------------- dropout
int rau_set_mode(rau_ctx* ctx, int mode) {
  NEED(ctx, "null ctx");
  NEED(mode == RAU_MODE_TRAIN || mode == RAU_MODE_EVAL, "bad mode %d", mode);
  ctx->mode = mode;
  return RAU_OK;
}
int rau_set_dropout_seed(rau_ctx* ctx, uint64_t seed, uint32_t step) {
  NEED(ctx, "null ctx");
  ctx->seed = seed;
  ctx->step = step;
  for (int i = 0; i < 5; ++i) ctx->mexplicit[i] = false;
  ctx->mod_masks_valid = false;
  // device copy for fill_masks (pageable source: staged at call time, ordered on the ctx stream)
  const uint64_t key[2] = {seed, (uint64_t)step};
  HIPC(hipMemcpyAsync(ctx->dkey, key, sizeof(key), hipMemcpyHostToDevice, ctx->st));
  return RAU_OK;
}
int rau_set_mask(rau_ctx* ctx, int site, const uint8_t* keep, size_t n) {
  NEED(ctx && keep, "null argument");
  NEED(site >= 0 && site < 5, "bad mask site %d", site);
  NEED(n == ctx->mcount[site], "mask site %d has %zu elements, caller passed %zu", site,
       ctx->mcount[site], n);
  std::vector<uint32_t> bits((n + 31) / 32, 0u);
  for (size_t i = 0; i < n; ++i)
    if (keep[i]) bits[i >> 5] |= 1u << (i & 31);
  HIPC(hipMemcpyAsync(ctx->mbits[site], bits.data(), bits.size() * 4, hipMemcpyHostToDevice,
                      ctx->st));
  HIPC(hipStreamSynchronize(ctx->st));
  ctx->mexplicit[site] = true;
  ctx->mod_masks_valid = false;
  return RAU_OK;
}
// only: -1 = every site on the ctx stream; rau_forward fills the feature-map site (by far the largest,
// and read by the bulk stream only) on the bulk stream instead: skip = RAU_MASK_X, then only = RAU_MASK_X
static int gen_masks(rau_ctx* ctx, int skip = -1, int only = -1, hipStream_t stream = nullptr) {
  if (ctx->mode != RAU_MODE_TRAIN) return 0;
  hipStream_t s = stream ? stream : ctx->st;
  for (int i = 0; i < 5; ++i)
    if (i != skip && (only < 0 || i == only) && !ctx->mexplicit[i] && ctx->mp[i] > 0.f)
      RUNS(s, "fill_masks", 0, ctx->mcount[i] / 8.0,
           fill_masks(s, ctx->seed, (uint32_t)i, ctx->step, ctx->mp[i], ctx->mcount[i],
                      ctx->mbits[i], ctx->dkey));
  return 0;
}
int rau_get_mask(rau_ctx* ctx, int site, uint8_t* keep, size_t n) {
  NEED(ctx && keep, "null argument");
  NEED(site >= 0 && site < 5, "bad mask site %d", site);
  NEED(n == ctx->mcount[site], "mask site %d has %zu elements, caller passed %zu", site,
       ctx->mcount[site], n);
  if (!ctx->mexplicit[site]) {
    const int m = ctx->mode;
    ctx->mode = RAU_MODE_TRAIN;
    const int rc = gen_masks(ctx);
    ctx->mode = m;
    if (rc) return rc;
  }
  std::vector<uint32_t> bits((n + 31) / 32);
  HIPC(hipMemcpyAsync(bits.data(), ctx->mbits[site], bits.size() * 4, hipMemcpyDeviceToHost,
                      ctx->st));
  HIPC(hipStreamSynchronize(ctx->st));
  for (size_t i = 0; i < n; ++i) keep[i] = (bits[i >> 5] >> (i & 31)) & 1u;
  return RAU_OK;
}

// ------------------------------------------------------------------ batch
}  // extern "C"

namespace {
// Host-side half of a batch hand-over: argument checks and the distinct-token index over the live
// positions (t < lens[b]) that makes the LookupTable gradient a fixed-order gather-sum.
// utok / ustart / upos must hold T*B, T*B + 1, T*B entries.
int index_batch(const rau_config& c, const int32_t* tokens, const int32_t* lens, const int32_t* labels,
                int32_t* utok, int32_t* ustart, int32_t* upos, int* max_len_out, int* nuniq_out) {
  int max_len = 0;
  for (int b = 0; b < c.B; ++b) {
    NEED(lens[b] >= 0 && lens[b] <= c.T, "lens[%d]=%d out of [0,%d]", b, lens[b], c.T);
    max_len = std::max(max_len, lens[b]);
  }
  for (size_t i = 0; i < (size_t)c.T * c.B; ++i)
    NEED(tokens[i] >= 1 && tokens[i] <= c.V, "token %d at %zu out of [1,%d]", tokens[i], i, c.V);
  if (labels)
    for (int b = 0; b < c.B; ++b)
      NEED(labels[b] >= 1 && labels[b] <= c.K, "labels[%d]=%d out of [1,%d]", b, labels[b], c.K);
  std::vector<std::pair<int32_t, int32_t>> pos;  // (token, position)
  for (int t = 0; t < max_len; ++t)
    for (int b = 0; b < c.B; ++b)
      if (t < lens[b]) pos.push_back({tokens[(size_t)t * c.B + b], t * c.B + b});
  std::sort(pos.begin(), pos.end());
  const size_t TB = (size_t)c.T * c.B;
  size_t nu = 0;
  for (size_t i = 0; i < pos.size(); ++i) {
    if (i == 0 || pos[i].first != pos[i - 1].first) {
      utok[nu] = pos[i].first;
      ustart[nu] = (int32_t)i;
      ++nu;
    }
    upos[i] = pos[i].second;
  }
  // pad to the maximum token count: a graph-captured embed_bwd launches T*B blocks, the surplus
  // ones see an empty range
  for (size_t i = nu; i < TB; ++i) utok[i] = 1;
  for (size_t i = nu; i <= TB; ++i) ustart[i] = (int32_t)pos.size();
  for (size_t i = pos.size(); i < TB; ++i) upos[i] = 0;
  *max_len_out = max_len;
  *nuniq_out = (int)nu;
  return RAU_OK;
}

// H2D copies of one batch into a set of device buffers, enqueued on `s`
int enqueue_batch(rau_ctx* ctx, hipStream_t s, const BatchSlot& d, const float* feats,
                  const int32_t* tokens, const int32_t* lens, const int32_t* labels,
                  const int32_t* utok, const int32_t* ustart, const int32_t* upos) {
  const rau_config& c = ctx->cfg;
  const size_t TB = (size_t)c.T * c.B;
  if (feats && ctx->Sp == c.S)   // dense on both sides: one linear copy (a DMA-engine transfer, no blit kernel)
    HIPC(hipMemcpyAsync(d.feats, feats, (size_t)c.B * c.D * c.S * sizeof(float), hipMemcpyHostToDevice, s));
  else if (feats)   // rows of S positions into rows of Sp (pad columns stay zero)
    HIPC(hipMemcpy2DAsync(d.feats, (size_t)ctx->Sp * sizeof(float), feats, (size_t)c.S * sizeof(float),
                          (size_t)c.S * sizeof(float), (size_t)c.B * c.D, hipMemcpyHostToDevice, s));
  HIPC(hipMemcpyAsync(d.tokens, tokens, TB * 4, hipMemcpyHostToDevice, s));
  HIPC(hipMemcpyAsync(d.lens_d, lens, (size_t)c.B * 4, hipMemcpyHostToDevice, s));
  if (labels) HIPC(hipMemcpyAsync(d.labels_d, labels, (size_t)c.B * 4, hipMemcpyHostToDevice, s));
  HIPC(hipMemcpyAsync(d.utok, utok, TB * 4, hipMemcpyHostToDevice, s));
  HIPC(hipMemcpyAsync(d.upos, upos, TB * 4, hipMemcpyHostToDevice, s));
  HIPC(hipMemcpyAsync(d.ustart, ustart, (TB + 1) * 4, hipMemcpyHostToDevice, s));
  return RAU_OK;
}

void make_current(rau_ctx* ctx, int si) {
  BatchSlot& s = ctx->slot[si];
  ctx->cur_slot = si;
  ctx->feats = s.feats; ctx->tokens = s.tokens; ctx->lens_d = s.lens_d; ctx->labels_d = s.labels_d;
  ctx->utok = s.utok; ctx->ustart = s.ustart; ctx->upos = s.upos;
  ctx->lens_h = s.lens;
  ctx->max_len = s.max_len;
  ctx->nuniq = s.nuniq;
  ctx->have_batch = s.have;
  ctx->have_labels = s.have_labels;
  ctx->fwd_done = false;
}

// second set of device buffers, pinned staging for both slots, copy stream, events
int ensure_async(rau_ctx* ctx) {
  if (ctx->async_ready) return RAU_OK;
  const rau_config& c = ctx->cfg;
  const size_t TB = (size_t)c.T * c.B, nf = (size_t)c.B * c.D * c.S;
  BatchSlot& s0 = ctx->slot[0];
  s0.feats = ctx->feats; s0.tokens = ctx->tokens; s0.lens_d = ctx->lens_d; s0.labels_d = ctx->labels_d;
  s0.utok = ctx->utok; s0.ustart = ctx->ustart; s0.upos = ctx->upos;
  s0.lens = ctx->lens_h; s0.max_len = ctx->max_len; s0.nuniq = ctx->nuniq;
  s0.have = ctx->have_batch; s0.have_labels = ctx->have_labels;
  BatchSlot& s1 = ctx->slot[1];
  if (int rc = dalloc(ctx, &s1.feats, (size_t)c.B * c.D * ctx->Sp)) return rc;
  if (int rc = dalloc(ctx, &s1.tokens, TB)) return rc;
  if (int rc = dalloc(ctx, &s1.lens_d, (size_t)c.B)) return rc;
  if (int rc = dalloc(ctx, &s1.labels_d, (size_t)c.B)) return rc;
  if (int rc = dalloc(ctx, &s1.utok, TB)) return rc;
  if (int rc = dalloc(ctx, &s1.ustart, TB + 1)) return rc;
  if (int rc = dalloc(ctx, &s1.upos, TB)) return rc;
  for (BatchSlot& s : ctx->slot) {
    // one pinned block per slot: feats | tokens | lens | labels | utok | ustart | upos
    const size_t words = nf + TB + 2 * (size_t)c.B + TB + (TB + 1) + TB;
    void* h = nullptr;
    hipError_t e = hipHostMalloc(&h, words * 4, hipHostMallocDefault);
    if (e != hipSuccess) return fail(RAU_ERR_NOMEM, "hipHostMalloc(batch staging, %zu bytes): %s", words * 4,
                                     hipGetErrorString(e));
    s.feats_h = static_cast<float*>(h);
    s.tokens_h = reinterpret_cast<int32_t*>(s.feats_h + nf);
    s.lens_p = s.tokens_h + TB;
    s.labels_h = s.lens_p + c.B;
    s.utok_h = s.labels_h + c.B;
    s.ustart_h = s.utok_h + TB;
    s.upos_h = s.ustart_h + TB + 1;
    HIPC(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
  }
  int plo = 0, phi = 0;
  hipDeviceGetStreamPriorityRange(&plo, &phi);
  HIPC(hipStreamCreateWithPriority(&ctx->stc, hipStreamNonBlocking, plo));
  ctx->async_ready = true;
  return RAU_OK;
}
}  // namespace

extern "C" {

int rau_set_batch(rau_ctx* ctx, const float* feats, const int32_t* tokens, const int32_t* lens,
                  const int32_t* labels) {
  NEED(ctx && tokens && lens, "null argument");
  const rau_config& c = ctx->cfg;
  const size_t TB = (size_t)c.T * c.B;
  std::vector<int32_t> utok(TB), ustart(TB + 1), upos(TB);
  int max_len = 0, nuniq = 0;
  if (int rc = index_batch(c, tokens, lens, labels, utok.data(), ustart.data(), upos.data(), &max_len, &nuniq))
    return rc;
  BatchSlot d;   // the CURRENT device buffers (slot 0 unless rau_use_batch switched)
  d.feats = ctx->feats; d.tokens = ctx->tokens; d.lens_d = ctx->lens_d; d.labels_d = ctx->labels_d;
  d.utok = ctx->utok; d.ustart = ctx->ustart; d.upos = ctx->upos;
  if (ctx->async_ready && ctx->slot[ctx->cur_slot].upload_pending)   // an async upload into the same buffers
    HIPC(hipStreamWaitEvent(ctx->st, ctx->slot[ctx->cur_slot].uploaded, 0));
  if (int rc = enqueue_batch(ctx, ctx->st, d, feats, tokens, lens, labels, utok.data(), ustart.data(),
                             upos.data()))
    return rc;
  HIPC(hipStreamSynchronize(ctx->st));   // the caller's (pageable) buffers are free on return
  ctx->lens_h.assign(lens, lens + c.B);
  ctx->max_len = max_len;
  ctx->nuniq = nuniq;
  ctx->have_batch = true;
  ctx->have_labels = labels != nullptr;
  ctx->fwd_done = false;
  if (ctx->async_ready) {
    BatchSlot& s = ctx->slot[ctx->cur_slot];
    s.lens = ctx->lens_h; s.max_len = max_len; s.nuniq = nuniq;
    s.have = true; s.have_labels = ctx->have_labels; s.upload_pending = false;
  }
  return RAU_OK;
}

int rau_batch_slot(rau_ctx* ctx, int slot, float** feats_host, int32_t** tokens_host,
                   int32_t** lens_host, int32_t** labels_host) {
  NEED(ctx, "null ctx");
  NEED(slot == 0 || slot == 1, "rau_batch_slot: slot %d (0 or 1)", slot);
  if (int rc = ensure_async(ctx)) return rc;
  BatchSlot& s = ctx->slot[slot];
  if (s.upload_pending) {   // the caller is about to overwrite the staging: its last copy must have left
    HIPC(hipEventSynchronize(s.uploaded));
    s.upload_pending = false;
  }
  if (feats_host) *feats_host = s.feats_h;
  if (tokens_host) *tokens_host = s.tokens_h;
  if (lens_host) *lens_host = s.lens_p;
  if (labels_host) *labels_host = s.labels_h;
  return RAU_OK;
}

int rau_set_batch_async(rau_ctx* ctx, int slot, const float* feats, const int32_t* tokens,
                        const int32_t* lens, const int32_t* labels, int has_labels) {
  NEED(ctx, "null ctx");
  NEED(slot == 0 || slot == 1, "rau_set_batch_async: slot %d (0 or 1)", slot);
  if (int rc = ensure_async(ctx)) return rc;
  const rau_config& c = ctx->cfg;
  BatchSlot& s = ctx->slot[slot];
  if (slot == ctx->cur_slot && ctx->fwd_done)
    return fail(RAU_ERR_STATE, "rau_set_batch_async: slot %d is the current batch of a forward pass whose "
                "backward has not run; upload into the other slot", slot);
  const size_t TB = (size_t)c.T * c.B, nf = (size_t)c.B * c.D * c.S;
  const bool copies = (feats && feats != s.feats_h) || (tokens && tokens != s.tokens_h) ||
                      (lens && lens != s.lens_p) || (labels && labels != s.labels_h);
  (void)copies;
  // The slot's previous upload may not have left its pinned staging yet: index_batch below rewrites the
  // pinned index arrays in every case, and the memcpys rewrite the rest, so wait for it either way.
  // (A caller that refills the staging IN PLACE must call rau_batch_slot(slot) before every refill:
  // that call performs the same wait before the caller's own writes -- include/rau.h.)
  if (s.upload_pending) {
    HIPC(hipEventSynchronize(s.uploaded));
    s.upload_pending = false;
  }
  // NULL = the caller has filled the slot's pinned staging in place (rau_batch_slot)
  if (feats && feats != s.feats_h) std::memcpy(s.feats_h, feats, nf * 4);
  if (tokens && tokens != s.tokens_h) std::memcpy(s.tokens_h, tokens, TB * 4);
  if (lens && lens != s.lens_p) std::memcpy(s.lens_p, lens, (size_t)c.B * 4);
  if (labels && labels != s.labels_h) std::memcpy(s.labels_h, labels, (size_t)c.B * 4);
  const bool with_labels = labels != nullptr || has_labels != 0;
  int max_len = 0, nuniq = 0;
  if (int rc = index_batch(c, s.tokens_h, s.lens_p, with_labels ? s.labels_h : nullptr, s.utok_h, s.ustart_h,
                           s.upos_h, &max_len, &nuniq))
    return rc;
  // device side: the slot's buffers may still be read by the last step that used them
  if (slot == ctx->cur_slot) {
    HIPC(hipEventRecord(s.consumed, ctx->st));
    s.consumed_valid = true;
  }
  if (s.consumed_valid) HIPC(hipStreamWaitEvent(ctx->stc, s.consumed, 0));
  if (int rc = enqueue_batch(ctx, ctx->stc, s, s.feats_h, s.tokens_h, s.lens_p,
                             with_labels ? s.labels_h : nullptr, s.utok_h, s.ustart_h, s.upos_h))
    return rc;
  HIPC(hipEventRecord(s.uploaded, ctx->stc));
  s.upload_pending = true;
  s.lens.assign(s.lens_p, s.lens_p + c.B);
  s.max_len = max_len;
  s.nuniq = nuniq;
  s.have = true;
  s.have_labels = with_labels;
  if (slot == ctx->cur_slot) {   // re-filled in place: the next forward waits for the copies
    make_current(ctx, slot);
    HIPC(hipStreamWaitEvent(ctx->st, s.uploaded, 0));
  }
  return RAU_OK;
}

int rau_use_batch(rau_ctx* ctx, int slot) {
  NEED(ctx, "null ctx");
  NEED(slot == 0 || slot == 1, "rau_use_batch: slot %d (0 or 1)", slot);
  if (int rc = ensure_async(ctx)) return rc;
  BatchSlot& s = ctx->slot[slot];
  if (!s.have) return fail(RAU_ERR_STATE, "rau_use_batch: slot %d holds no batch (rau_set_batch_async)", slot);
  if (slot != ctx->cur_slot) {
    // everything enqueued so far may still read the slot we are leaving (the forward's bulk work is
    // joined into the chain stream by its hop events, the backward's by its end-of-step joins)
    BatchSlot& p = ctx->slot[ctx->cur_slot];
    HIPC(hipEventRecord(p.consumed, ctx->st));
    p.consumed_valid = true;
  }
  make_current(ctx, slot);
  // the bulk and weight-gradient streams start each step behind an event of the chain stream,
  // so ordering the chain stream behind the upload orders all three
  HIPC(hipStreamWaitEvent(ctx->st, s.uploaded, 0));
  return RAU_OK;
}

int rau_batch_feats(rau_ctx* ctx, float** feats_dev) {
  NEED(ctx && feats_dev, "null argument");
  *feats_dev = ctx->feats;
  return RAU_OK;
}

}  // extern "C"

// ================================================================ one hop
// The recurrence-dependent half of one answering hop (SS:292-307 given this hop's I and
// P = Wp I + bp), in two parts:
//   hop_forward_chain -- what the NEXT hop has to wait for: q_embed's recurrent half, attention,
//     the attention LSTM (8 dependent launches; next_c / next_h land in c_out / h_out);
//   hop_forward_head  -- what nothing on the recurrence waits for: merge_feat, out_score,
//     out_do_pred and the criterion head (SS:277-283, 488, 518), batched over `nh` consecutive
//     hops ([nh*B] rows) on any stream with its own split-K workspace.
// rau_forward runs the heads on the weight-gradient stream (idle during the forward pass), so the
// hop-to-hop critical path is the chain alone; the module-level entry points run both on st.
int hop_forward_chain(rau_ctx* ctx, int h, const float* cp, const float* hp, float* c_out,
                      float* h_out, const float* Ih, const float* Pin) {
  const rau_config& c = ctx->cfg;
  const int B = c.B, S = ctx->Sp, SL = c.S, M = c.M, A = c.A, R = c.R;
  hipStream_t st = ctx->st;
  auto gflop = [](double m, double n, double k) { return 2.0 * m * n * k; };
  const size_t BM_ = (size_t)B * M, BS_ = (size_t)B * S, BR_ = (size_t)B * R;
  float* qf = ctx->qf + (size_t)h * BM_;
  float* ah = ctx->a + (size_t)h * BS_;
  float* jh = ctx->j + (size_t)h * BM_;
  float* g4 = ctx->g4 + (size_t)h * B * 4 * R;
  // Split-K partials of several of these GEMMs are summed by their CONSUMER kernel instead of a
  // reduce launch (each launch on this dependent chain costs 8-20 us next to the bulk GEMMs):
  // the slab is carved into four regions so that partials can stay alive side by side.
  const size_t reg = ctx->slab_floats / 4;
  float *slab_u = ctx->slab + reg, *slab_t = ctx->slab + 2 * reg;
  int ns_u = 0, ns_t = 0, ns_i = 0;
  size_t off[3];
  {  // the three Linears fed by h_prev alone -- q_embed's recurrent half (SS:234), attbymemory
     // (SS:287) and the attention LSTM's h2h (ATTLSTM.lua:7) -- in ONE launch
    const float* Wt[3] = {ctx->h_proj.W, ctx->att_mem.W, ctx->lstm_h2h.W};
    const int Nt[3] = {M, SL, 4 * R};
    RUN("small_gemm", gflop(B, M + SL + 4 * R, R), 0,
        gemm_nt_hetero_deferred(st, 3, B, R, hp, R, Wt, R, Nt, slab_t, reg + reg / 2, &ns_t, off));
  }
  float *slab_z = slab_t + off[1], *slab_g = slab_t + off[2];
  {  // qf = tanh(Yq + h_prev Wh^T): the q half (with both biases) was computed for all hops at once
    LinOpts o;
    o.addend = ctx->Yq + (ctx->yq_shared ? 0 : (size_t)h * BM_);
    o.add_rs = M;
    o.act = 1;
    RUN("lin_reduce", 0, 0, lin_reduce_epilogue(st, B, M, ns_t, slab_t + off[0], qf, M, o));
  }
  {  // attbycontent SS:244-252: u = qf Wa^T (+ ba inside att_fwd_fused)
    LinOpts o;
    o.slab = slab_u; o.slab_floats = reg; o.defer_splits = &ns_u;
    RUN("small_gemm", gflop(B, A, M), 0, gemm_nt(st, B, A, M, qf, M, ctx->att_q.W, M, ctx->u, A, o));
  }
  const int ns_z = ns_t, ns_h = ns_t;
  // tanh(P+u), score, softmax, attention-weighted sum: one pass per sample
  {
    AttPartials ap;
    ap.u_ns = ns_u; ap.u_bias = ctx->att_q.b;
    ap.z_ns = ns_z; ap.z_bias = ctx->att_mem.b;
    ap.SL = SL;
    ap.u_out = ctx->u + (size_t)h * B * A;   // tanh(P + u) itself is not kept: 25 % less traffic here
    // evaluate mode: no conv tile is resident while the hops run (I and P are hoisted), so the 16-wave form
    // fits and streams a sample faster (B = 256: 124.2 -> 129.5 k QA/s); the training step keeps 8
    ap.waves = ctx->mode == RAU_MODE_EVAL ? 16 : 0;
    if (!(ctx->att_split_env))
      RUN("att_fwd_fused", 2.0 * B * S * (A + M), ((double)B * A * S + BM_ * S) * 4,
          att_fwd_fused(st, B, M, A, S, Pin, slab_u, ctx->att_score.W, ctx->att_score.b, slab_z, Ih, qf,
                        nullptr, ah, ctx->jv, ap));
    else
      RUN("att_fwd_split", 2.0 * B * S * (A + M), ((double)B * A * S + BM_ * S) * 4,
          att_fwd_split(st, B, M, A, S, Pin, slab_u, ctx->att_score.W, ctx->att_score.b, slab_z, Ih, qf,
                        ah, ctx->jv, ctx->att_part, ap));
  }
  {  // classifier SS:265-283
    LINOPTS(o);
    o.slab_floats = reg;
    o.bias = ctx->feat_attprob.b;
    o.addend = ctx->jv;
    o.add_rs = M;
    RUN("small_gemm", gflop(B, M, SL), 0, gemm_nt(st, B, M, SL, ah, S, ctx->feat_attprob.W, SL, jh, M, o));
  }
  {  // j Wx^T partials right behind the recurrent ones; the cell kernel sums both + both biases
    LinOpts o;
    o.slab = slab_g + (size_t)ns_h * B * 4 * R; o.slab_floats = reg / 2; o.defer_splits = &ns_i;
    RUN("small_gemm", gflop(B, 4 * R, M), 0,
        gemm_nt(st, B, 4 * R, M, jh, M, ctx->lstm_i2h.W, M, g4, 4 * R, o));
    LstmFwdCells cells{};
    cells.n = 1;
    LstmFwdCell& C = cells.c[0];
    C.g4 = g4; C.has_input = 0; C.b1 = ctx->lstm_i2h.b; C.b2 = ctx->lstm_h2h.b;
    C.slab = slab_g; C.nsplit = ns_h + ns_i;
    C.c_prev = cp; C.cp_rs = R; C.c = c_out; C.c_rs = R; C.h = h_out; C.h_rs = R;
    C.tanhc = ctx->tc + (size_t)h * BR_;
    C.drop_out = nullptr; C.mask = nullptr; C.mask_e0 = 0; C.mscale = 1.f;
    RUN("lstm_fwd", 0, BR_ * 4.0 * 10, lstm_fwd_multi(st, GATES_ATT, B, R, cells));
  }
  return RAU_OK;
}

// merge_feat = dropout(j + h' Wo^T + bo), logits = merge_feat Wc^T + bc, do_pred, cross-entropy +
// first-max argmax for hops [h0, h0 + nh): rows = nh * B.  h' rows are ctx->hh slots h0+1 .. (the
// chain's h_out), `ws` a split-K workspace of >= 2 regions of `reg` floats owned by stream `s`.
int hop_forward_head(rau_ctx* ctx, hipStream_t s, float* ws, size_t reg, int h0, int nh,
                     const int32_t* labels) {
  const rau_config& c = ctx->cfg;
  const int B = c.B, M = c.M, R = c.R, K = c.K;
  const bool tr = ctx->mode == RAU_MODE_TRAIN;
  const uint32_t* m_mf = (tr && mask_p(ctx, RAU_MASK_MF) > 0.f) ? ctx->mbits[RAU_MASK_MF] : nullptr;
  auto gflop = [](double m, double n, double k) { return 2.0 * m * n * k; };
  const size_t BM_ = (size_t)B * M, BR_ = (size_t)B * R;
  const int rows = nh * B;
  const float* hnew = ctx->hh + (size_t)(h0 + 1) * BR_;
  const float* jh = ctx->j + (size_t)h0 * BM_;
  float* mfh = ctx->mf + (size_t)h0 * BM_;
  float* lg = ctx->logits + (size_t)h0 * B * K;
  int ns_c = 0;
  {
    LinOpts o;
    o.slab = ws; o.slab_floats = reg;
    o.bias = ctx->lstm_out.b;
    o.addend = jh;
    o.add_rs = M;
    o.emask = m_mf;
    o.emask_e0 = (size_t)h0 * BM_;
    o.emscale = 1.f / (1.f - mask_p(ctx, RAU_MASK_MF));
    RUNS(s, "head_gemm", gflop(rows, M, R), 0,
         gemm_nt(s, rows, M, R, hnew, R, ctx->lstm_out.W, R, mfh, M, o));
  }
  {  // out_score SS:280: partials finished (+ bias) by the criterion-head kernel
    LinOpts o;
    o.slab = ws + reg; o.slab_floats = reg; o.defer_splits = &ns_c;
    RUNS(s, "head_gemm", gflop(rows, K, M), 0,
         gemm_nt(s, rows, K, M, mfh, M, ctx->cls.W, M, lg, K, o));
  }
  RUNS(s, "ce_fwd", 0, (double)rows * K * 12,
       ce_fwd(s, rows, K, M, lg, labels, mfh, ctx->do_pred.W, ctx->do_pred.b,
              ctx->dl + (size_t)h0 * B * K, ctx->lossrow + (size_t)h0 * B,
              ctx->argmax_d + (size_t)h0 * B, ctx->dopred + (size_t)h0 * B, ws + reg, ns_c,
              ctx->cls.b, lg, B));
  return RAU_OK;
}

int hop_forward(rau_ctx* ctx, int h, const float* cp, const float* hp, float* c_out, float* h_out,
                const float* Ih, const float* Pin, const int32_t* labels) {
  if (int rc = hop_forward_chain(ctx, h, cp, hp, c_out, h_out, Ih, Pin)) return rc;
  if (h_out != ctx->hh + (size_t)(h + 1) * ctx->cfg.B * ctx->cfg.R)
    return fail(RAU_ERR_INVALID, "hop_forward: h_out must be the ctx's hop slot");
  const size_t reg = ctx->slab_floats / 4;
  return hop_forward_head(ctx, ctx->st, ctx->slab, reg, h, 1, labels);
}

// Backward of hop_forward (hand-derived, SURVEY 8a "exact backward of one hop"): from the
// gradients at {logits, next_c, next_h} to dpre, dg4, dj, dz, dS (in T), du, dq~ in the
// hop-h slots and {dc_prev, dh_prev}.  Weight gradients and the 1x1-conv gradients are formed
// by the callers from those slots.
//
// What is on the recurrence's critical path is kept to 10 dependent launches:
//   * dpre = (dl Wc) (.) mask and dhn = dpre Wo do not depend on the recurrence; the step-level
//     caller forms them for all hops at once up front (g.dpre_ready / g.dhn_all);
//   * the three contributions to dh_prev (dg Wr, dz Wm, dq~ Wh) stay K-split partials, laid out
//     back to back, and are summed by the NEXT hop's lstm_bwd (g.dh_part_*), so none of them
//     costs a reduce launch; dj Wf's partials are summed inside att_bwd_fused;
//   * dg Wx and dg Wr share their A operand: one batched launch when M == R.
int hop_backward(rau_ctx* ctx, int h, const float* cp, const float* Ih, const HopGrad& g) {
  const rau_config& c = ctx->cfg;
  const int B = c.B, S = ctx->Sp, SL = c.S, M = c.M, A = c.A, R = c.R, K = c.K;
  hipStream_t st = ctx->st;
  const bool tr = ctx->mode == RAU_MODE_TRAIN;
  const uint32_t* m_mf = (tr && mask_p(ctx, RAU_MASK_MF) > 0.f) ? ctx->mbits[RAU_MASK_MF] : nullptr;
  auto sc = [&](int site) { return 1.f / (1.f - mask_p(ctx, site)); };
  auto gflop = [](double m, double n, double k) { return 2.0 * m * n * k; };
  const size_t BM_ = (size_t)B * M, BS_ = (size_t)B * S, BR_ = (size_t)B * R;
  const float* qf = ctx->qf + (size_t)h * BM_;
  float* Th = ctx->T + (size_t)h * B * A * S;   // becomes dS in place
  const float* ah = ctx->a + (size_t)h * BS_;
  const float* g4 = ctx->g4 + (size_t)h * B * 4 * R;
  float* dpre = ctx->dpre + (size_t)h * BM_;
  float* dg4 = ctx->dg4 + (size_t)h * B * 4 * R;
  float* djh = ctx->dj + (size_t)h * BM_;
  float* dzh = ctx->dz + (size_t)h * BS_;
  float* duh = ctx->du + (size_t)h * B * A;
  float* dqt = ctx->dqt + (size_t)h * BM_;
  float* dc_out = g.dc_out;
  // split-K workspace: region 0 = partials that are reduced right away (LINOPTS default) or
  // consumed by the next kernel; regions 1..3 = this hop's dj partials followed by the dh_prev
  // partials, which live until the next hop's lstm_bwd has read them
  const size_t reg = ctx->slab_floats / 4;
  float* X = ctx->slab + reg;
  const size_t Xcap = 3 * reg;
  if (!g.dpre_ready) {
    {  // dmf = dlogits Wc ; dpre = dmf (.) mask   (do_pred grad is zero, SS:566)
      LINOPTS(o);
      o.slab_floats = reg;
      o.addend = g.dmf_add;   // module-level callers: gradient through do_pred (zero in feval)
      o.add_rs = M;
      o.emask = m_mf;
      o.emask_e0 = (size_t)h * BM_;
      o.emscale = sc(RAU_MASK_MF);
      RUN("small_gemm", gflop(B, M, K), 0,
          gemm_nn(st, B, M, K, g.dl, K, ctx->cls.W, M, dpre, M, o));
    }
    // dhn = dpre Wo + dh_next, the K-split partials of dpre Wo summed inside lstm_bwd
    int nsp = 0;
    LINOPTS(o);
    o.slab_floats = reg;
    o.defer_splits = &nsp;
    RUN("small_gemm", gflop(B, R, M), 0, gemm_nn(st, B, R, M, dpre, M, ctx->lstm_out.W, R, ctx->dhn, R, o));
    RUN("lstm_bwd", 0, BR_ * 4.0 * 12,
        lstm_bwd(st, GATES_ATT, B, R, g4, cp, R, ctx->tc + (size_t)h * BR_, nullptr, R, g.dh_next,
                 g.dc_next, dg4, dc_out, nullptr, 0, nullptr, nullptr, 0, ctx->slab, nsp));
  } else {
    // dhn rows of this hop + the previous hop_backward's dh_prev partials (+ dh_next if given)
    RUN("lstm_bwd", 0, BR_ * 4.0 * 12,
        lstm_bwd(st, GATES_ATT, B, R, g4, cp, R, ctx->tc + (size_t)h * BR_,
                 g.dhn_all + (size_t)h * BR_, R, g.dh_next, g.dc_next, dg4, dc_out, nullptr, 0,
                 nullptr, nullptr, 0, g.dh_part, g.dh_part_ns));
  }
  // dj partials = dg Wx ; dh_prev partials #1 = dg Wr
  int ns_j = 0, ns_h = 0;
  float* dhp = nullptr;      // start of this hop's dh_prev partials [ns_h][B][R]
  const bool dead = g.dh_prev_dead;
  if (M == R && !dead) {
    const float* Ap[2] = {dg4, dg4};
    const float* Wp[2] = {ctx->lstm_i2h.W, ctx->lstm_h2h.W};
    RUN("small_gemm", 2 * gflop(B, M, 4 * R), 0,
        gemm_nn_batched_deferred(st, 2, B, M, 4 * R, Ap, 4 * R, Wp, M, X, Xcap / 2, &ns_j));
    dhp = X + (size_t)ns_j * BM_;
    ns_h = ns_j;
  } else if (dead) {
    LinOpts o1;
    o1.slab = X; o1.slab_floats = Xcap / 4; o1.defer_splits = &ns_j;
    RUN("small_gemm", gflop(B, M, 4 * R), 0,
        gemm_nn(st, B, M, 4 * R, dg4, 4 * R, ctx->lstm_i2h.W, M, djh, M, o1));
    dhp = X + (size_t)ns_j * BM_;
  } else {
    LinOpts o1;
    o1.slab = X; o1.slab_floats = Xcap / 4; o1.defer_splits = &ns_j;
    RUN("small_gemm", gflop(B, M, 4 * R), 0,
        gemm_nn(st, B, M, 4 * R, dg4, 4 * R, ctx->lstm_i2h.W, M, djh, M, o1));
    dhp = X + (size_t)ns_j * BM_;
    LinOpts o2;
    o2.slab = dhp; o2.slab_floats = Xcap / 4; o2.defer_splits = &ns_h;
    RUN("small_gemm", gflop(B, R, 4 * R), 0,
        gemm_nn(st, B, R, 4 * R, dg4, 4 * R, ctx->lstm_h2h.W, R, nullptr, R, o2));
  }
  {  // dj = dpre + sum of partials
    LinOpts o;
    o.addend = dpre;
    o.add_rs = M;
    RUN("lin_reduce", 0, 0, lin_reduce_epilogue(st, B, M, ns_j, X, djh, M, o));
  }
  int ns_a = 0;
  {  // da = dj Wf  (+ attselect term and the gradient at the attprob OUTPUT inside att_bwd_fused)
    LINOPTS(o);
    o.slab_floats = reg;
    o.defer_splits = &ns_a;
    RUN("small_gemm", gflop(B, SL, M), 0,
        gemm_nn(st, B, SL, M, djh, M, ctx->feat_attprob.W, SL, ctx->da_lin, S, o));
  }
  if (!(ctx->att_split_env))
    RUN("att_bwd_fused", 2.0 * B * S * (A + M), ((double)B * A * S * 2 + BM_ * S) * 4,
        att_bwd_fused(st, B, M, A, S, Ih, djh, ah, ctx->slab, ctx->att_score.W, Th, dzh, duh,
                      ctx->dwsp + (size_t)h * B * A, ctx->I_shared ? ctx->P0 : Th,
                      ctx->u + (size_t)h * B * A, ns_a, SL, g.da_out,
                      ctx->ds16_step ? (void*)((uint16_t*)ctx->dS16 + (size_t)h * B * A * S) : nullptr,
                      ctx->bf16 ? 8 : 0));
  else
    RUN("att_bwd_split", 2.0 * B * S * (A + M), ((double)B * A * S * 2 + BM_ * S) * 4,
        att_bwd_split(st, B, M, A, S, Ih, djh, ah, ctx->slab, ctx->att_score.W, Th, dzh, duh,
                      ctx->dwsp + (size_t)h * B * A, ctx->I_shared ? ctx->P0 : Th,
                      ctx->u + (size_t)h * B * A, ctx->att_part, ns_a, SL, g.da_out));
  if (g.ev_conv_ready) HIPC(hipEventRecord(g.ev_conv_ready, st));
  const size_t used = (size_t)(dhp - X);
  if (!dead) {  // dh_prev partials #2 = dz Wm
    int ns = 0;
    LinOpts o;
    o.slab = dhp + (size_t)ns_h * BR_;
    o.slab_floats = (Xcap - used) / 3;
    o.defer_splits = &ns;
    RUN("small_gemm", gflop(B, R, SL), 0, gemm_nn(st, B, R, SL, dzh, S, ctx->att_mem.W, R, nullptr, R, o));
    ns_h += ns;
  }
  {  // dq~ = (dj + du Wa) (1 - qf^2)
    LINOPTS(o);
    o.slab_floats = reg;
    o.addend = djh;
    o.add_rs = M;
    o.ymul = qf;
    o.y_rs = M;
    RUN("small_gemm", gflop(B, M, A), 0, gemm_nn(st, B, M, A, duh, A, ctx->att_q.W, M, dqt, M, o));
  }
  if (!dead) {  // dh_prev partials #3 = dq~ Wh
    int ns = 0;
    LinOpts o;
    o.slab = dhp + (size_t)ns_h * BR_;
    o.slab_floats = (Xcap - used) / 3;
    o.defer_splits = &ns;
    RUN("small_gemm", gflop(B, R, M), 0, gemm_nn(st, B, R, M, dqt, M, ctx->h_proj.W, R, nullptr, R, o));
    ns_h += ns;
  }
  if (g.dh_part_out) {   // the next hop_backward's lstm_bwd sums them
    *g.dh_part_out = dead ? nullptr : dhp;
    *g.dh_part_ns_out = dead ? 0 : ns_h;
  } else {               // module-level callers want dh_prev itself
    LinOpts o;
    RUN("lin_reduce", 0, 0, lin_reduce_epilogue(st, B, R, ns_h, dhp, g.dh_out, R, o));
  }
  return RAU_OK;
}

extern "C" {

// ================================================================ forward
// The forward / backward seam (the bulk stream and the matrix pipes wait there for the chain).  RAU_SEAM=<mask>
// (A/B; default 3): 1 = a train-mode forward with labels also forms the backward's two
// recurrence-free head products (dpre, dhn) behind each group's criterion head on the third stream;
// 2 = a group's conv gradients start behind att_bwd of its first hop instead of behind the hop's last launch.
static int seam_mask() {
  static const int m = [] { const char* e = std::getenv("RAU_SEAM"); return e ? std::atoi(e) : 3; }();
  return m;
}
static bool head_dgrad_fwd(const rau_ctx* ctx) {
  return ctx->have_labels && ctx->mode == RAU_MODE_TRAIN && (seam_mask() & 1);
}

int rau_forward(rau_ctx* ctx) {
  NEED(ctx, "null ctx");
  if (!ctx->have_batch) return fail(RAU_ERR_STATE, "rau_forward: no batch (call rau_set_batch)");
  set_skinny_policy(ctx);
  const rau_config& c = ctx->cfg;
  const int B = c.B, E = c.E, Rq = c.Rq, D = c.D, S = ctx->Sp, SL = c.S, M = c.M, A = c.A, R = c.R,
            H = c.H, Q = ctx->Q;
  const int TL = ctx->max_len;
  hipStream_t st = ctx->st;
  if (int rc = gen_masks(ctx, RAU_MASK_X)) return rc;
  const bool tr = ctx->mode == RAU_MODE_TRAIN;
  auto mk = [&](int site) -> const uint32_t* {
    return (tr && mask_p(ctx, site) > 0.f) ? ctx->mbits[site] : nullptr;
  };
  auto sc = [&](int site) { return 1.f / (1.f - mask_p(ctx, site)); };
  const uint32_t *m_we = mk(RAU_MASK_WE), *m_rnn = mk(RAU_MASK_RNN), *m_q = mk(RAU_MASK_Q),
                 *m_x = mk(RAU_MASK_X);
  const size_t BRq = (size_t)B * Rq;
  auto gflop = [](double m, double n, double k) { return 2.0 * m * n * k; };

  // ---------------- bulk stream: everything about the feature map that does not
  // depend on the recurrence (overlaps the encoder and the hop loop below).
  // i_embed SS:238-242: in train mode each hop clone has its own dropout mask on
  // the feature map (SS:239, SS:343-347) -> xd[h] = X (.) mask_h, then GEMMs over
  // groups of hops; in evaluate mode I is hop-invariant and computed once.
  // attbycontent's ifeatproj (SS:247-249) is hop-invariant given I: P = Wp I + bp;
  // the per-hop half (+u, tanh, score, softmax, context) is att_fwd_fused.
  // Hops are launched in groups of `hop_group` so hop h's chain can start as soon
  // as its group is done while the bulk stream works on the later groups.
  // ---------------- encoder, SS:443-462
  // Layer-1 cell t+1 and layer-2 cell t do not depend on each other, so the two layers
  // advance as a wavefront: per step ONE batched split-K GEMM (h1 W_h2h1^T for layer 1;
  // x2 W_i2h2^T and h2 W_h2h2^T for layer 2; same shapes) and ONE two-cell LSTM kernel
  // that sums the partials -- TL+1 steps of 2 launches instead of 2*TL steps of 2.
  auto encoder_forward = [&]() -> int {
  if (TL > 0) {
    const int rows = TL * B;
    const size_t G4 = (size_t)B * 4 * Rq;
    RUN("embed_fwd", 0, rows * E * 8.0,
        embed_fwd(st, rows, E, c.V, ctx->grp[RAU_GROUP_EMBED].w, ctx->tokens, m_we, sc(RAU_MASK_WE),
                  ctx->we));
    const bool ws_path = (ctx->mode == RAU_MODE_EVAL ? ctx->enc_ws : ctx->enc_ws_train) && ctx->perr_h;
    // Layer-1 input projection of every token (no recurrence in it).  Only the first tokens' rows
    // are needed at once: they stay on this stream, the rest runs on the weight-gradient stream
    // (idle in the forward pass) and the wavefront waits for it where it first reads it.
    const int t_head = (!ws_path && side_split(ctx) && TL > kEncHeadTokens) ? kEncHeadTokens : TL;
    {
      LINOPTS(o);
      o.bias = ctx->i2h[0].b;
      o.bias2 = ctx->h2h[0].b;
      RUN("enc_i2h_gemm", gflop(t_head * B, 4 * Rq, E), 0,
          gemm_nt(st, t_head * B, 4 * Rq, E, ctx->we, E, ctx->i2h[0].W, E, ctx->G1, 4 * Rq, o));
    }
    if (t_head < TL) {
      HIPC(hipEventRecord(ctx->evG0, st));            // embed_fwd is done
      HIPC(hipStreamWaitEvent(ctx->st3, ctx->evG0, 0));
      LinOpts o;
      o.slab = ctx->slab3; o.slab_floats = ctx->slab3_floats;
      o.bias = ctx->i2h[0].b;
      o.bias2 = ctx->h2h[0].b;
      // in chunks, so that the recurrence only ever waits for the rows it is about to read
      static const int nchunk_env = [] { const char* e = std::getenv("RAU_ENC_CHUNKS"); return e ? std::atoi(e) : 0; }();
      const int nchunk = std::max(1, std::min(nchunk_env > 0 ? nchunk_env : kEncSideChunks,
                                              std::min(kEncSideChunks, TL - t_head)));
      const int per = (TL - t_head + nchunk - 1) / nchunk;
      for (int cch = 0; cch <= kEncSideChunks; ++cch) ctx->enc_chunk_tok[cch] = TL;
      for (int cch = 0, t0 = t_head; t0 < TL; ++cch, t0 += per) {
        const int nt = std::min(per, TL - t0), r1 = nt * B;
        ctx->enc_chunk_tok[cch] = t0;
        RUNS(ctx->st3, "enc_i2h_gemm", gflop(r1, 4 * Rq, E), 0,
             gemm_nt(ctx->st3, r1, 4 * Rq, E, ctx->we + (size_t)t0 * B * E, E, ctx->i2h[0].W, E,
                     ctx->G1 + (size_t)t0 * G4, 4 * Rq, o));
        HIPC(hipEventRecord(ctx->evGc[cch], ctx->st3));
      }
    }
    // the recurrence's wait in front of wavefront step s (layer-1 cell of token s, 1-based): the chunk that
    // starts with that token
    auto wait_i2h_rows = [&](int s) -> int {
      if (t_head >= TL) return 0;
      for (int cch = 0; cch < kEncSideChunks; ++cch)
        if (ctx->enc_chunk_tok[cch] == s - 1 && s - 1 < TL) HIPC(hipStreamWaitEvent(st, ctx->evGc[cch], 0));
      return 0;
    };
    if (ws_path) {
      // both layers, all tokens: one launch, weights resident in registers (enc_ws.hip)
      EncWsParams q{};
      q.B = B; q.R = Rq; q.TL = TL; q.bf16 = ctx->bf16 == 1;
      q.G1 = ctx->G1; q.G2 = ctx->G2; q.h1 = ctx->h1; q.c1 = ctx->c1; q.tc1 = ctx->tc1; q.x2 = ctx->x2;
      q.h2 = ctx->h2; q.c2 = ctx->c2; q.tc2 = ctx->tc2;
      q.Wh1 = ctx->h2h[0].W; q.Wi2 = ctx->i2h[1].W; q.Wh2 = ctx->h2h[1].W;
      q.bi2 = ctx->i2h[1].b; q.bh2 = ctx->h2h[1].b;
      q.mask = m_rnn; q.mscale = sc(RAU_MASK_RNN);
      q.cnt = ctx->ws_cnt; q.err = ctx->perr_d;
      RUN("enc_ws", (double)TL * 3 * gflop(B, 4 * Rq, Rq), 0, enc_ws_forward(st, GATES_DEEP, q));
      HIPC(hipMemcpyAsync(ctx->perr_h, ctx->perr_d, sizeof(int), hipMemcpyDeviceToHost, st));
      ctx->persist_used = true;
    } else
    for (int s = 1; s <= TL + 1; ++s) {
      if (int rc = wait_i2h_rows(s)) return rc;   // G1 rows of token s
      const float* Ap[3];
      const float* Wp[3];
      int nb = 0, i0 = -1, i1 = -1, i2 = -1;
      if (s <= TL && s >= 2) { Ap[nb] = ctx->h1 + (size_t)(s - 1) * BRq; Wp[nb] = ctx->h2h[0].W; i0 = nb++; }
      if (s >= 2) { Ap[nb] = ctx->x2 + (size_t)(s - 2) * BRq; Wp[nb] = ctx->i2h[1].W; i1 = nb++; }
      if (s >= 3) { Ap[nb] = ctx->h2 + (size_t)(s - 2) * BRq; Wp[nb] = ctx->h2h[1].W; i2 = nb++; }
      int nsp = 0;
      if (nb > 0)
        RUN("enc_h2h_gemm", gflop(B, 4 * Rq, Rq) * nb, 0,
            gemm_nt_batched_deferred(st, nb, B, 4 * Rq, Rq, Ap, Rq, Wp, Rq, ctx->slab,
                                     ctx->slab_floats, &nsp));
      LstmFwdCells cells{};
      if (s <= TL) {  // layer-1 cell t = s
        LstmFwdCell& C1 = cells.c[cells.n++];
        C1.g4 = ctx->G1 + (size_t)(s - 1) * G4;
        C1.has_input = 1;
        C1.slab = i0 >= 0 ? ctx->slab + (size_t)i0 * nsp * G4 : nullptr;
        C1.nsplit = i0 >= 0 ? nsp : 0;
        C1.c_prev = ctx->c1 + (size_t)(s - 1) * BRq; C1.cp_rs = Rq;
        C1.c = ctx->c1 + (size_t)s * BRq; C1.c_rs = Rq;
        C1.h = ctx->h1 + (size_t)s * BRq; C1.h_rs = Rq;
        C1.tanhc = ctx->tc1 + (size_t)(s - 1) * BRq;
        C1.drop_out = ctx->x2 + (size_t)(s - 1) * BRq;   // layer-2 input, DeepLSTM.lua:39
        C1.mask = m_rnn; C1.mask_e0 = (size_t)(s - 1) * BRq; C1.mscale = sc(RAU_MASK_RNN);
      }
      if (s >= 2) {  // layer-2 cell t = s - 1
        const int t = s - 1;
        LstmFwdCell& C2 = cells.c[cells.n++];
        C2.g4 = ctx->G2 + (size_t)(t - 1) * G4;
        C2.has_input = 0;
        C2.b1 = ctx->i2h[1].b; C2.b2 = ctx->h2h[1].b;
        C2.slab = ctx->slab + (size_t)i1 * nsp * G4;      // x2 W_i2h2^T then h2 W_h2h2^T partials
        C2.nsplit = i2 >= 0 ? 2 * nsp : nsp;
        C2.c_prev = ctx->c2 + (size_t)(t - 1) * BRq; C2.cp_rs = Rq;
        C2.c = ctx->c2 + (size_t)t * BRq; C2.c_rs = Rq;
        C2.h = ctx->h2 + (size_t)t * BRq; C2.h_rs = Rq;
        C2.tanhc = ctx->tc2 + (size_t)(t - 1) * BRq;
        C2.drop_out = nullptr; C2.mask = nullptr; C2.mask_e0 = 0; C2.mscale = 1.f;
      }
      RUN("lstm_fwd", 0, BRq * 4.0 * 10 * cells.n, lstm_fwd_multi(st, GATES_DEEP, B, Rq, cells));
    }
  }
  return 0;
  };
  ctx->I_shared = (m_x == nullptr);
  if (ctx->I_shared) {   // evaluate mode: one launch group
    ctx->cur.assign(H, 0);
    ctx->cur[0] = H;
  } else {
    ctx->cur = ctx->groups;
  }
  const std::vector<int>& gsz = ctx->cur;
  {
    hipStream_t sb = ctx->st2;
    HIPC(hipEventRecord(ctx->evA, st));
    HIPC(hipStreamWaitEvent(sb, ctx->evA, 0));
    // feature-map dropout, SS:239: unless the caller supplied the mask, the pass that makes the H masked
    // copies draws the keep bits itself (they have no other reader on this path)
    const bool x16 = m_x && ctx->xd16;   // bf16 mode: the hop copies of the feature map are stored as bf16
    const bool x_gen = m_x && !ctx->mexplicit[RAU_MASK_X] && SL == S && ((size_t)B * D * S) % 16 == 0;
    if (!x_gen)
      if (int rc = gen_masks(ctx, -1, RAU_MASK_X, sb)) return rc;
    RUNS(sb, "transpose", 0, (double)M * D * 8, transpose2d(sb, M, D, ctx->i_embed.W, ctx->WiT, ctx->WiT16));
    RUNS(sb, "transpose", 0, (double)A * M * 8, transpose2d(sb, A, M, ctx->att_i.W, ctx->WpT, ctx->WpT16));
    if (x_gen)
      RUNS(sb, "dropout_features", 0, (double)(x16 ? H + 2 : 2 * H + 2) * B * D * S * 2,
           dropout_features_gen(sb, ctx->seed, RAU_MASK_X, ctx->step, ctx->mp[RAU_MASK_X], ctx->dkey, H,
                                (size_t)B * D * S, ctx->feats, sc(RAU_MASK_X),
                                x16 ? (void*)ctx->xd16 : (void*)ctx->xd, x16 ? 1 : 0));
    else if (x16)
      RUNS(sb, "dropout_features", 0, (double)(H + 2) * B * D * S * 2,
           dropout_features_b16(sb, H, (size_t)B * D * S, ctx->feats, m_x, sc(RAU_MASK_X), ctx->xd16));
    else if (m_x)
      RUNS(sb, "dropout_features", 0, (double)(H + 1) * B * D * S * 4,
           dropout_features(sb, H, (size_t)B * D * S, ctx->feats, m_x, sc(RAU_MASK_X), ctx->xd, 0, SL,
                            S));
    for (int h0 = 0; h0 < H; h0 += gsz[h0]) {
      const int nBI = ctx->I_shared ? B : gsz[h0] * B;
      const size_t hb = ctx->I_shared ? 0 : (size_t)h0 * B;  // first (hop, sample) row
      const float* xin = m_x ? ctx->xd + hb * D * S : ctx->feats;
      float* Ig = ctx->I + hb * M * S;
      float* Pg = ctx->I_shared ? ctx->P0 : ctx->T + hb * A * S;
      if (x16)
        RUNS(sb, "conv_embed_fwd", gflop(M, (double)nBI * S, D),
             (double)nBI * D * S * 2 + (double)nBI * M * S * 4,
             conv_embed_fwd_b16(sb, nBI, D, S, M, (const uint16_t*)ctx->xd16 + hb * D * S, ctx->WiT16,
                                ctx->i_embed.b, Ig));
      else
        RUNS(sb, "conv_embed_fwd", gflop(M, (double)nBI * S, D),
             ((double)nBI * D * S + (double)nBI * M * S) * 4,
             conv_embed_fwd(sb, nBI, D, S, M, xin, ctx->WiT, ctx->i_embed.b, Ig, ctx->bf16));
      if (x16)
        RUNS(sb, "conv_att_pre", gflop(A, (double)nBI * S, M),
             ((double)nBI * M * S + (double)nBI * A * S) * 4,
             conv_att_pre_b16(sb, nBI, M, S, A, Ig, ctx->WpT16, ctx->att_i.b, Pg));
      else
        RUNS(sb, "conv_att_pre", gflop(A, (double)nBI * S, M),
             ((double)nBI * M * S + (double)nBI * A * S) * 4,
             conv_att_pre(sb, nBI, M, S, A, Ig, ctx->WpT, ctx->att_i.b, Pg, ctx->bf16));
      HIPC(hipEventRecord(ctx->evF[h0], sb));
    }
  }

  if (int rc = encoder_forward()) return rc;
  RUN("gather_q", 0, (double)B * Q * 8,
      gather_q(st, B, Rq, TL, ctx->lens_d, ctx->c1, ctx->h1, ctx->c2, ctx->h2, ctx->q));

  // ---------------- RAU hops, SS:467-520
  const size_t BM_ = (size_t)B * M, BR_ = (size_t)B * R;
  RUN("apply_mask", 0, (double)H * B * Q * 8,
      apply_mask(st, (size_t)H * B * Q, (size_t)B * Q, ctx->q, m_q, sc(RAU_MASK_Q), ctx->qd));
  bool q_split = false;
  {  // q_embed's question half (SS:233) for every hop clone; without dropout on q (evaluate mode) the
     // clones see the same rows: computed once
    ctx->yq_shared = (m_q == nullptr);
    // with dropout: hop 0's rows here, the other hops' rows on the weight-gradient stream (hop 1 waits)
    q_split = !ctx->yq_shared && side_split(ctx) && H > 1;
    const int qrows = (ctx->yq_shared || q_split) ? B : H * B;
    LINOPTS(o);
    o.bias = ctx->q_proj.b;
    o.bias2 = ctx->h_proj.b;
    RUN("q_proj_gemm", gflop(qrows, M, Q), 0,
        gemm_nt(st, qrows, M, Q, ctx->qd, Q, ctx->q_proj.W, Q, ctx->Yq, M, o));
  }
  if (q_split) {
    HIPC(hipEventRecord(ctx->evQ0, st));            // qd is complete
    HIPC(hipStreamWaitEvent(ctx->st3, ctx->evQ0, 0));
    LinOpts o;
    o.slab = ctx->slab3; o.slab_floats = ctx->slab3_floats;
    o.bias = ctx->q_proj.b;
    o.bias2 = ctx->h_proj.b;
    RUNS(ctx->st3, "q_proj_gemm", gflop((H - 1) * B, M, Q), 0,
         gemm_nt(ctx->st3, (H - 1) * B, M, Q, ctx->qd + (size_t)B * Q, Q, ctx->q_proj.W, Q,
                 ctx->Yq + (size_t)B * M, M, o));
    HIPC(hipEventRecord(ctx->evQ, ctx->st3));
  }
  // i_embed SS:238-242 does not depend on the recurrence: all hops in one launch.
  // Train mode: each hop clone has its own dropout mask on the feature map
  // (SS:239, SS:343-347) -> xd[h] = X (.) mask_h, then one GEMM over H*B samples.
  // Evaluate mode: dropout is the identity, so I is hop-invariant and computed once.
  HIPC(hipMemsetAsync(ctx->cc, 0, BR_ * sizeof(float), st));  // att_c, att_h zeros SS:362-365
  HIPC(hipMemsetAsync(ctx->hh, 0, BR_ * sizeof(float), st));
  const int32_t* labels = ctx->have_labels ? ctx->labels_d : nullptr;
  const size_t reg3 = ctx->slab3_floats / 4;
  const int head_max = (int)std::max<size_t>(1, reg3 / ((size_t)B * c.K));  // rows the logits slab holds
  int gstart = 0;
  for (int h = 0; h < H; ++h) {
    if (gsz[h]) {
      HIPC(hipStreamWaitEvent(st, ctx->evF[h], 0));  // this group's I and P are ready
      gstart = h;
    }
    if (h == 1 && q_split) HIPC(hipStreamWaitEvent(st, ctx->evQ, 0));   // Yq rows of hops 1..
    if (int rc = hop_forward_chain(ctx, h, ctx->cc + (size_t)h * BR_, ctx->hh + (size_t)h * BR_,
                                   ctx->cc + (size_t)(h + 1) * BR_, ctx->hh + (size_t)(h + 1) * BR_,
                                   ctx->I + (ctx->I_shared ? 0 : (size_t)h * BM_ * S),
                                   ctx->I_shared ? ctx->P0 : ctx->T + (size_t)h * B * A * S))
      return rc;
    // classifier + criterion heads of the finished hops: nothing on the recurrence waits for
    // them, so they run on the weight-gradient stream (idle in the forward pass), batched over
    // the hops of a launch group
    const bool group_end = h + 1 == H || gsz[h + 1] != 0;
    if (group_end || h + 1 - gstart >= head_max) {
      HIPC(hipEventRecord(ctx->evH[h], st));
      HIPC(hipStreamWaitEvent(ctx->st3, ctx->evH[h], 0));
      for (int h0 = gstart; h0 <= h; h0 += head_max)
        if (int rc = hop_forward_head(ctx, ctx->st3, ctx->slab3, reg3, h0,
                                      std::min(head_max, h + 1 - h0), labels))
          return rc;
      if (head_dgrad_fwd(ctx)) {
        // The backward's first two products do not depend on the recurrence either: dpre = (dl Wc) (.) mask
        // and dhn = dpre Wo of these hops follow their criterion head on the same (idle) stream, so the
        // forward / backward seam -- where the bulk stream and the matrix pipes wait for the chain -- is two
        // GEMMs shorter.  dl is not yet scaled by the hop weights (they arrive with rau_backward); both
        // products are linear in it, so rau_backward scales dl, dpre and dhn together.
        const int nh = h + 1 - gstart;
        const bool trm = ctx->mode == RAU_MODE_TRAIN;
        LinOpts o;
        o.slab = ctx->slab3; o.slab_floats = ctx->slab3_floats;
        o.emask = (trm && mask_p(ctx, RAU_MASK_MF) > 0.f) ? ctx->mbits[RAU_MASK_MF] : nullptr;
        o.emask_e0 = (size_t)gstart * BM_;
        o.emscale = 1.f / (1.f - mask_p(ctx, RAU_MASK_MF));
        RUNS(ctx->st3, "head_dgrad", 2.0 * nh * B * M * c.K, 0,
             gemm_nn(ctx->st3, nh * B, M, c.K, ctx->dl + (size_t)gstart * B * c.K, c.K, ctx->cls.W, M,
                     ctx->dpre + (size_t)gstart * BM_, M, o));
        LinOpts o2;
        o2.slab = ctx->slab3; o2.slab_floats = ctx->slab3_floats;
        RUNS(ctx->st3, "head_dgrad", 2.0 * nh * B * R * M, 0,
             gemm_nn(ctx->st3, nh * B, R, M, ctx->dpre + (size_t)gstart * BM_, M, ctx->lstm_out.W, R,
                     ctx->dhn + (size_t)gstart * BR_, R, o2));
      }
      gstart = h + 1;
    }
  }
  HIPC(hipEventRecord(ctx->evHd, ctx->st3));
  HIPC(hipStreamWaitEvent(st, ctx->evHd, 0));   // callers order against st only
  if (ctx->have_labels)
    RUN("loss_reduce", 0, 0, loss_reduce(st, H, B, ctx->lossrow, ctx->losses_d));
  ctx->fwd_done = true;
  ctx->dpre_fwd = head_dgrad_fwd(ctx);
  return RAU_OK;
}

// The caller's hop_w may be a temporary (and may be pinned memory, for which an async copy really
// is asynchronous): stage it in the ctx's own pinned buffer first.  Two slots, alternated, so the
// copy of step n is never overwritten by the host while step n+1's call prepares its own.
static int upload_hop_weights(rau_ctx* ctx, const float* hop_w) {
  const int H = ctx->cfg.H;
  const int sl = (ctx->hopw_slot ^= 1);
  float* stage = ctx->hopw_h + (size_t)sl * H;
  // a host that runs two or more steps ahead of the device must not overwrite a staging slot whose
  // copy has not been read yet: wait for the copy issued from this slot two calls ago
  if (!ctx->hopw_ev[sl]) HIPC(hipEventCreateWithFlags(&ctx->hopw_ev[sl], hipEventDisableTiming));
  else HIPC(hipEventSynchronize(ctx->hopw_ev[sl]));
  std::memcpy(stage, hop_w, (size_t)H * sizeof(float));
  HIPC(hipMemcpyAsync(ctx->hopw_d, stage, (size_t)H * sizeof(float), hipMemcpyHostToDevice, ctx->st));
  if (!ctx->capturing) HIPC(hipEventRecord(ctx->hopw_ev[sl], ctx->st));
  return 0;
}

// =============================================================== backward
int rau_backward(rau_ctx* ctx, const float* hop_w) {
  NEED(ctx && hop_w, "null argument");
  if (!ctx->fwd_done) return fail(RAU_ERR_STATE, "rau_backward: call rau_forward first");
  set_skinny_policy(ctx);
  if (!ctx->have_labels) return fail(RAU_ERR_STATE, "rau_backward: batch has no labels");
  const rau_config& c = ctx->cfg;
  const int B = c.B, E = c.E, Rq = c.Rq, D = c.D, S = ctx->Sp, SL = c.S, M = c.M, A = c.A, R = c.R,
            K = c.K, H = c.H, Q = ctx->Q;
  const int TL = ctx->max_len;
  // Backward launch groups of the bulk stream.  Forward groups want to be large (the flattened-
  // column kernels lose a ragged last round per launch); the backward kernels do not (per-sample
  // dgrad tiles and the split-K weight gradients fill whole rounds at any hop count), and a
  // group's gradients can only start once its LAST hop's chain is done, so the backward partition
  // is its own: RAU_BWD_GROUPS="1,1,2,2,2" (sizes in HOP order, must sum to H), default = the
  // forward partition.  Evaluate mode (I shared): one group.
  std::vector<int> bsz_store;
  if (!ctx->I_shared && !ctx->bgroups.empty()) bsz_store = ctx->bgroups;
  const std::vector<int>& gsz = bsz_store.empty() ? ctx->cur : bsz_store;
  hipStream_t st = ctx->st;
  const bool tr = ctx->mode == RAU_MODE_TRAIN;
  auto mk = [&](int site) -> const uint32_t* {
    return (tr && mask_p(ctx, site) > 0.f) ? ctx->mbits[site] : nullptr;
  };
  auto sc = [&](int site) { return 1.f / (1.f - mask_p(ctx, site)); };
  const uint32_t *m_we = mk(RAU_MASK_WE), *m_rnn = mk(RAU_MASK_RNN), *m_q = mk(RAU_MASK_Q);
  auto gflop = [](double m, double n, double k) { return 2.0 * m * n * k; };
  const size_t BM_ = (size_t)B * M, BS_ = (size_t)B * S, BR_ = (size_t)B * R;
  const size_t BRq = (size_t)B * Rq;
  ctx->fwd_done = false;  // dl is scaled in place below: one backward per forward

  // dpred:mul(w[h])  SS:569 / MS:568-570 / Full:587-589
  if (!ctx->capturing) {  // (rau_graph_step uploads the weights before it launches the graph)
    if (int rc = upload_hop_weights(ctx, hop_w)) return rc;
  }
  if (ctx->dpre_fwd)   // the forward formed dpre / dhn from the unscaled dl: scale all three
    RUN("scale_hops", 0, (double)H * B * (K + M + R) * 8,
        scale_hops3(st, H, ctx->hopw_d, (size_t)B * K, ctx->dl, (size_t)B * M, ctx->dpre, (size_t)B * R, ctx->dhn));
  else
    RUN("scale_hops", 0, (double)H * B * K * 8, scale_hops(st, H, (size_t)B * K, ctx->hopw_d, ctx->dl));

  // Hops behind the last one with a non-zero loss weight receive no gradient at all (zero
  // criterion gradient, zero recurrent gradient: Full/ResNet late-epoch gating, Full:587-589):
  // their backward is identically zero and is skipped -- HA hops are "active".
  int HA = 0;
  for (int h = 0; h < H; ++h)
    if (hop_w[h] != 0.f) HA = h + 1;

  // ---------------- RAU BPTT, SS:561-578
  // Off the recurrence: dpre = (dl Wc) (.) mask and dhn = dpre Wo for all active hops at once
  const uint32_t* m_mf = mk(RAU_MASK_MF);
  if (HA > 0 && !ctx->dpre_fwd) {
    LINOPTS(o);
    o.emask = m_mf;
    o.emask_e0 = 0;
    o.emscale = sc(RAU_MASK_MF);
    RUN("head_dgrad", gflop(HA * B, M, K), 0,
        gemm_nn(st, HA * B, M, K, ctx->dl, K, ctx->cls.W, M, ctx->dpre, M, o));
    LINOPTS(o2);
    RUN("head_dgrad", gflop(HA * B, R, M), 0,
        gemm_nn(st, HA * B, R, M, ctx->dpre, M, ctx->lstm_out.W, R, ctx->dhn, R, o2));
  }
  const float* dc_next = nullptr;  // grad_att_c / grad_att_h zeros, SS:561-562
  float* dh_part = nullptr;        // gradient at next_h: K-split partials left by the previous hop
  int dh_part_ns = 0;
  // bf16 mode, train-mode step on 14x14 maps: dS goes out as bf16 (its consumers below read that)
  ctx->ds16_step = ctx->dS16 && !ctx->I_shared && !ctx->att_split_env && conv_dz_fused_ok(S, M, ctx->bf16);
  struct Ds16Reset { rau_ctx* c; ~Ds16Reset() { c->ds16_step = false; } } ds16_reset{ctx};
  bool dq_side = false;
  int dq_rows_left = HA;           // hops [0, dq_rows_left) whose dq term this stream still owes
  for (int h = HA - 1; h >= 0; --h) {
    float* dc_out = ctx->dcn[h & 1];
    {
      HopGrad g{};
      g.dl = ctx->dl + (size_t)h * B * K;
      g.dc_next = dc_next;
      g.dh_next = nullptr;
      g.dc_out = dc_out;
      g.dh_out = nullptr;
      g.dpre_ready = 1;
      g.dhn_all = ctx->dhn;
      g.dh_part = dh_part;
      g.dh_part_ns = dh_part_ns;
      g.dh_part_out = &dh_part;
      g.dh_part_ns_out = &dh_part_ns;
      // h is the first hop of its launch group: its conv gradients may start behind att_bwd, three launches
      // before the hop's backward is over (it matters for the first group: the forward / backward seam)
      g.ev_conv_ready = (gsz[h] && (seam_mask() & 2)) ? ctx->evK[h] : nullptr;
      g.dh_prev_dead = h == 0;   // hop 0's prev_h is the constant initial state: nothing reads its gradient
      if (int rc = hop_backward(ctx, h, ctx->cc + (size_t)h * BR_,
                                ctx->I + (ctx->I_shared ? 0 : (size_t)h * BM_ * S), g))
        return rc;
    }
    dc_next = dc_out;
    // dq's per-hop terms dq~_h Wq (ConcatTable backward, SS:579) do not feed the recurrence: the
    // rows of a finished hop group go to the weight-gradient stream (idle until the hop loop
    // ends); only the last group's stay on this stream, in front of dq_reduce below
    if (gsz[h] && h > 0 && side_split(ctx)) {
      const int nh = std::min(h + gsz[h], HA) - h;
      HIPC(hipEventRecord(ctx->evK[h], st));
      HIPC(hipStreamWaitEvent(ctx->st3, ctx->evK[h], 0));
      LinOpts o;
      o.slab = ctx->slab3; o.slab_floats = ctx->slab3_floats;
      RUNS(ctx->st3, "q_proj_dgrad", gflop(nh * B, Q, M), 0,
           gemm_nn(ctx->st3, nh * B, Q, M, ctx->dqt + (size_t)h * BM_, M, ctx->q_proj.W, Q,
                   ctx->dQD + (size_t)h * B * Q, Q, o));
      HIPC(hipEventRecord(ctx->evDq, ctx->st3));
      dq_side = true;
      dq_rows_left = h;       // hops [0, h) are still to do
    }
    // ---------------- bulk stream: the 1x1-conv gradients are off the recurrence's
    // critical path (dZ only feeds weight gradients; the feature-map gradient is dead,
    // SS:579, never formed).  As soon as a hop group's chain is done its conv gradients
    // start on the bulk stream, overlapping the remaining hops and the encoder BPTT:
    // dZ = (Wp^T dS + dj (x) a)(1 - I^2); dWp += dS I^T; dWi += dZ X'^T.
    if (gsz[h]) {   // h is the first hop of its group: the whole group's attention backward is done
      hipStream_t sb = ctx->st2;
      if (!(seam_mask() & 2)) HIPC(hipEventRecord(ctx->evK[h], st));
      HIPC(hipStreamWaitEvent(sb, ctx->evK[h], 0));   // recorded inside hop_backward (ev_conv_ready)
      if (!ctx->I_shared) {
        const int nH = (std::min(h + gsz[h], HA) - h) * B;   // active hops of this group
        const size_t hb = (size_t)h * B;
        // Where the per-sample tiling applies the dgrad's epilogue also applies (1 - I^2) and
        // hands back per-sample row sums (the i_embed bias gradient): the i_embed weight gradient
        // then stages a plain operand (one global load and no VALU work per element less).
        const int dzf = conv_dz_fused_ok(S, M, ctx->bf16);
        if (dzf)
          RUNS(sb, "conv_att_dgrad", gflop(M, (double)nH * S, A),
               ((double)nH * A * S + 3.0 * nH * M * S) * 4,
               conv_att_dgrad_dz(sb, nH, M, S, A,
                                 ctx->ds16_step ? (const float*)((const uint16_t*)ctx->dS16 + hb * A * S)
                                                : ctx->T + hb * A * S,
                                 ctx->att_i.W, ctx->dj + hb * M,
                                 ctx->a + hb * S, ctx->I + hb * M * S,
                                 ctx->xd16 ? (float*)((uint16_t*)ctx->dZ + hb * M * S) : ctx->dZ + hb * M * S,
                                 ctx->dbi_part + hb * M, ctx->xd16 ? 1 : 0, ctx->bf16, ctx->ds16_step ? 1 : 0,
                                 chain_bound(ctx) ? 1 : 0));
        else
          RUNS(sb, "conv_att_dgrad", gflop(M, (double)nH * S, A),
               ((double)nH * A * S + 2.0 * nH * M * S) * 4,
               conv_att_dgrad(sb, nH, M, S, A, ctx->T + hb * A * S, ctx->att_i.W, ctx->dj + hb * M,
                              ctx->a + hb * S, ctx->dZ + hb * M * S, ctx->bf16));
        {   // RAU_BULK2: the att_i weight gradient (independent of dZ) on the second bulk stream
          if (ctx->ds16_step)
            RUNS(sb, "conv_att_wgrad", gflop(A, M, (double)nH * S),
                 (double)nH * A * S * 2 + (double)nH * M * S * 4,
                 conv_att_wgrad_ds16(sb, nH, M, S, A, (const uint16_t*)ctx->dS16 + hb * A * S,
                                     ctx->I + hb * M * S, ctx->att_i.dW, ctx->slab2));
          else
          RUNS(sb, "conv_att_wgrad", gflop(A, M, (double)nH * S),
               ((double)nH * A * S + (double)nH * M * S) * 4,
               conv_att_wgrad(sb, nH, M, S, A, ctx->T + hb * A * S, ctx->I + hb * M * S,
                              ctx->att_i.dW, ctx->slab2, ctx->bf16));
        }
        if (ctx->xd16 && dzf)
          RUNS(sb, "conv_embed_wgrad", gflop(M, D, (double)nH * S),
               (double)nH * M * S * 2 + (double)nH * D * S * 2,
               conv_embed_wgrad_b16(sb, nH, D, S, M, (const uint16_t*)ctx->dZ + hb * M * S,
                                    (const uint16_t*)ctx->xd16 + hb * D * S, ctx->i_embed.dW, ctx->slab2));
        else
          RUNS(sb, "conv_embed_wgrad", gflop(M, D, (double)nH * S),
               ((double)nH * M * S * (dzf ? 1 : 2) + (double)nH * D * S) * 4,
               conv_embed_wgrad(sb, nH, D, S, M, ctx->dZ + hb * M * S, ctx->I + hb * M * S,
                                  ctx->xd + hb * D * S, ctx->i_embed.dW, ctx->slab2, ctx->bf16,
                                  ctx->i_embed.db, dzf));
        if (dzf && h == 0)   // last group: i_embed bias gradient = column sums of the per-sample rows
          RUNS(sb, "colsum", 0, (double)HA * B * M * 4,
               colsum_acc(sb, HA * B, M, ctx->dbi_part, M, ctx->i_embed.db, ctx->coltmp2));
      } else {
        for (int hh2 = 0; hh2 < HA; ++hh2) {  // evaluate mode: I (and X) shared by all hops
          float* Th2 = ctx->T + (size_t)hh2 * B * A * S;
          float* dZh = ctx->dZ + (size_t)hh2 * BM_ * S;
          RUNS(sb, "conv_att_dgrad", gflop(M, (double)B * S, A), ((double)B * A * S + 2.0 * BM_ * S) * 4,
               conv_att_dgrad(sb, B, M, S, A, Th2, ctx->att_i.W, ctx->dj + (size_t)hh2 * BM_,
                              ctx->a + (size_t)hh2 * BS_, dZh, ctx->bf16));
          RUNS(sb, "conv_att_wgrad", gflop(A, M, (double)B * S), ((double)B * A * S + BM_ * S) * 4,
               conv_att_wgrad(sb, B, M, S, A, Th2, ctx->I, ctx->att_i.dW, ctx->slab2, ctx->bf16));
          RUNS(sb, "conv_embed_wgrad", gflop(M, D, (double)B * S), (BM_ * S + (double)B * D * S) * 4,
               conv_embed_wgrad(sb, B, D, S, M, dZh, ctx->I, ctx->feats, ctx->i_embed.dW, ctx->slab2,
                                ctx->bf16, ctx->i_embed.db));
        }
      }
    }
  }
  {  // dq = sum_h (dq~_h Wq) (.) mask_h     (ConcatTable backward, SS:579)
    LINOPTS(o);
    if (dq_rows_left > 0)
      RUN("q_proj_dgrad", gflop(dq_rows_left * B, Q, M), 0,
          gemm_nn(st, dq_rows_left * B, Q, M, ctx->dqt, M, ctx->q_proj.W, Q, ctx->dQD, Q, o));
    if (dq_side) HIPC(hipStreamWaitEvent(st, ctx->evDq, 0));
    RUN("dq_reduce", 0, (double)HA * B * Q * 4,
        dq_reduce(st, HA, (size_t)B * Q, ctx->dQD, m_q, sc(RAU_MASK_Q), ctx->dq));
  }
  // bulk stream tail: the join event (the i_embed bias gradient comes out of conv_embed_wgrad:
  // row sums of its staged dZ operand)
  {
    hipStream_t sb = ctx->st2;
    HIPC(hipEventRecord(ctx->evD, sb));
  }
  // ---------------- mult-group weight gradients, one GEMM per weight over all hops.
  // Throughput work nothing else waits for: third stream, so the encoder BPTT (the
  // long latency-bound chain) starts right after dq instead of behind ~40 launches.
  HIPC(hipEventRecord(ctx->evW, st));
  // Where the grouped Linear weight-gradient GEMMs run.  Two MFMA-bound kernels side by side share one
  // matrix pipe: the round-3 timeline shows the mult group's GEMM (943 us beside the bulk stream, ~500
  // alone) and the conv weight gradient it overlaps (827 us instead of ~450) each at about half their
  // stand-alone rate, i.e. nothing is gained by overlapping them, and the recurrence's kernels then
  // compete with two heavy kernels instead of one.  Where the bulk stream is the longer path (f32 step,
  // more than 64 samples) the mult group's GEMM therefore goes to the END of the bulk stream, behind the
  // last conv gradient (round 4: 9.43 -> 9.38 ms with dgrad_dma.hip; conv_att_dgrad 0.47 -> 0.54 and
  // conv_att_wgrad 0.57 -> 0.60 of peak in the step).  The encoder group's GEMM (after the BPTT) measured
  // the same on either stream and stays on the third.  RAU_WG_BULK=<mask> overrides (A/B, DESIGN.md section
  // 9): 1 = mult group on the bulk stream, 2 = encoder group too.  Not with the side-stream split (the
  // third stream's split-K workspace would be shared).
  static const int wg_env = [] { const char* e = std::getenv("RAU_WG_BULK"); return e ? std::atoi(e) : -1; }();
  const int wg_bulk = side_split(ctx) ? 0 : wg_env >= 0 ? wg_env : (chain_bound(ctx) ? 0 : 1);
  auto mult_wgrads = [&]() -> int {
    hipStream_t sw = (wg_bulk & 1) ? ctx->st2 : ctx->st3;
    HIPC(hipStreamWaitEvent(sw, ctx->evW, 0));
    const int rows = HA * B;                 // active hops only
    if (rows == 0) {
      HIPC(hipEventRecord(ctx->evM3, sw));
      return 0;
    }
    {  // dW += dY^T X and db += column sums of dY for the nine Linears: one grouped launch
      TnProblem pr[13];
      const int np = mult_wgrad_problems(ctx, pr);
      double fl = 0;
      for (int i = 0; i < np; ++i) fl += gflop(pr[i].M, pr[i].N, rows);
      // the workspace of the stream it runs on
      float* ws = (wg_bulk & 1) ? ctx->slab2 : ctx->slab3;
      const size_t ws_floats = (wg_bulk & 1) ? ctx->slab2_floats : ctx->slab3_floats;
      RUNS(sw, "wgrad_gemm", fl, 0, gemm_tn_group_acc(sw, pr, np, rows, ws, ws_floats, ctx->bf16 == 1));
    }
    float* ct = (wg_bulk & 1) ? ctx->coltmp2 : ctx->coltmp3;   // the column-sum scratch of the stream it runs on
    // att_score: dws = sum dz T ; dbs = sum dz.  att_i bias: sum dS.  i_embed bias: sum dZ.
    RUNS(sw, "colsum", 0, (double)rows * A * 4, colsum_acc(sw, rows, A, ctx->dwsp, A, ctx->att_score.dW, ct));
    HIPC(hipMemsetAsync(ctx->tmpS, 0, S * sizeof(float), sw));
    RUNS(sw, "colsum", 0, (double)rows * S * 4, colsum_acc(sw, rows, SL, ctx->dz, S, ctx->tmpS, ct));
    RUNS(sw, "colsum", 0, S * 4.0, colsum_acc(sw, SL, 1, ctx->tmpS, 1, ctx->att_score.db, ct));
    RUNS(sw, "colsum", 0, (double)rows * A * 4, colsum_acc(sw, rows, A, ctx->du, A, ctx->att_i.db, ct));
    HIPC(hipEventRecord(ctx->evM3, sw));   // with evD: the mult group's gradients are final
    return 0;
  };
  if (int rc = mult_wgrads()) return rc;

  // ---------------- encoder BPTT, SS:581-596 -- the same two-layer wavefront, reversed:
  // step u handles layer-2 cell u and layer-1 cell u+1; their incoming dh are the
  // K-split partials of the previous step's dG (dG2 W_h2h2, dG2 W_i2h2 through the
  // inter-layer dropout, dG1 W_h2h1), one batched GEMM, summed inside the cell kernel.
  if (TL > 0) {
    const int rows = TL * B;
    const size_t G4 = (size_t)B * 4 * Rq;
    // Encoder weight gradients, also on the weight-gradient stream.  Optionally (RAU_ENC_CHUNK)
    // in two time chunks -- the gradients of tokens t > uc are final once wavefront step uc is
    // done -- but running the first chunk under the second half of the BPTT slows that chain by
    // more than the shorter tail saves (measured 11.05 vs 10.6 ms), so the default is one chunk
    // behind the BPTT.
    auto enc_wgrads = [&](int t_lo, int t_hi, hipEvent_t ev) -> int {   // tokens t_lo < t <= t_hi
      if (t_hi <= t_lo) return 0;
      hipStream_t sw = (wg_bulk & 2) ? ctx->st2 : ctx->st3;
      HIPC(hipEventRecord(ev, st));
      HIPC(hipStreamWaitEvent(sw, ev, 0));
      const size_t r0 = (size_t)t_lo * B;
      const int nr = (t_hi - t_lo) * B;
      TnProblem pr[13];
      const int np = enc_wgrad_problems(ctx, r0, pr);
      double fl = 0;
      for (int i = 0; i < np; ++i) fl += gflop(pr[i].M, pr[i].N, nr);
      float* ws = (wg_bulk & 2) ? ctx->slab2 : ctx->slab3;   // the workspace of the stream it runs on
      const size_t ws_floats = (wg_bulk & 2) ? ctx->slab2_floats : ctx->slab3_floats;
      RUNS(sw, "wgrad_gemm", fl, 0, gemm_tn_group_acc(sw, pr, np, nr, ws, ws_floats, ctx->bf16 == 1));
      return 0;
    };
    const int hi = TL;
    for (int u = TL; u >= 0; --u) {
      const float* Ap[3];
      const float* Wp[3];
      int nb = 0, iq2 = -1, iqx = -1, iq1 = -1;
      if (u >= 1 && u + 1 <= TL) { Ap[nb] = ctx->dG2 + (size_t)u * G4; Wp[nb] = ctx->h2h[1].W; iq2 = nb++; }
      if (u + 1 <= TL) { Ap[nb] = ctx->dG2 + (size_t)u * G4; Wp[nb] = ctx->i2h[1].W; iqx = nb++; }
      if (u + 2 <= TL) { Ap[nb] = ctx->dG1 + (size_t)(u + 1) * G4; Wp[nb] = ctx->h2h[0].W; iq1 = nb++; }
      int nsp = 0;
      if (nb > 0)
        RUN("enc_h2h_dgrad", gflop(B, Rq, 4 * Rq) * nb, 0,
            gemm_nn_batched_deferred(st, nb, B, Rq, 4 * Rq, Ap, 4 * Rq, Wp, Rq, ctx->slab,
                                     ctx->slab_floats, &nsp));
      LstmBwdCells cells{};
      cells.lens = ctx->lens_d;
      cells.dq_rs = Q;
      if (u >= 1) {  // layer-2 cell t = u
        LstmBwdCell& C2 = cells.c[cells.n++];
        C2.gates = ctx->G2 + (size_t)(u - 1) * G4;
        C2.c_prev = ctx->c2 + (size_t)(u - 1) * BRq; C2.cp_rs = Rq;
        C2.tanhc = ctx->tc2 + (size_t)(u - 1) * BRq;
        C2.slabA = iq2 >= 0 ? ctx->slab + (size_t)iq2 * nsp * BRq : nullptr;
        C2.nA = iq2 >= 0 ? nsp : 0;
        C2.slabB = nullptr; C2.nBp = 0; C2.maskB = nullptr; C2.maskB_e0 = 0; C2.mscaleB = 1.f;
        C2.dc_next = u < TL ? ctx->edc[1][(u + 1) & 1] : nullptr;
        C2.dsum = ctx->dG2 + (size_t)(u - 1) * G4;
        C2.dc_prev = ctx->edc[1][u & 1];
        C2.t = u; C2.dq_c = ctx->dq + 2 * Rq; C2.dq_h = ctx->dq + 3 * Rq;
      }
      if (u + 1 <= TL) {  // layer-1 cell t = u + 1
        const int t = u + 1;
        LstmBwdCell& C1 = cells.c[cells.n++];
        C1.gates = ctx->G1 + (size_t)u * G4;
        C1.c_prev = ctx->c1 + (size_t)u * BRq; C1.cp_rs = Rq;
        C1.tanhc = ctx->tc1 + (size_t)u * BRq;
        C1.slabA = iq1 >= 0 ? ctx->slab + (size_t)iq1 * nsp * BRq : nullptr;
        C1.nA = iq1 >= 0 ? nsp : 0;
        C1.slabB = ctx->slab + (size_t)iqx * nsp * BRq;   // dG2[t] W_i2h2, then the dropout mask
        C1.nBp = nsp;
        C1.maskB = m_rnn; C1.maskB_e0 = (size_t)u * BRq; C1.mscaleB = sc(RAU_MASK_RNN);
        C1.dc_next = t < TL ? ctx->edc[0][(t + 1) & 1] : nullptr;
        C1.dsum = ctx->dG1 + (size_t)u * G4;
        C1.dc_prev = ctx->edc[0][t & 1];
        C1.t = t; C1.dq_c = ctx->dq; C1.dq_h = ctx->dq + Rq;
      }
      RUN("lstm_bwd", 0, BRq * 4.0 * 12 * cells.n, lstm_bwd_multi(st, GATES_DEEP, B, Rq, cells));
    }
    {  // gradient w.r.t. the word embeddings' tanh output, all tokens at once
      LINOPTS(o);
      RUN("enc_i2h_dgrad", gflop(rows, E, 4 * Rq), 0,
          gemm_nn(st, rows, E, 4 * Rq, ctx->dG1, 4 * Rq, ctx->i2h[0].W, E, ctx->dwe, E, o));
    }
    RUN("embed_bwd", 0, (double)rows * E * 12,
        embed_bwd(st, ctx->capturing ? c.T * B : ctx->nuniq, E, ctx->utok, ctx->ustart, ctx->upos, ctx->dwe, ctx->we, m_we,
                  sc(RAU_MASK_WE), ctx->grp[RAU_GROUP_EMBED].g));
    if (int rc = enc_wgrads(0, hi, ctx->evE)) return rc;
  }
  HIPC(hipEventRecord(ctx->evW3, ctx->st3));
  HIPC(hipStreamWaitEvent(st, ctx->evW3, 0));
  if (wg_bulk) HIPC(hipEventRecord(ctx->evD, ctx->st2));   // the bulk stream got weight-gradient work behind its convs
  HIPC(hipStreamWaitEvent(st, ctx->evD, 0));  // join: every gradient is ordered on st
  HIPC(hipEventRecord(ctx->evEnd, st));
  ctx->bwd_done = true;
  ctx->graph_last = false;
  return RAU_OK;
}

// One training step's forward + backward (optionally with the gradient zeroing in front) as ONE
// hipGraph launch.  The three streams, their fork/join events and every kernel argument are
// captured once per step "shape" -- (mode, longest question, active hops, which mask sites are
// explicit, zeroing or not) -- and replayed; what changes from step to step lives in device
// memory: the batch (rau_set_batch), the Philox key (rau_set_dropout_seed) and the hop weights
// (uploaded here, in front of the launch).
int rau_graph_step(rau_ctx* ctx, const float* hop_w, int zero_grads_first) {
  NEED(ctx && hop_w, "null argument");
  if (!ctx->have_batch || !ctx->have_labels)
    return fail(RAU_ERR_STATE, "rau_graph_step: needs a batch with labels (rau_set_batch)");
  if (ctx->prof_on) return fail(RAU_ERR_STATE, "rau_graph_step: profiling must be off");
  const int H = ctx->cfg.H;
  int HA = 0;
  for (int h = 0; h < H; ++h)
    if (hop_w[h] != 0.f) HA = h + 1;
  uint64_t key = (uint64_t)ctx->mode | ((uint64_t)ctx->max_len << 2) | ((uint64_t)HA << 12) |
                 ((uint64_t)(zero_grads_first != 0) << 22);
  for (int i = 0; i < 5; ++i) key |= (uint64_t)ctx->mexplicit[i] << (24 + i);
  key |= (uint64_t)ctx->cur_slot << 30;   // the captured kernels hold the batch slot's device pointers
  if (int rc = upload_hop_weights(ctx, hop_w)) return rc;
  hipGraphExec_t exec = nullptr;
  for (auto& g : ctx->graphs)
    if (g.first == key) exec = g.second;
  if (!exec) {
    hipGraph_t graph = nullptr;
    HIPC(hipStreamBeginCapture(ctx->st, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    int rc = zero_grads_first ? rau_zero_grads(ctx) : 0;
    if (!rc) rc = rau_forward(ctx);
    if (!rc) rc = rau_backward(ctx, hop_w);
    ctx->capturing = false;
    hipError_t e = hipStreamEndCapture(ctx->st, &graph);
    if (rc) {
      if (graph) hipGraphDestroy(graph);
      return rc;
    }
    if (e != hipSuccess)
      return fail(RAU_ERR_DEVICE, "hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess)
      return fail(RAU_ERR_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(e));
    ctx->graphs.push_back({key, exec});
  }
  HIPC(hipGraphLaunch(exec, ctx->st));
  HIPC(hipEventRecord(ctx->evEnd, ctx->st));   // a real (non-captured) end-of-step event
  ctx->fwd_done = false;
  ctx->bwd_done = true;
  ctx->graph_last = true;
  return RAU_OK;
}

int rau_wait_grads(rau_ctx* ctx, int group, void* hip_stream) {
  if (int rc = check_group(ctx, group)) return rc;
  if (!ctx->bwd_done) return fail(RAU_ERR_STATE, "rau_wait_grads: no rau_backward to wait for");
  hipStream_t s = (hipStream_t)hip_stream;
  if (group == RAU_GROUP_MULT && !ctx->graph_last) {
    // final once the bulk stream's conv gradients (evD) and the weight-gradient stream's
    // mult block (evM3) are done -- i.e. before the encoder BPTT, which they overlap
    HIPC(hipStreamWaitEvent(s, ctx->evD, 0));
    HIPC(hipStreamWaitEvent(s, ctx->evM3, 0));
  } else {
    HIPC(hipStreamWaitEvent(s, ctx->evEnd, 0));
  }
  return RAU_OK;
}

// ================================================================ results
// The persistent encoder's error word, read after a host synchronisation of the ctx stream: a bounded
// wait on a sibling workgroup's progress counter gave up (its siblings were not all resident: another
// process or context held the CUs).  THAT step's results are invalid and the call says so -- once:
// the word is cleared and the ctx falls back to the launch-per-step encoder for the rest of its life,
// so repeating the step succeeds instead of failing forever.
static int persist_check(rau_ctx* ctx) {
  if (!(ctx->persist_used && ctx->perr_h && *ctx->perr_h)) return RAU_OK;
  *ctx->perr_h = 0;
  hipMemsetAsync(ctx->perr_d, 0, sizeof(int), ctx->st);
  hipStreamSynchronize(ctx->st);
  ctx->enc_ws = ctx->enc_ws_train = false;
  ctx->persist_used = false;
  ctx->fwd_done = false;
  return fail(RAU_ERR_DEVICE, "persistent encoder: a bounded wait on another workgroup's progress counter gave up; "
                              "the results of that step are invalid -- repeat it: this context now uses the "
                              "launch-per-step encoder");
}
int rau_sync(rau_ctx* ctx) {
  NEED(ctx, "null ctx");
  HIPC(hipStreamSynchronize(ctx->st));
  return persist_check(ctx);
}
static int d2h(rau_ctx* ctx, void* host, const void* dev, size_t bytes) {
  NEED(ctx && host, "null argument");
  HIPC(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->st));
  HIPC(hipStreamSynchronize(ctx->st));
  return persist_check(ctx);   // same check as rau_sync: never hand back such results as OK
}
int rau_get_losses(rau_ctx* ctx, float* losses) {
  NEED(ctx, "null ctx");
  return d2h(ctx, losses, ctx->losses_d, ctx->cfg.H * sizeof(float));
}
int rau_get_argmax(rau_ctx* ctx, int32_t* ans) {
  NEED(ctx, "null ctx");
  return d2h(ctx, ans, ctx->argmax_d, (size_t)ctx->cfg.H * ctx->cfg.B * 4);
}
int rau_get_logits(rau_ctx* ctx, float* logits) {
  NEED(ctx, "null ctx");
  return d2h(ctx, logits, ctx->logits, (size_t)ctx->cfg.H * ctx->cfg.B * ctx->cfg.K * 4);
}
int rau_get_dopred(rau_ctx* ctx, float* dopred) {
  NEED(ctx, "null ctx");
  return d2h(ctx, dopred, ctx->dopred, (size_t)ctx->cfg.H * ctx->cfg.B * 4);
}
int rau_get_attention(rau_ctx* ctx, float* att) {
  NEED(ctx, "null ctx");
  NEED(att, "null argument");
  HIPC(hipMemcpy2DAsync(att, (size_t)ctx->cfg.S * 4, ctx->a, (size_t)ctx->Sp * 4,
                        (size_t)ctx->cfg.S * 4, (size_t)ctx->cfg.H * ctx->cfg.B,
                        hipMemcpyDeviceToHost, ctx->st));
  HIPC(hipStreamSynchronize(ctx->st));
  return RAU_OK;
}
int rau_get_question_state(rau_ctx* ctx, float* q) {
  NEED(ctx, "null ctx");
  return d2h(ctx, q, ctx->q, (size_t)ctx->cfg.B * ctx->Q * 4);
}
int rau_get_att_state(rau_ctx* ctx, float* c, float* h) {
  NEED(ctx, "null ctx");
  const size_t BR_ = (size_t)ctx->cfg.B * ctx->cfg.R, n = (size_t)ctx->cfg.H * BR_;
  if (c)
    if (int rc = d2h(ctx, c, ctx->cc + BR_, n * 4)) return rc;
  if (h)
    if (int rc = d2h(ctx, h, ctx->hh + BR_, n * 4)) return rc;
  return RAU_OK;
}

// ================================================================= update
int rau_noise_clip_adam(rau_ctx* ctx, int64_t step_t, float lr, float mult_lr, float beta1,
                        float beta2, float eps, float eta, float gamma, float clip,
                        uint64_t noise_seed, float* out_norms) {
  NEED(ctx, "null ctx");
  NEED(step_t >= 0 && gamma > 0.f && eta >= 0.f && clip > 0.f, "bad update hyper-parameters");
  hipStream_t st = ctx->st;
  // SS:598-599: var = eta / ((step_t+1) * gamma)
  const float nstd = eta > 0.f ? std::sqrt(eta / ((float)(step_t + 1) * gamma)) : 0.f;
  for (int gi = 0; gi < 3; ++gi) {
    Group& g = ctx->grp[gi];
    if (!g.m) {
      if (int rc = dalloc(ctx, &g.m, g.n)) return rc;
      if (int rc = dalloc(ctx, &g.v, g.n)) return rc;
    }
    g.adam_t += 1;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)g.adam_t);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)g.adam_t);
    const float group_lr = gi == RAU_GROUP_MULT ? mult_lr : lr;  // SS:770-772
    const float stepsize = (float)(group_lr * std::sqrt(bc2) / bc1);
    RUN("noise_sqnorm", 0, g.n * 8.0,
        add_noise_sqnorm(st, g.n, g.g, nstd, noise_seed + (uint64_t)step_t * 3 + gi, (uint32_t)gi,
                         ctx->npart));
    RUN("finish_norm", 0, 0, finish_norm(st, 1024, ctx->npart, ctx->norms_d + gi));
    RUN("clip_adam", 0, g.n * 28.0,
        clip_adam(st, g.n, g.w, g.g, g.m, g.v, ctx->norms_d + gi, clip, stepsize, beta1, beta2, eps));
  }
  if (out_norms) return d2h(ctx, out_norms, ctx->norms_d, 3 * sizeof(float));
  return RAU_OK;
}

// ================================================================= timing
int rau_stream(rau_ctx* ctx, void** hip_stream) {
  NEED(ctx && hip_stream, "null argument");
  *hip_stream = (void*)ctx->st;
  return RAU_OK;
}
int rau_timer_begin(rau_ctx* ctx) {
  NEED(ctx, "null ctx");
  HIPC(hipEventRecord(ctx->ev0, ctx->st));
  return RAU_OK;
}
int rau_timer_end(rau_ctx* ctx, float* ms) {
  NEED(ctx && ms, "null argument");
  HIPC(hipEventRecord(ctx->ev1, ctx->st));
  HIPC(hipEventSynchronize(ctx->ev1));
  HIPC(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return RAU_OK;
}
int rau_prof_enable(rau_ctx* ctx, int on) {
  NEED(ctx, "null ctx");
  if (!on)
    if (int rc = prof_collect(ctx)) return rc;
  ctx->prof_on = on != 0;
  ctx->prof_sparse = on == 2;
  return RAU_OK;
}
int rau_prof_reset(rau_ctx* ctx) {
  NEED(ctx, "null ctx");
  if (int rc = prof_collect(ctx)) return rc;
  ctx->pcls.clear();
  return RAU_OK;
}
int rau_prof_count(rau_ctx* ctx) {
  if (!ctx) return fail(RAU_ERR_INVALID, "null ctx");
  if (int rc = prof_collect(ctx)) return rc;
  return (int)ctx->pcls.size();
}
int rau_prof_entry(rau_ctx* ctx, int index, const char** name, int64_t* launches, double* total_ms,
                   double* flops, double* bytes) {
  NEED(ctx, "null ctx");
  NEED(index >= 0 && index < (int)ctx->pcls.size(), "prof index %d out of range", index);
  const ProfCls& p = ctx->pcls[index];
  if (name) *name = p.name.c_str();
  if (launches) *launches = p.launches;
  if (total_ms) *total_ms = p.ms;
  if (flops) *flops = p.flops;
  if (bytes) *bytes = p.bytes;
  return RAU_OK;
}

}  // extern "C"
