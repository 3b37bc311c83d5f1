// rau_ctx.h -- the rau_ctx object shared by the translation units that implement
// include/rau.h (internal to librau.so; not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/rau.h"
#include "kernels.h"

using namespace rau;

// LinOpts pre-wired with the ctx's split-K workspace
constexpr int kEncHeadTokens = 4;   // tokens whose layer-1 input projection stays on the chain stream
constexpr int kEncSideChunks = 4;   // the other tokens' projection: that many launches on the third stream, each with
                                    // its own event -- the recurrence waits for the chunk it is about to read, not
                                    // for all of them (one launch of 22 x 256 rows took 0.3-0.45 ms beside the bulk
                                    // tiles and held the encoder at token 5)
#define LINOPTS(name) LinOpts name; name.slab = ctx->slab; name.slab_floats = ctx->slab_floats

// ------------------------------------------------------------------ errors
// sets the calling thread's rau_last_error() message and returns `code`
__attribute__((visibility("hidden"))) int fail(int code, const char* fmt, ...);
#define HIPC(expr)                                                                        \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(RAU_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                  __FILE__, __LINE__);                                                    \
  } while (0)
#define NEED(cond, ...)                               \
  do {                                                \
    if (!(cond)) return fail(RAU_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// ------------------------------------------------------------------ ctx
struct Lin {  // one Linear (or 1x1 conv) inside a flat group
  float *W, *b, *dW, *db;
  int out, in;
};
struct Entry {
  std::string name;
  size_t off;
  int rows, cols;
};
struct Group {
  float* w = nullptr;
  float* g = nullptr;
  float* m = nullptr;  // Adam moments (allocated on first update)
  float* v = nullptr;
  size_t n = 0;
  int64_t adam_t = 0;
  std::vector<Entry> layout;
};
struct ProfRec {
  int cls;
  hipEvent_t a, b;
  int sid;   // 0 chain, 1 bulk, 2 weight-gradient stream
};
struct ProfCls {
  std::string name;
  int64_t launches = 0;
  double ms = 0, flops = 0, bytes = 0;
};

// One of the two batch slots of the asynchronous upload path (rau_set_batch_async / rau_use_batch):
// device buffers, pinned host staging the loader may fill in place, the host-side metadata
// rau_forward needs, and the two events that order uploads against the steps.
struct BatchSlot {
  float* feats = nullptr;                                   // device, [B][D][Sp]
  int32_t *tokens = nullptr, *lens_d = nullptr, *labels_d = nullptr;
  int32_t *utok = nullptr, *ustart = nullptr, *upos = nullptr;
  float* feats_h = nullptr;                                 // pinned host, dense [B][D][S]
  int32_t *tokens_h = nullptr, *lens_p = nullptr, *labels_h = nullptr;
  int32_t *utok_h = nullptr, *ustart_h = nullptr, *upos_h = nullptr;
  std::vector<int32_t> lens;
  int max_len = 0, nuniq = 0;
  bool have = false, have_labels = false;
  hipEvent_t uploaded = nullptr;    // recorded on the copy stream behind the slot's H2D copies
  hipEvent_t consumed = nullptr;    // recorded on the chain stream when the ctx switches away from the slot
  bool upload_pending = false, consumed_valid = false;
};

struct rau_ctx {
  rau_config cfg;
  int Q;
  int Sp = 0;                 // position pitch: cfg.S rounded up to a multiple of 4 (7x7 maps: 49 -> 52);
                              // every [.., S]-shaped device tensor uses it, pad columns hold zeros
  int bf16 = 0;               // cfg.dtype == RAU_BF16: bf16-operand conv GEMMs
  hipStream_t st = nullptr;    // chain stream: recurrences, small GEMMs; what callers order against
  hipStream_t st2 = nullptr;   // bulk stream: hop-batched 1x1-conv GEMMs, overlapped with the chain
  hipStream_t st3 = nullptr;   // weight-gradient stream: throughput GEMMs nobody waits for until the end
  hipEvent_t evA = nullptr, evD = nullptr, evW = nullptr, evE = nullptr, evW3 = nullptr,
             evM3 = nullptr, evEnd = nullptr, evE1 = nullptr, evHd = nullptr,
             evG0 = nullptr, evG = nullptr, evQ0 = nullptr, evQ = nullptr, evDq = nullptr;   // side-stream forks / joins
  hipEvent_t evGc[kEncSideChunks] = {};   // layer-1 input projection, chunk c done on the third stream
  int enc_chunk_tok[kEncSideChunks + 1] = {};   // first token of chunk c (last entry: TL) in the current forward
  std::vector<hipEvent_t> evH;       // per hop: forward chain done (the head stream waits on it)
  std::vector<hipEvent_t> evF, evK;  // per hop group: forward bulk done / backward chain done
  // hops per bulk launch (pipelines the bulk GEMMs with the hop loops): gsize[h] = n if hops
  // [h, h+n) form one launch group, else 0.  `groups` is the configured partition, `cur` the one
  // the last forward used (evaluate mode: one group of H).
  std::vector<int> groups, cur, bgroups;   // bgroups: backward partition (empty = same as forward)
  std::vector<void*> allocs;
  Group grp[3];
  // mult
  Lin q_proj, h_proj, i_embed, att_q, att_i, att_score, att_mem, feat_attprob, lstm_i2h,
      lstm_h2h, lstm_out, cls, do_pred;
  // rnn
  Lin i2h[2], h2h[2];
  // batch
  float* feats = nullptr;
  int32_t *tokens = nullptr, *lens_d = nullptr, *labels_d = nullptr;
  std::vector<int32_t> lens_h;
  int max_len = 0;
  bool have_batch = false, have_labels = false;
  int nuniq = 0;
  int32_t *utok = nullptr, *ustart = nullptr, *upos = nullptr;
  // asynchronous, double-buffered upload (allocated at the first rau_batch_slot / rau_set_batch_async)
  BatchSlot slot[2];
  int cur_slot = 0;
  bool async_ready = false;
  hipStream_t stc = nullptr;         // copy stream
  hipEvent_t hopw_ev[2] = {nullptr, nullptr};   // hop-weight staging slots: H2D copy done
  // dropout
  int mode = RAU_MODE_TRAIN;
  uint32_t* mbits[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t mcount[5] = {0, 0, 0, 0, 0};
  bool mexplicit[5] = {false, false, false, false, false};
  float mp[5];        // drop probability of the device's Philox masks: p quantised to 1/256
  float mp_exact[5];  // the configured p: scale 1/(1-p) of caller-supplied masks (nn.Dropout's)
  uint64_t seed = 0;
  uint32_t step = 0;
  // encoder activations
  float *we, *G1, *G2, *c1, *h1, *c2, *h2, *tc1, *tc2, *x2, *q;
  // RAU activations
  float *xd;          // [H][B][D][S] feature map after per-hop dropout (train mode)
  void* WiT16 = nullptr;  // with xd16: bf16 copies of the transposed conv weights WiT, WpT
  void* WpT16 = nullptr;
  void* dS16 = nullptr;  // with xd16: the attention backward's dS as bf16 [H][B][A][S] (both consumers round it anyway)
  bool dpre_fwd = false;    // the last forward already formed dpre / dhn of every hop (unscaled by the hop weights)
  bool ds16_step = false;   // set by rau_backward while its hop loop runs: hop_backward writes dS16, not T
  void* xd16 = nullptr;  // RAU_BF16 step path, S % 4 == 0: the same maps stored as bf16 (xd stays unwritten)
  bool I_shared = false;  // evaluate mode: i_embed output is hop-invariant, computed once
  bool yq_shared = false; // no dropout on q (evaluate mode): q_embed's question half is hop-invariant, rows of hop 0 only
  float *WiT, *WpT;   // i_embed / ifeatproj weights transposed ([D][M], [M][A]), refreshed per forward
  float *P0;          // [B][A][S] hop-invariant attention pre-activation (evaluate mode)
  float *qd, *Yq, *qf, *I, *T, *u, *zm, *a, *jv, *j, *g4, *cc, *hh, *tc, *mf, *logits,
      *dl, *lossrow, *dopred, *losses_d, *hopw_d;
  int32_t* argmax_d;
  float* att_part = nullptr;  // [B][chunks][S] partial column sums of the split attention kernels
  bool att_split_env = false; // RAU_ATT_SPLIT: 4-wave row-chunk attention kernels instead of the fused ones
  // persistent encoder forward (enc_ws.hip): device error word (a bounded spin gave up), copied to pinned memory behind the launch
  bool enc_ws = false;          // weight-stationary persistent encoder forward (enc_ws.hip): evaluate mode
  bool enc_ws_train = false;    // ... and in training steps
  int side_split_env = -1;      // RAU_SIDE_SPLIT=0|1 (A/B variable); -1 = by shape, see side_split()
  unsigned* ws_cnt = nullptr;   // its 16 progress counters
  int* perr_d = nullptr;
  int* perr_h = nullptr;
  bool persist_used = false;
  float* hopw_h = nullptr;    // pinned staging of the hop weights, 2 slots of H
  int hopw_slot = 0;
  // backward temporaries
  // dZ holds dI (gradient at i_embed's OUTPUT); the tanh derivative is applied by its consumers
  float *dpre, *dhn, *dg4, *dcn[2], *dhp[2], *dj, *da_lin, *dz, *du, *dwsp, *dZ,
      *dqt, *dQD, *dq, *slab, *slab2, *slab3, *coltmp3, *tmpS, *coltmp2, *dbi_part;
  size_t slab3_floats = 0, slab2_floats = 0;
  float *dG1, *dG2, *dwe, *edc[2][2];
  size_t slab_floats = 0;
  // module-level entry points (rau_modules.hip); allocated on first use
  bool mod_ready = false;
  float *m_state = nullptr, *m_dstate = nullptr;   // [T][B][Q] packed DeepLSTM state slots
  float *m_tmp[4] = {nullptr, nullptr, nullptr, nullptr};  // [B][max(Rq,R,M)] scratch
  float *m_dq = nullptr, *m_dc = nullptr, *m_dh = nullptr;  // [H][B][Q], [H][B][R], [H][B][R]
  float *m_Xp = nullptr, *m_a = nullptr, *m_da = nullptr, *m_dXd = nullptr;  // re-pitching (S % 4 != 0)
  float *m_dX = nullptr, *m_dZ = nullptr;          // [B][D][S], [B][M][S]: feature-map gradient, on request
  float *m_add = nullptr, *m_s = nullptr, *m_zero = nullptr;  // [B][M], [B], zeros [B][max(Q,R)]
  float *m_loss = nullptr;                         // [H] criterion outputs
  uint64_t mod_masks_seed = 0;                     // (seed, step) the device masks were drawn for
  uint32_t mod_masks_step = 0;
  bool mod_masks_valid = false;
  // native data-parallel exchange (rau_comm_*; RCCL loaded with dlopen on first use)
  void* comm = nullptr;          // ncclComm_t
  hipStream_t st_comm = nullptr;
  hipEvent_t evC = nullptr;
  int comm_ranks = 0;
  // hipGraph replay of a whole step (rau_graph_step): one executable graph per step "shape"
  uint64_t* dkey = nullptr;      // device copy of (seed, step): what fill_masks reads
  bool capturing = false;
  bool graph_last = false;       // the last backward ran inside a graph (its events are graph-internal)
  std::vector<std::pair<uint64_t, hipGraphExec_t>> graphs;
  // update
  float *npart = nullptr, *norms_d = nullptr;
  bool fwd_done = false, bwd_done = false;
  // timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool prof_on = false;
  bool prof_sparse = false;   // rau_prof_enable(ctx, 2): chain-stream launches are NOT bracketed except phase
                              // markers, so the recurrence runs at its un-profiled speed under the timeline
  std::vector<ProfCls> pcls;
  std::vector<ProfRec> precs;
  std::vector<hipEvent_t> evpool;
};

// Effective drop probability of a mask site.  Masks the device draws itself (Philox, 8-bit draws)
// drop with p quantised to 1/256 and scale by 1/(1-pq), so that E[mask * scale] = 1 exactly;
// caller-supplied masks (rau_set_mask) are nn.Dropout's: scale 1/(1-p) with the configured p.
// Non-recurrent GEMMs of the recurrence's stream (layer-1 input projection beyond the first tokens,
// q_embed's question half of hops 1.., the dq terms of finished backward groups) run on the
// weight-gradient stream where the recurrence is the longer path: evaluate mode, bf16 mode (forward
// phase bound by the encoder), contexts of up to 64 samples.  In the f32 step at 256 samples the bulk
// stream is the longer path and the extra concurrency costs it 1 %.
inline bool chain_bound(const rau_ctx* ctx) {
  return ctx->mode == RAU_MODE_EVAL || ctx->bf16 || ctx->cfg.B <= 64;
}
inline bool side_split(const rau_ctx* ctx) {
  if (ctx->side_split_env >= 0) return ctx->side_split_env != 0;
  return chain_bound(ctx);
}
// The recurrence's skinny GEMMs with 32-deep stages (skinny_dma.hip, NH = 2) under the same predicate
// (RAU_SKINNY_DEEP=0|1 overrides): set for the calling thread at every step-level entry point.
inline void set_skinny_policy(const rau_ctx* ctx) {
  static const int env = [] { const char* e = std::getenv("RAU_SKINNY_DEEP"); return e ? (std::atoi(e) != 0 ? 1 : 0) : -1; }();
  skinny_dma_set_deep(env >= 0 ? env : (chain_bound(ctx) ? 1 : 0));
  lin_set_bf16(ctx->bf16 == 1);
}
inline float mask_p(const rau_ctx* ctx, int site) {
  return ctx->mexplicit[site] ? ctx->mp_exact[site] : ctx->mp[site];
}

template <typename Tp>
static int dalloc(rau_ctx* c, Tp** p, size_t count) {
  void* d = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(Tp);
  hipError_t e = hipMalloc(&d, bytes);
  if (e != hipSuccess)
    return fail(RAU_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  e = hipMemsetAsync(d, 0, bytes, c->st);
  if (e != hipSuccess) return fail(RAU_ERR_DEVICE, "hipMemsetAsync: %s", hipGetErrorString(e));
  c->allocs.push_back(d);
  *p = reinterpret_cast<Tp*>(d);
  return 0;
}

struct LayoutBuilder {
  Group* g;
  size_t off = 0;
  Lin take(const char* name, int out, int in) {
    Lin l;
    l.out = out;
    l.in = in;
    g->layout.push_back({std::string(name) + ".weight", off, out, in});
    l.W = reinterpret_cast<float*>(off);
    off += (size_t)out * in;
    g->layout.push_back({std::string(name) + ".bias", off, out, 1});
    l.b = reinterpret_cast<float*>(off);
    off += out;
    return l;
  }
};
static inline void bind(Lin& l, const Group& g) {
  const size_t ow = reinterpret_cast<size_t>(l.W), ob = reinterpret_cast<size_t>(l.b);
  l.W = g.w + ow;
  l.b = g.w + ob;
  l.dW = g.g + ow;
  l.db = g.g + ob;
}

static inline int prof_class(rau_ctx* c, const char* name) {
  for (size_t i = 0; i < c->pcls.size(); ++i)
    if (c->pcls[i].name == name) return (int)i;
  c->pcls.push_back(ProfCls{name});
  return (int)c->pcls.size() - 1;
}
static inline hipEvent_t prof_event(rau_ctx* c) {
  if (!c->evpool.empty()) {
    hipEvent_t e = c->evpool.back();
    c->evpool.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

// chain-stream classes the sparse timeline keeps (one launch each per phase boundary)
static inline bool prof_marker(const char* n) {
  for (const char* m : {"embed_fwd", "gather_q", "loss_reduce", "scale_hops", "dq_reduce", "embed_bwd"})
    if (std::strcmp(n, m) == 0) return true;
  return false;
}
// Measurement hook of the DEVELOPMENT build only (make dev -> librau_dev.so, -DRAU_DEV_HOOKS; the shipped
// library compiles it to `false`): RAU_DEV_SKIP=class1,class2,.. drops every launch of the named kernel
// classes.  Timing only -- the numerics of such a run are meaningless -- used for the per-class
// sensitivity tables (tools/dev_skip.sh, profiles/r04_forward_gap.md).  Consumers of split-K partials
// whose producer was skipped see a count of 0 or fail closed (split_guard.hip), never read out of bounds.
#ifdef RAU_DEV_HOOKS
static inline bool rau_dev_skipped(const char* cname) {
  static const std::string list = [] { const char* e = std::getenv("RAU_DEV_SKIP"); return std::string(e ? e : ""); }();
  if (list.empty()) return false;
  size_t pos = 0;
  const std::string n(cname);
  while (pos <= list.size()) {
    const size_t c = list.find(',', pos);
    const std::string tok = list.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
    if (tok == n) return true;
    if (c == std::string::npos) break;
    pos = c + 1;
  }
  return false;
}
#define RAU_DEV_SKIPPED(cname) rau_dev_skipped(cname)
#else
#define RAU_DEV_SKIPPED(cname) false
#endif
// Launch wrapper: counts launches/FLOPs/bytes per kernel class and, when
// profiling is on, brackets the launch with HIP events on the ctx stream.
#define RUN(cname, fl, by, expr) RUNS(ctx->st, cname, fl, by, expr)
#define RUNS(rstream, cname, fl, by, expr)                                                \
  do {                                                                                    \
    ProfRec pr_;                                                                          \
    int pc_ = -1;                                                                         \
    if (ctx->prof_on && !(ctx->prof_sparse && (rstream) == ctx->st && !prof_marker(cname))) { \
      pc_ = prof_class(ctx, cname);                                                       \
      ctx->pcls[pc_].launches++;                                                          \
      ctx->pcls[pc_].flops += (double)(fl);                                               \
      ctx->pcls[pc_].bytes += (double)(by);                                               \
      pr_.cls = pc_;                                                                      \
      pr_.a = prof_event(ctx);                                                            \
      pr_.b = prof_event(ctx);                                                            \
      pr_.sid = (rstream) == ctx->st ? 0 : (rstream) == ctx->st2 ? 1 : 2; \
      hipEventRecord(pr_.a, rstream);                                                     \
    }                                                                                     \
    hipError_t e_ = RAU_DEV_SKIPPED(cname) ? hipSuccess : (expr);                         \
    if (pc_ >= 0) {                                                                       \
      hipEventRecord(pr_.b, rstream);                                                     \
      ctx->precs.push_back(pr_);                                                          \
    }                                                                                     \
    if (e_ == kSplitStateError)   /* a consumer of split-K partials refused a stale span */ \
      return fail(RAU_ERR_STATE, "kernel %s: split-K partials outside their workspace, "  \
                  "nothing launched (%s:%d)", cname, __FILE__, __LINE__);                  \
    if (e_ != hipSuccess)                                                                 \
      return fail(RAU_ERR_DEVICE, "kernel %s: %s (%s:%d)", cname, hipGetErrorString(e_),  \
                  __FILE__, __LINE__);                                                    \
  } while (0)

// ---- shared between the step-level path (rau_ctx.hip) and the module-level entry
// points (rau_modules.hip)
struct HopGrad {
  const float* dl;        // [B,K] gradient at the logits (already scaled by the hop weight)
  const float* dc_next;   // [B,R] or null (zeros)
  const float* dh_next;
  const float* dmf_add;   // [B,M] or null: extra gradient at merge_feat before its dropout
  const float* da_out;    // [B,S] or null: gradient at the attprob output
  float* dc_out;          // [B,R] out: gradient at prev_c
  float* dh_out;          // [B,R] out: gradient at prev_h (written only when dh_part_out is null)
  // step-level fast path (rau_backward): what does not depend on the recurrence is formed for
  // all hops up front, and dh_prev travels from hop to hop as K-split partials
  int dpre_ready;         // ctx->dpre rows of this hop already hold (dl Wc) (.) mask
  const float* dhn_all;   // [H*B,R] dpre Wo for all hops (with dpre_ready)
  const float* dh_part;   // [dh_part_ns][B,R] partials of the gradient at next_h (or null)
  int dh_part_ns;
  float** dh_part_out;    // non-null: leave dh_prev as partials and report them here
  int* dh_part_ns_out;
  hipEvent_t ev_conv_ready;  // non-null: recorded on the chain stream right behind att_bwd -- everything the
                             // bulk stream's conv gradients of this hop read (dS, dj) exists from there on
  bool dh_prev_dead;      // the gradient at prev_h has no consumer (step-level hop 0: the initial state is a
                          // constant): its three products are not formed
};
__attribute__((visibility("hidden"))) int hop_forward(rau_ctx* ctx, int h, const float* cp,
    const float* hp, float* c_out, float* h_out, const float* Ih, const float* Pin,
    const int32_t* labels);
__attribute__((visibility("hidden"))) int hop_forward_chain(rau_ctx* ctx, int h, const float* cp,
    const float* hp, float* c_out, float* h_out, const float* Ih, const float* Pin);
__attribute__((visibility("hidden"))) int hop_forward_head(rau_ctx* ctx, hipStream_t s, float* ws,
    size_t reg, int h0, int nh, const int32_t* labels);
__attribute__((visibility("hidden"))) int hop_backward(rau_ctx* ctx, int h, const float* cp,
    const float* Ih, const HopGrad& g);
