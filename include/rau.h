/*
 * rau.h -- C ABI of librau.so: the MI355X-native Recurrent Answering Unit
 * forward/backward (hand-written HIP for gfx950 behind plain pointers).
 *
 * This is the drop-in boundary for ONE path of HyeonwooNoh/RAU_VQA: the tensor
 * half of `feval` (reference experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua
 * :428-596, "SS" below) plus the network it drives (SS:198-316,
 * model/ATTLSTM.lua, model/DeepLSTM.lua).  The reference reaches that path
 * through the Torch7 nn.Module protocol, not through an FFI of its own, so each
 * entry point below cites the nn.Module call sites it replaces.  A LuaJIT
 * `ffi.cdef` shim that re-presents these calls as :forward/:backward/
 * :getParameters objects is in bindings/rau.lua; INTEGRATION.md shows the
 * reference-side patch.
 *
 * Conventions
 *  - every function returns 0 on success or a negative rau_status; the message
 *    for the calling thread's last failure is rau_last_error().  Nothing throws
 *    or longjmps across the boundary (Lua `error()` is raised by the shim).
 *  - no global state: everything hangs off an opaque rau_ctx (one per GPU,
 *    driven by one host thread at a time -- same rule as a Lua state).
 *  - the ctx owns all device memory.  Host pointers passed in are owned by the
 *    caller and are consumed before the call returns unless stated otherwise.
 *  - all work is enqueued on the ctx's HIP stream; host-visible results are
 *    valid after rau_sync() (or any call documented as synchronising).
 *  - ids are 1-based like the reference's Lua tensors: token ids 1..V with
 *    1 = ZEROPAD (utils/vqa_prepro_loader.lua:1393), answer ids 1..K, and the
 *    returned argmax ids are 1..K with torch.max's first-max tie rule (SS:488).
 *  - there is NO CPU fallback: rau_create fails if no gfx950 device is usable.
 */
#ifndef RAU_H
#define RAU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAU_ABI_VERSION 5

typedef enum rau_status {
  RAU_OK = 0,
  RAU_ERR_INVALID = -1,   /* bad argument / shape / id out of range */
  RAU_ERR_DEVICE = -2,    /* HIP error, no device, wrong architecture */
  RAU_ERR_STATE = -3,     /* call order (e.g. backward before forward) */
  RAU_ERR_NOMEM = -4
} rau_status;

typedef enum rau_group {   /* flat parameter groups, SS:322-324 getParameters() */
  RAU_GROUP_EMBED = 0,     /* protos.word_embed */
  RAU_GROUP_RNN = 1,       /* protos.rnn  (DeepLSTM) */
  RAU_GROUP_MULT = 2       /* protos.multimodal */
} rau_group;

typedef enum rau_mode {    /* m:training() / m:evaluate(), SS:449-450,479,648-649,676 */
  RAU_MODE_TRAIN = 0,
  RAU_MODE_EVAL = 1
} rau_mode;

typedef enum rau_mask_site {   /* the five nn.Dropout sites on the path */
  RAU_MASK_WE = 0,   /* [T,B,E]   word_embed Dropout(0.5), SS:205 */
  RAU_MASK_RNN = 1,  /* [T,B,Rq]  DeepLSTM inter-layer dropout, DeepLSTM.lua:39 */
  RAU_MASK_Q = 2,    /* [H,B,Q]   q_embed Dropout(0.5), SS:233 */
  RAU_MASK_X = 3,    /* [H,B,D,S] i_embed Dropout(0.5) on the feature map, SS:239 */
  RAU_MASK_MF = 4    /* [H,B,M]   classifier merge_feat Dropout(0.5), SS:277 */
} rau_mask_site;

typedef enum rau_dtype {
  RAU_F32 = 0,       /* f32 operands, f32 MFMA accumulate (exact fmaf chain) */
  RAU_BF16 = 1,      /* BASELINE.json configs[2] (Ours_ResNet 14x14x2048, "bf16 MFMA gate/classifier
                      * GEMMs"): every GEMM on the path takes bf16-rounded operands (round to nearest
                      * even) with f32 accumulation -- the five 1x1-conv GEMMs (i_embed, ifeatproj and
                      * their gradients: 94-98 % of the FLOPs) and the three products of every Linear
                      * layer (LSTM gates, hop projections, classifier: y = x W^T, dx = dy W,
                      * dW = dy^T x).  Biases, the cell / attention / softmax / loss arithmetic and
                      * the parameters, states and gradients in memory stay f32; the one-column
                      * Linears (att_score, out_do_pred) are dot products in f32.  Where the 14x14
                      * bf16 tiles do not apply (M % 128, A % 32) the attention input gradient stays
                      * exact f32; the emulating oracle (oracle/ref_torch.py bf16=True) follows the
                      * same rule. */
  RAU_F32S = 2       /* same five GEMMs with every f32 operand SPLIT into three bf16 terms
                      * (hi + mid + lo = all 24 significand bits) and six bf16 MFMA products per
                      * operand pair, f32 accumulate: f32-grade accuracy (dropped terms <= 2^-24
                      * relative) on the 16x faster bf16 matrix pipe.  Not bitwise the fmaf chain
                      * of RAU_F32; selectable, never the default. */
} rau_dtype;

/* Network hyper-parameters: the hard-coded locals of SS:202,209-229 plus the
 * data-defined sizes.  Q (question state width) is 4*Rq: 2 layers x {c,h}. */
typedef struct rau_config {
  int32_t B;    /* batch size per GPU (opt.batch_size) */
  int32_t T;    /* seq_len: rows of the token matrix x[T,B] */
  int32_t V;    /* vocab_size incl. ZEROPAD */
  int32_t E;    /* embed_dim = 200, SS:202 */
  int32_t Rq;   /* rnn_size = 512, SS:209 (nrnn_layer fixed at 2, SS:210) */
  int32_t D;    /* cnnout_dim 512 | 2048, SS:216 */
  int32_t S;    /* cnnout_w*cnnout_h = 196 (14x14) or 49 (7x7, the scripts' default), SS:219;
                 * not a multiple of 4: padded internally (callers always see dense [.., S]) */
  int32_t M;    /* multfeat_dim = 512, SS:220 */
  int32_t A;    /* attfeat_dim = 256, SS:221 */
  int32_t R;    /* att_rnn_size = 512, SS:225 (1 layer, dropout 0) */
  int32_t K;    /* answer_size = 1000, SS:222 */
  int32_t H;    /* nHop */
  float p_we, p_rnn, p_q, p_x, p_mf;  /* dropout probabilities, 0.5 each (device masks resolve p to
                                       * 1/256; the 1/(1-p) scale uses the same quantised p) */
  int32_t dtype;      /* rau_dtype */
  int32_t device_id;  /* HIP device ordinal (opt.gpuid) */
} rau_config;

typedef struct rau_ctx rau_ctx;

/* Fills *cfg with the reference's defaults (Ours_SS, 14x14x512, nhop 8). */
void rau_default_config(rau_config* cfg);

const char* rau_last_error(void);
int rau_abi_version(void);

/* Builds the three protos + their clones (SS:200-347): allocates parameters,
 * gradients, activations for T tokens and H hops, and the ctx stream. */
int rau_create(const rau_config* cfg, rau_ctx** out);
void rau_destroy(rau_ctx* ctx);

/* ---- parameters: m:getParameters(), SS:322-324 ------------------------------
 * Flat DEVICE buffers (weights, gradients) of a group and its length in floats.
 * Layout: rau_layout_entry() lists each tensor; weight [out,in] then bias [out]
 * per layer (DESIGN.md section "Flat parameter layout"). */
int rau_params(rau_ctx* ctx, int group, float** weights, float** grads, size_t* n);
int rau_layout_count(const rau_ctx* ctx, int group);
int rau_layout_entry(const rau_ctx* ctx, int group, int index, const char** name,
                     size_t* offset, int32_t* rows, int32_t* cols);
/* host <-> device copies of a whole group (synchronising) */
int rau_set_params(rau_ctx* ctx, int group, const float* host, size_t n);
int rau_get_params(rau_ctx* ctx, int group, float* host, size_t n);
int rau_get_grads(rau_ctx* ctx, int group, float* host, size_t n);
int rau_set_grads(rau_ctx* ctx, int group, const float* host, size_t n);
/* param:uniform(lo,hi) on each flat vector, SS:352-354 (Philox, not Torch's MT) */
int rau_init_uniform(rau_ctx* ctx, uint64_t seed, float lo, float hi);
/* embed_grad:zero() rnn_grad:zero() mult_grad:zero(), SS:429-431 */
int rau_zero_grads(rau_ctx* ctx);

/* ---- mode and dropout --------------------------------------------------------
 * :training()/:evaluate() on every clone.  In TRAIN mode the five dropout sites
 * draw masks either from explicit keep flags (rau_set_mask, parity tests) or
 * from the Philox4x32-10 stream keyed by (seed, site, step) (rau_set_dropout_seed;
 * regenerated on device inside rau_forward). */
int rau_set_mode(rau_ctx* ctx, int mode);
int rau_set_dropout_seed(rau_ctx* ctx, uint64_t seed, uint32_t step);
/* keep: uint8 0/1 flags for the whole site in the shape listed at rau_mask_site;
 * n = element count.  Switches that site to explicit masks until
 * rau_set_dropout_seed is called again. */
int rau_set_mask(rau_ctx* ctx, int site, const uint8_t* keep, size_t n);
/* reads back the keep flags the next/last forward uses (tests; synchronising) */
int rau_get_mask(rau_ctx* ctx, int site, uint8_t* keep, size_t n);

/* ---- batch: what next_batch_feat returns + the H2D of SS:434-439 -------------
 * feats [B,D,S] float (NCHW with W*H flattened), tokens [T,B] int32, lens [B]
 * int32 (0..T), labels [B] int32 (1..K) or NULL for inference.  Copies to the
 * ctx's device buffers; also builds the per-token position index that makes
 * the LookupTable gradient a deterministic gather-sum. */
int rau_set_batch(rau_ctx* ctx, const float* feats, const int32_t* tokens,
                  const int32_t* lens, const int32_t* labels);
/* device pointer of the resident feature buffer (producer may write it directly) */
int rau_batch_feats(rau_ctx* ctx, float** feats_dev);

/* ---- asynchronous, double-buffered upload: SS:434-439 behind the loader's prefetch -------------
 * The reference re-uploads feats / x / x_len / y every iteration (SS:434-439) while its loader's
 * worker thread assembles the next batch (utils/vqa_prepro_loader.lua:931-958).  Here the ctx owns
 * TWO batch slots, each with device buffers and PINNED host staging:
 *   rau_batch_slot(slot)       -> host pointers of the slot's staging (feats [B,D,S], tokens [T,B],
 *                                 lens [B], labels [B]); the loader assembles the batch in place.
 *                                 Waits (on the host) until the slot's previous upload has left it:
 *                                 CALL IT BEFORE EVERY IN-PLACE REFILL, never cache its pointers across
 *                                 iterations (the host may run two steps ahead of the copy stream).
 *   rau_set_batch_async(slot, feats, tokens, lens, labels, has_labels)
 *                              -> checks the ids, builds the token index (host), enqueues the H2D
 *                                 copies on a dedicated copy stream and returns; no stream is
 *                                 synchronised.  A NULL pointer = "already in the slot's staging";
 *                                 a non-NULL one is copied into it first (one host memcpy).
 *                                 has_labels tells whether in-place labels are present.
 *   rau_use_batch(slot)        -> the slot becomes the resident batch; the step's streams are ordered
 *                                 behind its upload by an event, not by a host wait.
 * Upload batch n+1 into the other slot while step n runs; the copy stream itself waits for the last
 * step that read the slot being refilled.  rau_set_batch stays the synchronous form (it writes the
 * current slot). */
int rau_batch_slot(rau_ctx* ctx, int slot, float** feats_host, int32_t** tokens_host,
                   int32_t** lens_host, int32_t** labels_host);
int rau_set_batch_async(rau_ctx* ctx, int slot, const float* feats, const int32_t* tokens,
                        const int32_t* lens, const int32_t* labels, int has_labels);
int rau_use_batch(rau_ctx* ctx, int slot);

/* ---- the hot path ------------------------------------------------------------
 * rau_forward : SS:443-520  encoder unroll, length select, H-hop RAU, per-hop
 *               CrossEntropyCriterion forward, first-max argmax.
 * rau_backward: SS:561-596  per-hop criterion backward scaled by hop_w[h]
 *               (SS:569 nHop / MS:568-570 one / Full:587-589 0|1), RAU BPTT,
 *               d_do_pred*0 and gradattprob=0, dq=sum over hops, encoder BPTT,
 *               LookupTable scatter.  Gradients ACCUMULATE into the flat grad
 *               buffers like accGradParameters; call rau_zero_grads first. */
int rau_forward(rau_ctx* ctx);
int rau_backward(rau_ctx* ctx, const float* hop_w /* [H] host */);

/* ---- module-level entry points: one call per nn.Module :forward / :backward -----
 * For hosts that keep feval's own loops (SS:443-596) and call the clones one by one.
 * t in [0,T) / h in [0,H) select the clone (the reference's embed_clones[t+1],
 * lstm_clones[t+1], multimodal_clones[h+1], criteria[h+1]; SS:340-347): it fixes the
 * clone's dropout-mask slice and the ctx-owned slot its activations are saved in.
 * All tensor arguments are DEVICE pointers (16-byte aligned, dense row-major), in the
 * reference's table orders; NULL for an optional input means "zeros" (or, for tokens /
 * X / labels, "the resident batch of rau_set_batch").  Outputs are returned as
 * pointers to ctx-owned slots indexed by (t|h): valid until the next :forward /
 * :backward of the same clone -- the lifetime rule of nn.Module's self.output /
 * self.gradInput.  :backward ACCUMULATES the clone's parameter gradients into the flat
 * gradient buffers (accGradParameters).  Work is enqueued on the ctx stream; nothing
 * synchronises except rau_criterion_forward when `loss` is non-NULL.
 * The step-level rau_forward/rau_backward above compute the same values faster
 * (they batch across clones); do not interleave the two within one step. */

/* word_embed = LookupTable -> Dropout(0.5) -> Tanh, SS:203-206; :forward SS:451,
 * :backward SS:593 (LookupTable has no gradInput).  tokens_dev [B] int32 1-based.
 * Ids in DEVICE tensors (tokens_dev here, labels_dev of rau_criterion_*) cannot be range-checked
 * by the call: the kernels clamp them into [1, V] / [1, K], so a bad id uses a wrong row where
 * the reference's LookupTable / ClassNLLCriterion would raise -- it never faults the GPU.  (The
 * host arrays of rau_set_batch ARE checked and rejected with RAU_ERR_INVALID.) */
int rau_embed_forward(rau_ctx* ctx, int t, const int32_t* tokens_dev, float** we /* [B,E] */);
int rau_embed_backward(rau_ctx* ctx, int t, const int32_t* tokens_dev, const float* d_we);

/* DeepLSTM.create(E, Rq, 2, 0.5), model/DeepLSTM.lua:14-71: {x [B,E], state [B,4Rq] =
 * [c1 h1 c2 h2]} -> state' [B,4Rq]; :forward SS:452, :backward SS:592 returning
 * {d_x [B,E], d_state [B,4Rq]}.  The length-select / dq row replacement of SS:455-461,
 * 584-591 stays in the host loop, as in the reference. */
int rau_deeplstm_forward(rau_ctx* ctx, int t, const float* x, const float* state,
                         float** state_out);
int rau_deeplstm_backward(rau_ctx* ctx, int t, const float* x, const float* state,
                          const float* d_state_out, float** d_x, float** d_state);

/* protos.multimodal, SS:292-307: {q [B,4Rq], X [B,D,S], c [B,R], h [B,R]} ->
 * {logits [B,K], do_pred [B], attprob [B,S], c' [B,R], h' [B,R]}; :forward SS:480,
 * :backward SS:571 with gradOutput {d_logits, d_do_pred, d_attprob, d_c, d_h}
 * (d_do_pred / d_attprob may be NULL = zeros, which is what feval passes, SS:566,573)
 * returning {d_q, d_X, d_c, d_h}.  d_X is the gradient feval discards (SS:579): pass
 * d_X = NULL to skip computing it. */
int rau_multimodal_forward(rau_ctx* ctx, int h, const float* q, const float* X,
                           const float* c_prev, const float* h_prev, float** logits,
                           float** do_pred, float** attprob, float** c_out, float** h_out);
int rau_multimodal_backward(rau_ctx* ctx, int h, const float* q, const float* X,
                            const float* c_prev, const float* h_prev, const float* d_logits,
                            const float* d_do_pred, const float* d_attprob,
                            const float* d_c, const float* d_h, float** d_q, float** d_X,
                            float** d_c_prev, float** d_h_prev);

/* nn.CrossEntropyCriterion (sizeAverage), SS:310: :forward SS:518 -> *loss (host,
 * may be NULL); :backward SS:565-569 -> d_logits [B,K] = scale*(softmax-onehot)/B,
 * scale = the dpred:mul(w) of SS:569.  labels_dev [B] int32 1-based. */
int rau_criterion_forward(rau_ctx* ctx, int h, const float* logits, const int32_t* labels_dev,
                          float* loss);
int rau_criterion_backward(rau_ctx* ctx, int h, const float* logits,
                           const int32_t* labels_dev, float scale, float** d_logits);

/* ---- device tensors: the tensor algebra feval does BETWEEN module calls ----------------------
 * The reference's loops copy state rows where x_len[k] == t (`rnn_out[k] = lst[k]`, SS:455-461;
 * the dq row replacement, SS:584-591), accumulate (`uni_pred:add(pred[1])`, SS:522-526), take
 * `torch.max(pred[1], 2)` and `ans:eq(y):sum()` (SS:488-492) and zero-fill state tensors
 * (SS:357-413) -- on the reference's CUDA box through cutorch.  An MI355X host has no cutorch, so
 * those few operations are exported on plain device pointers (dense row-major float / int32),
 * enqueued on the ctx stream.  rau_dev_alloc'ed memory is zero-filled and owned by the ctx (freed
 * by rau_destroy, or earlier by rau_dev_free).  rau_dev_sum / _count_eq / _upload / _download
 * synchronise; the others do not. */
int rau_dev_alloc(rau_ctx* ctx, size_t n_floats, float** out);
int rau_dev_free(rau_ctx* ctx, float* p);
int rau_dev_fill(rau_ctx* ctx, float* dst, size_t n, float value);
int rau_dev_copy(rau_ctx* ctx, float* dst, const float* src, size_t n);
int rau_dev_axpy(rau_ctx* ctx, float* y, const float* x, size_t n, float alpha);   /* y += alpha x */
int rau_dev_scale(rau_ctx* ctx, float* x, size_t n, float alpha);
/* The tensor statements of the reference's optimizer call, `adam(x, dx, lr, beta1, beta2, epsilon,
 * state)` on each flat vector (SS:770-772, utils/optim_updates.lua:59-87): y += alpha x1 x2,
 * y += alpha x1 / x2, in-place sqrt, x += value -- and the whole function as one pass,
 * rau_dev_adam: m = beta1 m + (1-beta1) dx; v = beta2 v + (1-beta2) dx^2;
 * x -= lr sqrt(1-beta2^t)/(1-beta1^t) m / (sqrt(v) + eps), t = state.t after its increment (>= 1). */
int rau_dev_addcmul(rau_ctx* ctx, float* y, float alpha, const float* x1, const float* x2, size_t n);
int rau_dev_addcdiv(rau_ctx* ctx, float* y, float alpha, const float* x1, const float* x2, size_t n);
int rau_dev_sqrt(rau_ctx* ctx, float* x, size_t n);
int rau_dev_add_scalar(rau_ctx* ctx, float* x, size_t n, float value);
int rau_dev_adam(rau_ctx* ctx, float* x, const float* dx, float* m, float* v, size_t n, float lr,
                 float beta1, float beta2, float eps, int32_t t);
/* dst[k,:] = src[k,:] for the rows k with key_dev[k] == value (SS:455-461, 584-591) */
int rau_dev_select_rows(rau_ctx* ctx, float* dst, const float* src, int32_t rows, int32_t cols,
                        const int32_t* key_dev, int32_t value);
/* torch.max(x, 2): per-row maximum and FIRST maximal index, 1-based (either output may be NULL) */
int rau_dev_rowmax(rau_ctx* ctx, const float* x, int32_t rows, int32_t cols, float* max_dev,
                   int32_t* argmax_dev);
int rau_dev_sum(rau_ctx* ctx, const float* x, size_t n, double* out_host);
int rau_dev_count_eq(rau_ctx* ctx, const int32_t* a_dev, const int32_t* b_dev, int32_t n,
                     int32_t* count_host);
int rau_dev_upload(rau_ctx* ctx, void* dst_dev, const void* host, size_t bytes);
int rau_dev_download(rau_ctx* ctx, void* host, const void* src_dev, size_t bytes);

/* rau_zero_grads (optional) + rau_forward + rau_backward as ONE hipGraph launch: the three
 * streams, their fork/join events and all kernel arguments are captured once per step shape
 * (mode, longest question, number of active hops, explicit-mask sites) and replayed; the
 * batch, the Philox key and hop_w are read from device memory, so they may change freely
 * between launches.  Same results as the two calls, bit for bit. */
int rau_graph_step(rau_ctx* ctx, const float* hop_w /* [H] host */, int zero_grads_first);

/* ---- results (valid after rau_sync; these calls synchronise themselves) ------ */
int rau_sync(rau_ctx* ctx);
int rau_get_losses(rau_ctx* ctx, float* losses /* [H] */);
int rau_get_argmax(rau_ctx* ctx, int32_t* ans /* [H,B] 1-based */);
int rau_get_logits(rau_ctx* ctx, float* logits /* [H,B,K] */);
int rau_get_dopred(rau_ctx* ctx, float* dopred /* [H,B] */);
int rau_get_attention(rau_ctx* ctx, float* att /* [H,B,S] */);
int rau_get_question_state(rau_ctx* ctx, float* q /* [B,Q] */);
int rau_get_att_state(rau_ctx* ctx, float* c /* [H,B,R] */, float* h /* [H,B,R] */);

/* ---- update: SS:597-630 + utils/optim_updates.lua:59-87 (row "next-1") -------
 * Gradient noise N(0, eta/((step_t+1)*gamma)), per-group L2 clip, Adam with
 * epsilon outside the sqrt; lr applies to EMBED and RNN, mult_lr to MULT
 * (SS:770-772).  noise_seed keys the on-device normal generator; eta = 0
 * disables noise.  out_norms[3] (host, may be NULL) receives pre-clip norms. */
int rau_noise_clip_adam(rau_ctx* ctx, int64_t step_t, float lr, float mult_lr,
                        float beta1, float beta2, float eps, float eta,
                        float gamma, float clip, uint64_t noise_seed,
                        float* out_norms);

/* ---- data-parallel exchange (new: the reference is single-GPU) --------------------
 * One process and one rau_ctx per GPU.  Rank 0 obtains a communicator id and hands it to
 * the other ranks by any host channel (file, socket, MPI, torch.distributed); every rank
 * then calls rau_comm_init.  rau_allreduce_grads averages the three flat gradient buffers
 * in place over RCCL/xGMI after rau_backward -- the mult bucket underneath the encoder
 * BPTT, everything ordered by stream events -- after which every rank applies the
 * identical rau_noise_clip_adam (same noise_seed).  RCCL is bound at first use. */
#define RAU_COMM_ID_BYTES 128
int rau_comm_unique_id(void* id, size_t bytes /* >= RAU_COMM_ID_BYTES */);
int rau_comm_init(rau_ctx* ctx, int nranks, int rank, const void* id, size_t bytes);
int rau_allreduce_grads(rau_ctx* ctx);
int rau_comm_destroy(rau_ctx* ctx);

/* ---- timing / interop ---------------------------------------------------------
 * The ctx's hipStream_t (as void*) so a host can order its own work (e.g. an
 * RCCL all-reduce of the flat grad buffers issued through torch.distributed)
 * after the ctx's kernels without a device-wide sync. */
int rau_stream(rau_ctx* ctx, void** hip_stream);
/* Makes `hip_stream` (a hipStream_t of the same device) wait until the gradients of
 * `group` from the last rau_backward are final, without blocking the host or the ctx
 * stream.  RAU_GROUP_MULT is final BEFORE the encoder BPTT has run, so a host can put
 * that bucket's all-reduce on a side stream underneath the rest of the backward pass
 * (rau_vqa_amd/dist.py); the other two groups are final at the end of rau_backward. */
int rau_wait_grads(rau_ctx* ctx, int group, void* hip_stream);
/* HIP-event bracket on the ctx stream: kernel time of everything enqueued
 * between begin and end, in milliseconds (end synchronises). */
int rau_timer_begin(rau_ctx* ctx);
int rau_timer_end(rau_ctx* ctx, float* ms);
/* Average duration (ms) and launch count of the named kernel class since the
 * last rau_prof_reset, measured with HIP events on the ctx stream when
 * profiling is enabled (adds two events per launch; off by default).
 * on = 2: sparse -- only the launches of the bulk and weight-gradient streams and the
 * recurrence's phase markers are bracketed, so the recurrence keeps its un-profiled pace. */
int rau_prof_enable(rau_ctx* ctx, int on);
int rau_prof_reset(rau_ctx* ctx);
int rau_prof_count(rau_ctx* ctx);
int rau_prof_entry(rau_ctx* ctx, int index, const char** name, int64_t* launches,
                   double* total_ms, double* flops, double* bytes);

/* ---- diagnostics: host-side predicates of the library, callable without a device --------------
 * (what tests/test_host_logic.py pins on the CPU; nothing here touches the GPU or a ctx)
 *
 * rau_split_guard_check: the range check every consumer of split-K partial sums applies before it
 * launches (lin_reduce_epilogue, the LSTM cell kernels, the attention kernels, the criterion head,
 * splitk_reduce_acc): would a consumer reading `nsplit` partials of `per_split_floats` floats at
 * `offset` floats into a workspace of `ws_floats` floats be launched?  RAU_OK, or RAU_ERR_STATE --
 * the code rau_forward / rau_backward return instead of launching when a split count or slab
 * offset is stale.
 * rau_enc_ws_coresident: 1 if the weight-stationary persistent encoder (whose workgroups wait on each
 * other's progress counters) may be selected for `batch` samples on a device that admits
 * `blocks_per_cu` of its workgroups per CU on `n_cus` CUs, else 0. */
int rau_split_guard_check(size_t ws_floats, size_t offset, int nsplit, size_t per_split_floats);
int rau_enc_ws_coresident(int batch, int blocks_per_cu, int n_cus);

#ifdef __cplusplus
}
#endif
#endif /* RAU_H */
