/*
 * rau_oracle.h -- C interface of the CPU oracle for the RAU forward/backward path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference algorithm
 * (HyeonwooNoh/RAU_VQA, Lua/Torch7), used as the checker by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing in the
 * product path (rau_vqa_amd/, librau.so) may include, link or call it.
 *
 * PARITY UNPINNED: the reference is Lua/Torch7, has no tests, fixtures or golden
 * vectors, and no Lua toolchain exists in the build image, so this restatement
 * cannot be checked against the reference's own outputs.  It is cross-validated
 * against an independently written PyTorch-autograd restatement
 * (oracle/ref_torch.py) in fp64 instead (tests/test_oracle_agree.py).
 *
 * Reference files restated (file:line under /root/reference):
 *   experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua:198-316   network graph
 *   experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua:428-596   feval fwd/bwd
 *   model/ATTLSTM.lua:4-74    attention LSTM cell (gate order i, g, f, o)
 *   model/DeepLSTM.lua:14-71  2-layer question LSTM (gate order i, f, o | g)
 */
#ifndef RAU_ORACLE_H
#define RAU_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Shapes; names follow SURVEY.md section 8. */
typedef struct rau_oracle_cfg {
  int32_t B;   /* batch */
  int32_t T;   /* max question length (tokens array is [T,B]) */
  int32_t V;   /* vocabulary size; token ids are 1..V, 1 = ZEROPAD */
  int32_t E;   /* word embedding dim (200, SS:202) */
  int32_t Rq;  /* question LSTM size (512, SS:209); 2 layers fixed (SS:210) */
  int32_t D;   /* CNN feature channels (512 | 2048) */
  int32_t S;   /* spatial positions (14*14) */
  int32_t M;   /* multfeat_dim (512, SS:220) */
  int32_t A;   /* attfeat_dim (256, SS:221) */
  int32_t R;   /* att_rnn_size (512, SS:225) */
  int32_t K;   /* answer classes (1000, SS:222) */
  int32_t H;   /* hops (nHop) */
  float p_we, p_rnn, p_q, p_x, p_mf; /* dropout probabilities (0.5 each) */
} rau_oracle_cfg;

/* Flat parameter group sizes (floats).  Layout documented in DESIGN.md and
 * mirrored by include/rau.h: weight then bias per layer, in the order listed
 * in BASELINE.md section 2.3. */
size_t rau_oracle_n_embed(const rau_oracle_cfg* c);
size_t rau_oracle_n_rnn(const rau_oracle_cfg* c);
size_t rau_oracle_n_mult(const rau_oracle_cfg* c);

/* Inputs of one training step.  masks are uint8 keep-flags (1 = keep); a NULL
 * mask pointer means "no dropout at that site" (evaluate mode).
 *   feats  [B,D,S]   tokens [T,B] (1-based ids)   lens [B]   labels [B] (1-based)
 *   m_we [T,B,E]  m_rnn [T,B,Rq]  m_q [H,B,Q]  m_x [H,B,D,S]  m_mf [H,B,M]
 *   hop_w [H]   per-hop scale of dlogits (SS:569 nHop; MS: 1; Full: 0/1)
 * Outputs (any may be NULL):
 *   losses [H]  argmax [H,B] (1-based, first max)  logits [H,B,K]  dopred [H,B]
 *   att [H,B,S]  q [B,Q]  att_c, att_h [H,B,R]
 *   g_embed/g_rnn/g_mult: gradients, ACCUMULATED (+=) like nn accGradParameters;
 *   if all three are NULL only the forward pass runs.
 */
#define RAU_ORACLE_DECL(SUF, REAL)                                              \
  int rau_oracle_step_##SUF(                                                    \
      const rau_oracle_cfg* cfg, const REAL* embed, const REAL* rnn,            \
      const REAL* mult, const REAL* feats, const int32_t* tokens,               \
      const int32_t* lens, const int32_t* labels, const uint8_t* m_we,          \
      const uint8_t* m_rnn, const uint8_t* m_q, const uint8_t* m_x,             \
      const uint8_t* m_mf, const REAL* hop_w, REAL* losses, int32_t* argmax,    \
      REAL* logits, REAL* dopred, REAL* att, REAL* q, REAL* att_c, REAL* att_h, \
      REAL* g_embed, REAL* g_rnn, REAL* g_mult);

RAU_ORACLE_DECL(f32, float)
RAU_ORACLE_DECL(f64, double)

/* Counter-based dropout mask: the same Philox4x32-10 stream the HIP path uses
 * (rau_vqa_amd/csrc/philox.h).  Fills keep[n] with 0/1 for flat element
 * indices 0..n-1 of mask site `site` (0=we 1=rnn 2=q 3=x 4=mf) at training
 * step `step`. */
void rau_oracle_fill_mask(uint64_t seed, uint32_t site, uint32_t step, float p,
                          size_t n, uint8_t* keep);

/* Gradient-noise + clip + Adam (SS:597-630, utils/optim_updates.lua:59-87),
 * one flat group.  noise[n] is supplied by the caller (N(0,1) samples, scaled
 * inside by sqrt(eta/((step_t+1)*gamma))); may be NULL for no noise. */
#define RAU_ORACLE_UPD_DECL(SUF, REAL)                                          \
  void rau_oracle_noise_clip_adam_##SUF(                                        \
      size_t n, REAL* x, REAL* g, REAL* m, REAL* v, const REAL* noise,          \
      int64_t step_t, int64_t adam_t, REAL lr, REAL beta1, REAL beta2,          \
      REAL eps, REAL eta, REAL gamma, REAL clip, REAL* out_norm);
RAU_ORACLE_UPD_DECL(f32, float)
RAU_ORACLE_UPD_DECL(f64, double)

#ifdef __cplusplus
}
#endif
#endif
