--[[ rau.lua -- LuaJIT FFI shim over librau.so (include/rau.h).

Presents the reference's nn.Module call surface for the RAU hot path so that
experiments/Ours_*/LstmAttCtrlGradNoiseDontSelect.lua keeps its structure:
  :training() / :evaluate()          (SS:449-450, 479, 648-649, 676)
  :getParameters()                   (SS:322-324)  -> flat param / grad handles
  feval's tensor half                (SS:428-596)  -> rau:forward() / rau:backward(w)
  adam(x, dx, lr, ...) x 3 + noise + clip (SS:597-630, 770-772) -> rau:update(...)

No LuaJIT/Torch7 toolchain exists in the build image, so this file is shipped as
source and is NOT exercised by the tests; the Python ctypes binding
(rau_vqa_amd/_lib.py) is the executable proof of the identical ABI.
INTEGRATION.md shows the reference-side patch that uses this file.
]]
local ffi = require 'ffi'

ffi.cdef[[
typedef struct rau_config {
  int32_t B, T, V, E, Rq, D, S, M, A, R, K, H;
  float p_we, p_rnn, p_q, p_x, p_mf;
  int32_t dtype, device_id;
} rau_config;
typedef struct rau_ctx rau_ctx;
void rau_default_config(rau_config* cfg);
const char* rau_last_error(void);
int rau_abi_version(void);
int rau_create(const rau_config* cfg, rau_ctx** out);
void rau_destroy(rau_ctx* ctx);
int rau_params(rau_ctx* ctx, int group, float** weights, float** grads, size_t* n);
int rau_layout_count(const rau_ctx* ctx, int group);
int rau_layout_entry(const rau_ctx* ctx, int group, int index, const char** name,
                     size_t* offset, int32_t* rows, int32_t* cols);
int rau_set_params(rau_ctx* ctx, int group, const float* host, size_t n);
int rau_get_params(rau_ctx* ctx, int group, float* host, size_t n);
int rau_get_grads(rau_ctx* ctx, int group, float* host, size_t n);
int rau_set_grads(rau_ctx* ctx, int group, const float* host, size_t n);
int rau_init_uniform(rau_ctx* ctx, uint64_t seed, float lo, float hi);
int rau_zero_grads(rau_ctx* ctx);
int rau_set_mode(rau_ctx* ctx, int mode);
int rau_set_dropout_seed(rau_ctx* ctx, uint64_t seed, uint32_t step);
int rau_set_mask(rau_ctx* ctx, int site, const uint8_t* keep, size_t n);
int rau_get_mask(rau_ctx* ctx, int site, uint8_t* keep, size_t n);
int rau_set_batch(rau_ctx* ctx, const float* feats, const int32_t* tokens,
                  const int32_t* lens, const int32_t* labels);
int rau_batch_feats(rau_ctx* ctx, float** feats_dev);
int rau_forward(rau_ctx* ctx);
int rau_backward(rau_ctx* ctx, const float* hop_w);
int rau_embed_forward(rau_ctx* ctx, int t, const int32_t* tokens_dev, float** we);
int rau_embed_backward(rau_ctx* ctx, int t, const int32_t* tokens_dev, const float* d_we);
int rau_deeplstm_forward(rau_ctx* ctx, int t, const float* x, const float* state,
                         float** state_out);
int rau_deeplstm_backward(rau_ctx* ctx, int t, const float* x, const float* state,
                          const float* d_state_out, float** d_x, float** d_state);
int rau_multimodal_forward(rau_ctx* ctx, int h, const float* q, const float* X,
                           const float* c_prev, const float* h_prev, float** logits,
                           float** do_pred, float** attprob, float** c_out, float** h_out);
int rau_multimodal_backward(rau_ctx* ctx, int h, const float* q, const float* X,
                            const float* c_prev, const float* h_prev, const float* d_logits,
                            const float* d_do_pred, const float* d_attprob,
                            const float* d_c, const float* d_h, float** d_q, float** d_X,
                            float** d_c_prev, float** d_h_prev);
int rau_criterion_forward(rau_ctx* ctx, int h, const float* logits, const int32_t* labels_dev,
                          float* loss);
int rau_criterion_backward(rau_ctx* ctx, int h, const float* logits,
                           const int32_t* labels_dev, float scale, float** d_logits);
int rau_graph_step(rau_ctx* ctx, const float* hop_w, int zero_grads_first);
int rau_sync(rau_ctx* ctx);
int rau_get_losses(rau_ctx* ctx, float* losses);
int rau_get_argmax(rau_ctx* ctx, int32_t* ans);
int rau_get_logits(rau_ctx* ctx, float* logits);
int rau_get_dopred(rau_ctx* ctx, float* dopred);
int rau_get_attention(rau_ctx* ctx, float* att);
int rau_get_question_state(rau_ctx* ctx, float* q);
int rau_get_att_state(rau_ctx* ctx, float* c, float* h);
int rau_noise_clip_adam(rau_ctx* ctx, int64_t step_t, float lr, float mult_lr,
                        float beta1, float beta2, float eps, float eta, float gamma,
                        float clip, uint64_t noise_seed, float* out_norms);
int rau_stream(rau_ctx* ctx, void** hip_stream);
int rau_wait_grads(rau_ctx* ctx, int group, void* hip_stream);
int rau_comm_unique_id(void* id, size_t bytes);
int rau_comm_init(rau_ctx* ctx, int nranks, int rank, const void* id, size_t bytes);
int rau_allreduce_grads(rau_ctx* ctx);
int rau_comm_destroy(rau_ctx* ctx);
int rau_timer_begin(rau_ctx* ctx);
int rau_timer_end(rau_ctx* ctx, float* ms);
]]

local C = ffi.load(os.getenv('RAU_LIB') or 'librau.so')
local GROUP = { embed = 0, rnn = 1, mult = 2 }

local function check(rc)
  if rc ~= 0 then error('librau: ' .. ffi.string(C.rau_last_error()), 3) end
end

local RAU = {}
RAU.__index = RAU

-- opt: the reference's hard-coded locals (SS:202-229) plus data-defined sizes
function RAU.new(opt)
  local cfg = ffi.new('rau_config[1]')
  C.rau_default_config(cfg)
  for k, v in pairs(opt) do cfg[0][k] = v end
  local h = ffi.new('rau_ctx*[1]')
  check(C.rau_create(cfg, h))
  local self = setmetatable({ h = ffi.gc(h[0], C.rau_destroy), cfg = cfg[0] }, RAU)
  return self
end

-- nn.Module surface ----------------------------------------------------------
function RAU:training() check(C.rau_set_mode(self.h, 0)); return self end
function RAU:evaluate() check(C.rau_set_mode(self.h, 1)); return self end

-- m:getParameters(): returns {ptr=device float*, grad=device float*, n=count}
function RAU:getParameters(group)
  local w, g, n = ffi.new('float*[1]'), ffi.new('float*[1]'), ffi.new('size_t[1]')
  check(C.rau_params(self.h, GROUP[group], w, g, n))
  return { ptr = w[0], grad = g[0], n = tonumber(n[0]) }
end

-- param:uniform(-0.08, 0.08), SS:352-354
function RAU:reset(seed, lo, hi) check(C.rau_init_uniform(self.h, seed or 123, lo or -0.08, hi or 0.08)) end

-- feats: FloatTensor [B,D,W,H]; x: IntTensor [T,B]; x_len, y: IntTensor [B]  (loader.lua:1009)
function RAU:setBatch(feats, x, x_len, y)
  check(C.rau_set_batch(self.h, feats:data(), x:data(), x_len:data(), y and y:data() or nil))
end

-- forward half of feval (SS:443-520); returns per-hop losses as a Lua table
function RAU:forward(seed, step)
  if seed then check(C.rau_set_dropout_seed(self.h, seed, step or 0)) end
  check(C.rau_forward(self.h))
  local H = self.cfg.H
  local l = ffi.new('float[?]', H)
  check(C.rau_get_losses(self.h, l))
  local t = {}
  for i = 1, H do t[i] = l[i - 1] end
  return t
end

-- backward half (SS:561-596); hop_w = per-hop criterion-gradient scale (SS:569 / Full:587-589)
function RAU:backward(hop_w)
  local H = self.cfg.H
  local w = ffi.new('float[?]', H)
  for i = 1, H do w[i - 1] = hop_w[i] end
  check(C.rau_backward(self.h, w))
end

function RAU:zeroGradParameters() check(C.rau_zero_grads(self.h)) end

-- noise + clip + adam x 3 groups (SS:597-630, 770-772)
function RAU:update(step_t, lr, mult_lr, eta, gamma, clip, seed)
  local norms = ffi.new('float[3]')
  check(C.rau_noise_clip_adam(self.h, step_t, lr, mult_lr, 0.9, 0.999, 1e-8, eta or 0.01,
                              gamma or 0.55, clip or 0.1, seed or 0, norms))
  return norms[0], norms[1], norms[2]
end

-- answers: IntTensor [H,B] of 1-based class ids (torch.max first-max rule, SS:488)
function RAU:answers(out) check(C.rau_get_argmax(self.h, out:data())); return out end
function RAU:logits(out) check(C.rau_get_logits(self.h, out:data())); return out end
function RAU:attention(out) check(C.rau_get_attention(self.h, out:data())); return out end
function RAU:sync() check(C.rau_sync(self.h)) end

-- data parallel (one process per GPU): rank 0 calls RAU.commId() and ships the 128-byte string
-- to the other ranks (file, socket, ...); every rank then calls rau:commInit(n, rank, id) once
-- and rau:allreduceGrads() between rau:backward(w) and rau:update(...)
function RAU.commId()
  local id = ffi.new('char[128]')
  check(C.rau_comm_unique_id(id, 128))
  return ffi.string(id, 128)
end
function RAU:commInit(nranks, rank, id) check(C.rau_comm_init(self.h, nranks, rank, id, #id)) end
function RAU:allreduceGrads() check(C.rau_allreduce_grads(self.h)) end

-- Module-level clones ---------------------------------------------------------
-- For scripts that keep feval's own loops (SS:443-596).  Arguments and results are
-- device float* / int32_t* (cdata); a result points into a ctx-owned slot that stays
-- valid until the same clone runs again (the lifetime of nn.Module's self.output).
-- Indices are 1-based like embed_clones[t] / lstm_clones[t] / multimodal_clones[h].
local function clone(self, kind, i)
  local m = { rau = self, i = i - 1 }
  if kind == 'embed' then
    function m:forward(x_t)
      local o = ffi.new('float*[1]')
      check(C.rau_embed_forward(self.rau.h, self.i, x_t, o)); self.output = o[0]
      return self.output
    end
    function m:backward(x_t, d_we) check(C.rau_embed_backward(self.rau.h, self.i, x_t, d_we)) end
  elseif kind == 'rnn' then
    function m:forward(inp)   -- {x, state}
      local o = ffi.new('float*[1]')
      check(C.rau_deeplstm_forward(self.rau.h, self.i, inp[1], inp[2], o)); self.output = o[0]
      return self.output
    end
    function m:backward(inp, d_state_out)
      local dx, ds = ffi.new('float*[1]'), ffi.new('float*[1]')
      check(C.rau_deeplstm_backward(self.rau.h, self.i, inp[1], inp[2], d_state_out, dx, ds))
      self.gradInput = { dx[0], ds[0] }
      return self.gradInput
    end
  elseif kind == 'multimodal' then
    function m:forward(inp)   -- {q, X, c, h}
      local o = {}
      for k = 1, 5 do o[k] = ffi.new('float*[1]') end
      check(C.rau_multimodal_forward(self.rau.h, self.i, inp[1], inp[2], inp[3], inp[4],
                                     o[1], o[2], o[3], o[4], o[5]))
      self.output = { o[1][0], o[2][0], o[3][0], o[4][0], o[5][0] }  -- {logits, dp, a, c, h}
      return self.output
    end
    function m:backward(inp, g)   -- g = {d_logits, d_do_pred|nil, d_attprob|nil, d_c, d_h}
      local o = {}
      for k = 1, 4 do o[k] = ffi.new('float*[1]') end
      check(C.rau_multimodal_backward(self.rau.h, self.i, inp[1], inp[2], inp[3], inp[4],
                                      g[1], g[2], g[3], g[4], g[5], o[1], nil, o[3], o[4]))
      self.gradInput = { o[1][0], nil, o[3][0], o[4][0] }   -- {d_q, (d_X dead, SS:579), d_c, d_h}
      return self.gradInput
    end
  elseif kind == 'criterion' then
    function m:forward(logits, y)
      local l = ffi.new('float[1]')
      check(C.rau_criterion_forward(self.rau.h, self.i, logits, y, l))
      return l[0]
    end
    function m:backward(logits, y, scale)
      local o = ffi.new('float*[1]')
      check(C.rau_criterion_backward(self.rau.h, self.i, logits, y, scale or 1, o))
      return o[0]
    end
  end
  function m:training() self.rau:training() end
  function m:evaluate() self.rau:evaluate() end
  return m
end
function RAU:embedClone(t) return clone(self, 'embed', t) end
function RAU:rnnClone(t) return clone(self, 'rnn', t) end
function RAU:multimodalClone(h) return clone(self, 'multimodal', h) end
function RAU:criterion(h) return clone(self, 'criterion', h) end

return RAU
