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
int rau_batch_slot(rau_ctx* ctx, int slot, float** feats_host, int32_t** tokens_host,
                   int32_t** lens_host, int32_t** labels_host);
int rau_set_batch_async(rau_ctx* ctx, int slot, const float* feats, const int32_t* tokens,
                        const int32_t* lens, const int32_t* labels, int has_labels);
int rau_use_batch(rau_ctx* ctx, int slot);
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
int rau_dev_alloc(rau_ctx* ctx, size_t n_floats, float** out);
int rau_dev_free(rau_ctx* ctx, float* p);
int rau_dev_fill(rau_ctx* ctx, float* dst, size_t n, float value);
int rau_dev_copy(rau_ctx* ctx, float* dst, const float* src, size_t n);
int rau_dev_axpy(rau_ctx* ctx, float* y, const float* x, size_t n, float alpha);
int rau_dev_scale(rau_ctx* ctx, float* x, size_t n, float alpha);
int rau_dev_addcmul(rau_ctx* ctx, float* y, float alpha, const float* x1, const float* x2, size_t n);
int rau_dev_addcdiv(rau_ctx* ctx, float* y, float alpha, const float* x1, const float* x2, size_t n);
int rau_dev_sqrt(rau_ctx* ctx, float* x, size_t n);
int rau_dev_add_scalar(rau_ctx* ctx, float* x, size_t n, float value);
int rau_dev_adam(rau_ctx* ctx, float* x, const float* dx, float* m, float* v, size_t n, float lr,
                 float beta1, float beta2, float eps, int32_t t);
int rau_dev_select_rows(rau_ctx* ctx, float* dst, const float* src, int32_t rows, int32_t cols,
                        const int32_t* key_dev, int32_t value);
int rau_dev_rowmax(rau_ctx* ctx, const float* x, int32_t rows, int32_t cols, float* max_dev,
                   int32_t* argmax_dev);
int rau_dev_sum(rau_ctx* ctx, const float* x, size_t n, double* out_host);
int rau_dev_count_eq(rau_ctx* ctx, const int32_t* a_dev, const int32_t* b_dev, int32_t n,
                     int32_t* count_host);
int rau_dev_upload(rau_ctx* ctx, void* dst_dev, const void* host, size_t bytes);
int rau_dev_download(rau_ctx* ctx, void* host, const void* src_dev, size_t bytes);
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
int rau_prof_enable(rau_ctx* ctx, int on);
int rau_prof_reset(rau_ctx* ctx);
int rau_prof_count(rau_ctx* ctx);
int rau_prof_entry(rau_ctx* ctx, int index, const char** name, int64_t* launches,
                   double* total_ms, double* flops, double* bytes);
int rau_split_guard_check(size_t ws_floats, size_t offset, int nsplit, size_t per_split_floats);
int rau_enc_ws_coresident(int batch, int blocks_per_cu, int n_cus);
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
  -- `life.alive` is what device-tensor finalizers look at: the collector may run them after the
  -- context's own finalizer, and rau_destroy has then freed their memory already
  local life = { alive = true }
  local self = setmetatable({ h = ffi.gc(h[0], function(p) life.alive = false; C.rau_destroy(p) end),
                              cfg = cfg[0], life = life, scratch = {} }, RAU)
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

-- Asynchronous, double-buffered upload (slot = 0 | 1): what SS:434-439 does every iteration, moved
-- behind the loader's prefetch.  rau:batchSlot(slot) returns host tensors OVER the slot's pinned
-- staging (torch storages on foreign memory: no copy, not owned) for next_batch_feat's worker to
-- assemble the batch in; rau:setBatchAsync(slot) enqueues the upload of what is in them (pass
-- tensors to have them copied into the staging first); rau:useBatch(slot) makes it the resident
-- batch -- the step's streams wait for the copies by an event, the host never does.
function RAU:batchSlot(slot)
  local f, x, l, y = ffi.new('float*[1]'), ffi.new('int32_t*[1]'), ffi.new('int32_t*[1]'), ffi.new('int32_t*[1]')
  check(C.rau_batch_slot(self.h, slot, f, x, l, y))
  local c = self.cfg
  local function addr(p) return tonumber(ffi.cast('intptr_t', p)) end
  return {
    feats = torch.FloatTensor(torch.FloatStorage(c.B * c.D * c.S, addr(f[0]))):resize(c.B, c.D, c.S),
    x = torch.IntTensor(torch.IntStorage(c.T * c.B, addr(x[0]))):resize(c.T, c.B),
    x_len = torch.IntTensor(torch.IntStorage(c.B, addr(l[0]))),
    y = torch.IntTensor(torch.IntStorage(c.B, addr(y[0]))),
  }
end
function RAU:setBatchAsync(slot, feats, x, x_len, y, has_labels)
  check(C.rau_set_batch_async(self.h, slot, feats and feats:data() or nil, x and x:data() or nil,
                              x_len and x_len:data() or nil, y and y:data() or nil,
                              (has_labels == false) and 0 or 1))
end
function RAU:useBatch(slot) check(C.rau_use_batch(self.h, slot)) end

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

-- Device tensors ---------------------------------------------------------------
-- What feval's loops do BETWEEN module calls (`rnn_out[k] = lst[k]`, `uni_pred:add(pred[1])`,
-- `torch.max(pred[1], 2)`, `ans:eq(y):sum()`, SS:455-461, 482-492, 522-526, 584-591) needs
-- tensor-shaped values.  cutorch does not exist on an MI355X host, so the clones below return
-- RAU.Tensor objects: {ptr = device float*, size = {rows, cols}, rau = owner}, dense row-major,
-- with the handful of methods those loops use, each one small C-ABI call (rau_dev_*).
-- Views of ctx-owned slots keep nn.Module's self.output lifetime; RAU.Tensor.new allocates.
local Tensor = {}
local IntTensor = {}
local function numel(sz) local n = 1; for _, v in ipairs(sz) do n = n * v end; return n end
-- Owned device memory is returned to the context when its tensor is collected (or by :free());
-- the finalizer sits on the pointer cdata, so views (which copy the address, not the cdata) never
-- free anything.
local function dev_alloc(rau, n)
  local p = ffi.new('float*[1]')
  check(C.rau_dev_alloc(rau.h, n, p))
  local h, life = rau.h, rau.life
  return ffi.gc(p[0], function(q) if life.alive then C.rau_dev_free(h, q) end end)
end
local function dev_release(t)
  if t.owned then
    local q = ffi.gc(t.ptr, nil)
    t.owned = false
    if t.rau.life.alive then check(C.rau_dev_free(t.rau.h, ffi.cast('float*', q))) end
  end
end
local function ptr_of(x) if type(x) == 'table' then return x.ptr end; return x end

Tensor.__index = function(t, k)
  if type(k) == 'number' then return Tensor.row(t, k) end     -- t[k]: 1-based row view
  return Tensor[k]
end
Tensor.__newindex = function(t, k, v)
  if type(k) == 'number' then Tensor.row(t, k):copy(v) else rawset(t, k, v) end   -- t[k] = row
end
function Tensor.wrap(rau, ptr, ...)
  return setmetatable({ rau = rau, ptr = ptr, size = { ... } }, Tensor)
end
function Tensor.new(rau, ...)                                  -- zero-filled, like torch.zeros
  local sz = { ... }
  return setmetatable({ rau = rau, ptr = dev_alloc(rau, numel(sz)), size = sz, owned = true }, Tensor)
end
function Tensor:nElement() return numel(self.size) end
function Tensor:dim() return #self.size end
function Tensor:row(k)                                         -- 1-based, view
  local cols = numel(self.size) / self.size[1]
  assert(k >= 1 and k <= self.size[1], 'row index out of range')
  local v = Tensor.wrap(self.rau, self.ptr + (k - 1) * cols, cols)
  -- pointer arithmetic on the cdata makes a NEW cdata without the finalizer: the view keeps a
  -- reference to its owner so that the owner (and the memory) outlives it
  rawset(v, 'base', rawget(self, 'base') or self)
  return v
end
function Tensor:zero() check(C.rau_dev_fill(self.rau.h, self.ptr, self:nElement(), 0)); return self end
function Tensor:fill(v) check(C.rau_dev_fill(self.rau.h, self.ptr, self:nElement(), v)); return self end
function Tensor:copy(src)                                      -- device tensor or host FloatTensor
  if getmetatable(src) == Tensor then
    assert(src:nElement() == self:nElement(), 'size mismatch')
    check(C.rau_dev_copy(self.rau.h, self.ptr, src.ptr, self:nElement()))
  else
    check(C.rau_dev_upload(self.rau.h, self.ptr, src:data(), self:nElement() * 4))
  end
  return self
end
function Tensor:clone() return Tensor.new(self.rau, unpack(self.size)):copy(self) end
function Tensor:add(a, x)                                      -- :add(x), :add(alpha, x) or :add(scalar)
  if x == nil and type(a) == 'number' then
    check(C.rau_dev_add_scalar(self.rau.h, self.ptr, self:nElement(), a)); return self
  end
  if x == nil then a, x = 1, a end
  assert(x:nElement() == self:nElement(), 'size mismatch')
  check(C.rau_dev_axpy(self.rau.h, self.ptr, x.ptr, self:nElement(), a)); return self
end
function Tensor:mul(a) check(C.rau_dev_scale(self.rau.h, self.ptr, self:nElement(), a)); return self end
-- what utils/optim_updates.lua's adam() does to its flat vectors (lines 76-86)
function Tensor:addcmul(a, x1, x2)                             -- self += a * x1 * x2
  check(C.rau_dev_addcmul(self.rau.h, self.ptr, a, x1.ptr, x2.ptr, self:nElement())); return self
end
function Tensor:addcdiv(a, x1, x2)                             -- self += a * x1 / x2
  check(C.rau_dev_addcdiv(self.rau.h, self.ptr, a, x1.ptr, x2.ptr, self:nElement())); return self
end
function Tensor:sqrt() check(C.rau_dev_sqrt(self.rau.h, self.ptr, self:nElement())); return self end
function Tensor:div(a) return self:mul(1 / a) end
function Tensor:sum()
  local o = ffi.new('double[1]')
  check(C.rau_dev_sum(self.rau.h, self.ptr, self:nElement(), o)); return o[0]
end
function Tensor:mean() return self:sum() / self:nElement() end
-- torch.max(t, 2): values [rows,1] and 1-based first-max indices [rows,1] (SS:488)
function Tensor:max(dim)
  assert(dim == 2 and #self.size == 2, 'only max over dimension 2 of a matrix')
  local r, c = self.size[1], self.size[2]
  -- results live in a ring of four context-owned slots per row count (feval calls this once per
  -- hop per iteration, SS:488: no allocation on that path); like self.output they stay valid
  -- until the slot comes round again
  local ring = self.rau.scratch[r]
  if not ring then
    ring = { k = 0 }
    for s = 1, 4 do ring[s] = { Tensor.new(self.rau, r, 1), IntTensor.new(self.rau, r, 1) } end
    self.rau.scratch[r] = ring
  end
  ring.k = ring.k % 4 + 1
  local v, i = ring[ring.k][1], ring[ring.k][2]
  check(C.rau_dev_rowmax(self.rau.h, self.ptr, r, c, v.ptr, i.ptr))
  return v, i
end
-- dst rows k with key[k] == value take src's rows: the whole `for k=1,B do if x_len[k]==t ...`
-- loop of SS:455-461 / SS:584-591 as one call (key: device IntTensor)
function Tensor:selectRows(src, key, value)
  local r = self.size[1]
  check(C.rau_dev_select_rows(self.rau.h, self.ptr, src.ptr, r, self:nElement() / r, key.ptr, value))
  return self
end
function Tensor:float()                                        -- host copy (torch.FloatTensor)
  local t = torch.FloatTensor(unpack(self.size))
  check(C.rau_dev_download(self.rau.h, t:data(), self.ptr, self:nElement() * 4)); return t
end
function Tensor:free() dev_release(self) end

IntTensor.__index = IntTensor
function IntTensor.new(rau, ...)
  local sz = { ... }
  local base = dev_alloc(rau, numel(sz))                       -- 4-byte elements either way
  -- `base` carries the finalizer; the int32 view of it is what the calls take
  return setmetatable({ rau = rau, base = base, ptr = ffi.cast('int32_t*', base), size = sz,
                        owned = true }, IntTensor)
end
function IntTensor:free()
  if self.owned then
    local t = { rau = self.rau, ptr = self.base, owned = true }
    self.owned = false
    dev_release(t)
  end
end
function IntTensor:copy(src)                                   -- host IntTensor -> device
  check(C.rau_dev_upload(self.rau.h, self.ptr, src:data(), numel(self.size) * 4)); return self
end
function IntTensor:int()
  local t = torch.IntTensor(unpack(self.size))
  check(C.rau_dev_download(self.rau.h, t:data(), self.ptr, numel(self.size) * 4)); return t
end
-- ans:eq(y):sum() in one call (SS:489-492)
function IntTensor:eqSum(other)
  local o = ffi.new('int32_t[1]')
  check(C.rau_dev_count_eq(self.rau.h, self.ptr, other.ptr, numel(self.size), o)); return o[0]
end
RAU.Tensor, RAU.IntTensor = Tensor, IntTensor
function RAU:zeros(...) return Tensor.new(self, ...) end       -- torch.zeros(...):cuda()
function RAU:ints(host) return IntTensor.new(self, host:nElement()):copy(host) end

-- nn.Module surface completeness: parameters live on the device from the start and the clones
-- share them by construction, so these are identities (SS:319, 340-346); updateParameters is
-- never called by the reference (updates go through adam on the flat vectors, SS:770-772); for
-- hosts that do call it, it is nn.Module's plain SGD step x = x - lr * dx on all three groups.
function RAU:cuda() return self end
function RAU:clone() return self end
function RAU:float() return self end
function RAU:updateParameters(lr)
  for _, g in ipairs({ 'embed', 'rnn', 'mult' }) do
    local p = self:getParameters(g)
    check(C.rau_dev_axpy(self.h, p.ptr, p.grad, p.n, -lr))
  end
end
-- flat parameter / gradient vectors as device tensors: params, grads = rau:flat('mult')
function RAU:flat(group)
  local p = self:getParameters(group)
  return Tensor.wrap(self, p.ptr, p.n), Tensor.wrap(self, p.grad, p.n)
end

-- adam(x, dx, lr, beta1, beta2, epsilon, state): the signature and state fields of
-- utils/optim_updates.lua:59-87, for x, dx = rau:flat(group) -- so SS:770-772 runs unchanged.
-- state.m / state.v are device tensors created on first use (x.new(#dx):zero() in the reference),
-- state.t counts calls; the five tensor statements run as one pass (rau_dev_adam).  No state.tmp:
-- sqrt(v) + epsilon never leaves registers.
function RAU.adam(x, dx, lr, beta1, beta2, epsilon, state)
  beta1, beta2, epsilon = beta1 or 0.9, beta2 or 0.999, epsilon or 1e-8
  if not state.m then
    state.t = 0
    state.m = Tensor.new(x.rau, x:nElement())
    state.v = Tensor.new(x.rau, x:nElement())
  end
  state.t = state.t + 1
  check(C.rau_dev_adam(x.rau.h, x.ptr, dx.ptr, state.m.ptr, state.v.ptr, x:nElement(), lr, beta1, beta2,
                       epsilon, state.t))
end

-- Module-level clones ---------------------------------------------------------
-- For scripts that keep feval's own loops (SS:443-596).  Arguments are RAU.Tensor / RAU.IntTensor
-- objects (or raw device cdata pointers); results are RAU.Tensor VIEWS of ctx-owned slots that stay
-- valid until the same clone runs again (the lifetime of nn.Module's self.output).
-- Indices are 1-based like embed_clones[t] / lstm_clones[t] / multimodal_clones[h].
local function clone(self, kind, i)
  local m = { rau = self, i = i - 1 }
  local cfg, Q = self.cfg, 4 * self.cfg.Rq
  if kind == 'embed' then
    function m:forward(x_t)
      local o = ffi.new('float*[1]')
      check(C.rau_embed_forward(self.rau.h, self.i, ptr_of(x_t), o))
      self.output = Tensor.wrap(self.rau, o[0], cfg.B, cfg.E)
      return self.output
    end
    function m:backward(x_t, d_we) check(C.rau_embed_backward(self.rau.h, self.i, ptr_of(x_t), ptr_of(d_we))) end
  elseif kind == 'rnn' then
    function m:forward(inp)   -- {x, state}
      local o = ffi.new('float*[1]')
      check(C.rau_deeplstm_forward(self.rau.h, self.i, ptr_of(inp[1]), ptr_of(inp[2]), o))
      self.output = Tensor.wrap(self.rau, o[0], cfg.B, Q)
      return self.output
    end
    function m:backward(inp, d_state_out)
      local dx, ds = ffi.new('float*[1]'), ffi.new('float*[1]')
      check(C.rau_deeplstm_backward(self.rau.h, self.i, ptr_of(inp[1]), ptr_of(inp[2]),
                                    ptr_of(d_state_out), dx, ds))
      self.gradInput = { Tensor.wrap(self.rau, dx[0], cfg.B, cfg.E), Tensor.wrap(self.rau, ds[0], cfg.B, Q) }
      return self.gradInput
    end
  elseif kind == 'multimodal' then
    function m:forward(inp)   -- {q, X, c, h}
      local o = {}
      for k = 1, 5 do o[k] = ffi.new('float*[1]') end
      check(C.rau_multimodal_forward(self.rau.h, self.i, ptr_of(inp[1]), ptr_of(inp[2]),
                                     ptr_of(inp[3]), ptr_of(inp[4]), o[1], o[2], o[3], o[4], o[5]))
      local r = self.rau                                             -- {logits, dp, a, c, h}
      self.output = { Tensor.wrap(r, o[1][0], cfg.B, cfg.K), Tensor.wrap(r, o[2][0], cfg.B),
                      Tensor.wrap(r, o[3][0], cfg.B, cfg.S), Tensor.wrap(r, o[4][0], cfg.B, cfg.R),
                      Tensor.wrap(r, o[5][0], cfg.B, cfg.R) }
      return self.output
    end
    function m:backward(inp, g)   -- g = {d_logits, d_do_pred|nil, d_attprob|nil, d_c, d_h}
      local o = {}
      for k = 1, 4 do o[k] = ffi.new('float*[1]') end
      check(C.rau_multimodal_backward(self.rau.h, self.i, ptr_of(inp[1]), ptr_of(inp[2]),
                                      ptr_of(inp[3]), ptr_of(inp[4]), ptr_of(g[1]), ptr_of(g[2]),
                                      ptr_of(g[3]), ptr_of(g[4]), ptr_of(g[5]), o[1], nil, o[3], o[4]))
      local r = self.rau                                    -- {d_q, (d_X dead, SS:579), d_c, d_h}
      self.gradInput = { Tensor.wrap(r, o[1][0], cfg.B, Q), nil, Tensor.wrap(r, o[3][0], cfg.B, cfg.R),
                         Tensor.wrap(r, o[4][0], cfg.B, cfg.R) }
      return self.gradInput
    end
  elseif kind == 'criterion' then
    function m:forward(logits, y)
      local l = ffi.new('float[1]')
      check(C.rau_criterion_forward(self.rau.h, self.i, ptr_of(logits), ptr_of(y), l))
      return l[0]
    end
    function m:backward(logits, y, scale)
      local o = ffi.new('float*[1]')
      check(C.rau_criterion_backward(self.rau.h, self.i, ptr_of(logits), ptr_of(y), scale or 1, o))
      return Tensor.wrap(self.rau, o[0], cfg.B, cfg.K)
    end
  end
  function m:training() self.rau:training() end
  function m:evaluate() self.rau:evaluate() end
  function m:cuda() return self end
  function m:clone() return self end                 -- clones share parameters by construction
  function m:getParameters() return self.rau:flat(kind == 'embed' and 'embed' or kind == 'rnn' and 'rnn' or 'mult') end
  return m
end
function RAU:embedClone(t) return clone(self, 'embed', t) end
function RAU:rnnClone(t) return clone(self, 'rnn', t) end
function RAU:multimodalClone(h) return clone(self, 'multimodal', h) end
function RAU:criterion(h) return clone(self, 'criterion', h) end

return RAU
