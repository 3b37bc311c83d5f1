"""nn.Module-shaped clone objects over librau's module-level entry points.

The reference's feval (experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua
:443-596, "SS") drives three families of weight-sharing clones one call at a
time: ``embed_clones[t]``, ``lstm_clones[t]``, ``multimodal_clones[h]`` plus
``criteria[h]``.  These classes present exactly that surface (``forward`` /
``backward`` with the reference's table orders) on top of ``rau_*_forward`` /
``rau_*_backward`` in include/rau.h; tensors are torch CUDA tensors used purely as
device memory (their ``data_ptr()`` crosses the C ABI), and outputs are zero-copy
views of the ctx-owned slots -- valid until the same clone runs again, like
``self.output`` of an nn.Module.

``feval`` re-states the reference's own loops over these clones; the step-level
``RAU.forward/backward`` computes the same numbers with cross-clone batching.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .dist import device_view


def _p(t):
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous()):
        raise ValueError("module-level arguments must be contiguous CUDA tensors")
    return C.c_void_p(t.data_ptr())


class _Clone:
    def __init__(self, rau, index: int):
        self.rau, self.i = rau, index
        self._lib, self._h = rau._lib, rau._h
        self.dev = rau.cfg.device_id

    def _view(self, ptr, *shape):
        n = 1
        for s in shape:
            n *= s
        return device_view(ptr.value, n, self.dev).view(*shape)


class EmbedClone(_Clone):
    """embed_clones[t]: LookupTable -> Dropout -> Tanh (SS:203-206)."""

    def forward(self, x_t):
        out = C.c_void_p()
        L.check(self._lib.rau_embed_forward(self._h, self.i, _p(x_t), C.byref(out)))
        return self._view(out, self.rau.cfg.B, self.rau.cfg.E)

    def backward(self, x_t, d_we):
        L.check(self._lib.rau_embed_backward(self._h, self.i, _p(x_t), _p(d_we)))


class DeepLSTMClone(_Clone):
    """lstm_clones[t]: {x, state} -> state' (model/DeepLSTM.lua:14-71)."""

    def forward(self, x, state):
        out = C.c_void_p()
        L.check(self._lib.rau_deeplstm_forward(self._h, self.i, _p(x), _p(state), C.byref(out)))
        return self._view(out, self.rau.cfg.B, self.rau.cfg.Q)

    def backward(self, x, state, d_state_out):
        dx, ds = C.c_void_p(), C.c_void_p()
        L.check(self._lib.rau_deeplstm_backward(self._h, self.i, _p(x), _p(state),
                                                _p(d_state_out), C.byref(dx), C.byref(ds)))
        c = self.rau.cfg
        return self._view(dx, c.B, c.E), self._view(ds, c.B, c.Q)


class MultimodalClone(_Clone):
    """multimodal_clones[h]: {q, X, c, h} -> {logits, do_pred, attprob, c', h'} (SS:292-307)."""

    def forward(self, q, X, c_prev, h_prev):
        outs = [C.c_void_p() for _ in range(5)]
        L.check(self._lib.rau_multimodal_forward(self._h, self.i, _p(q), _p(X), _p(c_prev),
                                                 _p(h_prev), *[C.byref(o) for o in outs]))
        c = self.rau.cfg
        return (self._view(outs[0], c.B, c.K), self._view(outs[1], c.B),
                self._view(outs[2], c.B, c.S), self._view(outs[3], c.B, c.R),
                self._view(outs[4], c.B, c.R))

    def backward(self, q, X, c_prev, h_prev, d_logits, d_do_pred=None, d_attprob=None, d_c=None,
                 d_h=None, want_dX=False):
        outs = [C.c_void_p() for _ in range(4)]
        refs = [C.byref(o) for o in outs]
        if not want_dX:
            refs[1] = None
        L.check(self._lib.rau_multimodal_backward(
            self._h, self.i, _p(q), _p(X), _p(c_prev), _p(h_prev), _p(d_logits), _p(d_do_pred),
            _p(d_attprob), _p(d_c), _p(d_h), *refs))
        c = self.rau.cfg
        dX = self._view(outs[1], c.B, c.D, c.S) if want_dX else None
        return (self._view(outs[0], c.B, c.Q), dX, self._view(outs[2], c.B, c.R),
                self._view(outs[3], c.B, c.R))


class CriterionClone(_Clone):
    """criteria[h]: nn.CrossEntropyCriterion with sizeAverage (SS:310)."""

    def forward(self, logits, y):
        loss = C.c_float()
        L.check(self._lib.rau_criterion_forward(self._h, self.i, _p(logits), _p(y),
                                                C.byref(loss)))
        return loss.value

    def backward(self, logits, y, scale=1.0):
        out = C.c_void_p()
        L.check(self._lib.rau_criterion_backward(self._h, self.i, _p(logits), _p(y), float(scale),
                                                 C.byref(out)))
        return self._view(out, self.rau.cfg.B, self.rau.cfg.K)


def feval(rau, feats, x, x_len, y, hop_w):
    """The tensor half of the reference's feval, loop for loop (SS:443-596), on the clones.

    feats [B,D,S] float32, x [T,B] int32, x_len [B] int32, y [B] int32: CUDA tensors.
    Gradients accumulate into the ctx's flat buffers (zero them first).  Returns
    (losses[H], argmax[H,B] as torch tensors of 1-based ids).
    """
    ext = torch.cuda.ExternalStream(rau.stream(), device=feats.device)
    with torch.cuda.stream(ext):   # torch's glue ops join the ctx's own stream order
        return _feval(rau, feats, x, x_len, y, hop_w)


def _feval(rau, feats, x, x_len, y, hop_w):
    c = rau.cfg
    dev = feats.device
    emb = [EmbedClone(rau, t) for t in range(c.T)]
    rnn = [DeepLSTMClone(rau, t) for t in range(c.T)]
    mm = [MultimodalClone(rau, h) for h in range(c.H)]
    crit = [CriterionClone(rau, h) for h in range(c.H)]
    max_len = int(x_len.max().item())                      # SS:444
    # ---- encoder forward, SS:446-462
    state = [torch.zeros(c.B, c.Q, device=dev)]            # init_state, SS:358
    we = []
    rnn_out = torch.zeros(c.B, c.Q, device=dev)
    for t in range(max_len):
        we.append(emb[t].forward(x[t]))
        state.append(rnn[t].forward(we[t], state[t]))
        sel = (x_len == t + 1).unsqueeze(1)                # SS:455-461
        rnn_out = torch.where(sel, state[t + 1], rnn_out)
    # ---- hops forward, SS:467-520
    att_c = [torch.zeros(c.B, c.R, device=dev)]            # SS:362-365
    att_h = [torch.zeros(c.B, c.R, device=dev)]
    logits, losses, answers = [], [], []
    for h in range(c.H):
        lg, _dp, _a, cn, hn = mm[h].forward(rnn_out, feats, att_c[h], att_h[h])
        logits.append(lg)
        att_c.append(cn)
        att_h.append(hn)
        losses.append(crit[h].forward(lg, y))              # SS:518
        answers.append(torch.argmax(lg, dim=1) + 1)        # SS:488 (ties: see rau_get_argmax)
    # ---- hops backward, SS:561-579
    d_c = d_h = None                                       # zeros, SS:561-562
    d_q = torch.zeros(c.B, c.Q, device=dev)
    for h in reversed(range(c.H)):
        dl = crit[h].backward(logits[h], y, float(hop_w[h]))   # SS:565-569
        dq_h, _dX, d_c, d_h = mm[h].backward(rnn_out, feats, att_c[h], att_h[h], dl, None, None,
                                              d_c, d_h)
        d_q += dq_h                                        # ConcatTable backward, SS:579
    # ---- encoder backward, SS:581-596
    d_state = torch.zeros(c.B, c.Q, device=dev)
    for t in reversed(range(max_len)):
        sel = (x_len == t + 1).unsqueeze(1)                # rows REPLACED by dq, SS:584-591
        d_out = torch.where(sel, d_q, d_state)
        d_x, d_state = rnn[t].backward(we[t], state[t], d_out)
        emb[t].backward(x[t], d_x)
    return torch.tensor(losses), torch.stack(answers)


# --------------------------------------------------------------------------------------------
# Device tensors without torch: the Python twin of bindings/rau.lua's RAU.Tensor.  Every method is
# one rau_dev_* call of the C ABI, so a host with no tensor library on the GPU (LuaJIT + CPU
# Torch7 on an MI355X box) can keep feval's own loops.  feval_dev below is those loops written
# with nothing else -- the executable proof of what the Lua shim does.
class DevTensor:
    """Dense row-major float32 device tensor: {ptr, size, rau}; views share memory."""

    def __init__(self, rau, ptr: int, size, owned=False, base=None):
        self.rau, self.ptr, self.size, self.owned = rau, int(ptr), tuple(size), owned
        # a view keeps its owner alive: the owner's finalizer frees the memory the view points into
        self._base = base

    def __del__(self):
        # owned memory goes back to the context when the tensor is collected (views own nothing);
        # a context that is already closed has freed it itself
        try:
            if self.owned and getattr(self.rau, "_h", None):
                self.rau._lib.rau_dev_free(self.rau._h, self.ptr)
        except Exception:
            pass
        self.owned = False

    @classmethod
    def zeros(cls, rau, *size):
        out = C.c_void_p()
        n = 1
        for v in size:
            n *= v
        L.check(rau._lib.rau_dev_alloc(rau._h, n, C.byref(out)))
        return cls(rau, out.value, size, owned=True)

    @classmethod
    def wrap(cls, rau, ptr, *size):
        return cls(rau, ptr.value if isinstance(ptr, C.c_void_p) else ptr, size)

    def numel(self):
        n = 1
        for v in self.size:
            n *= v
        return n

    def row(self, k):                 # 0-based here (Lua: 1-based), view
        cols = self.numel() // self.size[0]
        return DevTensor(self.rau, self.ptr + 4 * k * cols, (cols,), base=self._base or self)

    def __getitem__(self, k):
        return self.row(k)

    def __setitem__(self, k, v):      # t[k] = row  (rnn_out[k] = lst[k], SS:455-461)
        self.row(k).copy(v)

    def zero(self):
        return self.fill(0.0)

    def fill(self, v):
        L.check(self.rau._lib.rau_dev_fill(self.rau._h, self.ptr, self.numel(), float(v)))
        return self

    def copy(self, src):
        if isinstance(src, DevTensor):
            assert src.numel() == self.numel()
            L.check(self.rau._lib.rau_dev_copy(self.rau._h, self.ptr, src.ptr, self.numel()))
        else:                          # host array
            import numpy as np
            a = np.ascontiguousarray(src, np.float32)
            assert a.size == self.numel()
            L.check(self.rau._lib.rau_dev_upload(self.rau._h, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def add(self, a, x=None):          # :add(x) or :add(alpha, x)
        if x is None:
            a, x = 1.0, a
        assert x.numel() == self.numel()
        L.check(self.rau._lib.rau_dev_axpy(self.rau._h, self.ptr, x.ptr, self.numel(), float(a)))
        return self

    def mul(self, a):
        L.check(self.rau._lib.rau_dev_scale(self.rau._h, self.ptr, self.numel(), float(a)))
        return self

    # the tensor statements of utils/optim_updates.lua:76-86
    def add_scalar(self, v):
        L.check(self.rau._lib.rau_dev_add_scalar(self.rau._h, self.ptr, self.numel(), float(v)))
        return self

    def addcmul(self, a, x1, x2):      # self += a * x1 * x2
        L.check(self.rau._lib.rau_dev_addcmul(self.rau._h, self.ptr, float(a), x1.ptr, x2.ptr,
                                              self.numel()))
        return self

    def addcdiv(self, a, x1, x2):      # self += a * x1 / x2
        L.check(self.rau._lib.rau_dev_addcdiv(self.rau._h, self.ptr, float(a), x1.ptr, x2.ptr,
                                              self.numel()))
        return self

    def sqrt(self):
        L.check(self.rau._lib.rau_dev_sqrt(self.rau._h, self.ptr, self.numel()))
        return self

    def sum(self):
        out = C.c_double()
        L.check(self.rau._lib.rau_dev_sum(self.rau._h, self.ptr, self.numel(), C.byref(out)))
        return out.value

    def max(self, dim):                # torch.max(t, 2): (values [r,1], 1-based first-max ids [r,1])
        assert dim == 2 and len(self.size) == 2
        r, c = self.size
        # a ring of four context-owned result slots per row count: feval calls this once per hop
        # per iteration (SS:488), so nothing is allocated on that path; like self.output a result
        # stays valid until its slot comes round again
        rings = self.rau.__dict__.setdefault("_max_scratch", {})
        ring = rings.get(r)
        if ring is None:
            ring = rings[r] = {"k": 0, "slots": [(DevTensor.zeros(self.rau, r, 1),
                                                  DevTensor.zeros(self.rau, r, 1)) for _ in range(4)]}
        ring["k"] = (ring["k"] + 1) % 4
        v, i = ring["slots"][ring["k"]]
        L.check(self.rau._lib.rau_dev_rowmax(self.rau._h, self.ptr, r, c, v.ptr, i.ptr))
        i.is_int = True
        return v, i

    def select_rows(self, src, key, value):
        r = self.size[0]
        L.check(self.rau._lib.rau_dev_select_rows(self.rau._h, self.ptr, src.ptr, r,
                                                  self.numel() // r, key.ptr, int(value)))
        return self

    def eq_sum(self, other):           # ans:eq(y):sum(), SS:489-492 (int32 tensors)
        out = C.c_int32()
        L.check(self.rau._lib.rau_dev_count_eq(self.rau._h, self.ptr, other.ptr, self.numel(),
                                               C.byref(out)))
        return out.value

    def numpy(self, dtype=None):
        import numpy as np
        a = np.empty(self.size, np.int32 if dtype == "int32" else np.float32)
        L.check(self.rau._lib.rau_dev_download(self.rau._h, a.ctypes.data, self.ptr, a.nbytes))
        return a

    @classmethod
    def ints(cls, rau, host):
        import numpy as np
        a = np.ascontiguousarray(host, np.int32)
        t = cls.zeros(rau, *a.shape)
        L.check(rau._lib.rau_dev_upload(rau._h, t.ptr, a.ctypes.data, a.nbytes))
        return t

    def free(self):
        if self.owned:
            L.check(self.rau._lib.rau_dev_free(self.rau._h, self.ptr))
            self.owned = False


def adam(x, dx, lr, beta1=0.9, beta2=0.999, epsilon=1e-8, state=None):
    """`adam(x, dx, lr, beta1, beta2, epsilon, state)` of utils/optim_updates.lua:59-87 on one flat
    device vector (x, dx = rau.flat(group)); state is a dict holding m, v (DevTensors, created on
    first use) and t -- the Python twin of bindings/rau.lua's RAU.adam, so that SS:770-772 runs
    unchanged above the C ABI."""
    if state is None:
        raise TypeError("adam: pass a dict as `state` (it carries m, v and t between calls)")
    if "m" not in state:
        state["t"] = 0
        state["m"] = DevTensor.zeros(x.rau, x.numel())
        state["v"] = DevTensor.zeros(x.rau, x.numel())
    state["t"] += 1
    L.check(x.rau._lib.rau_dev_adam(x.rau._h, x.ptr, dx.ptr, state["m"].ptr, state["v"].ptr, x.numel(),
                                    float(lr), float(beta1), float(beta2), float(epsilon), state["t"]))


def adam_statements(x, dx, lr, beta1=0.9, beta2=0.999, epsilon=1e-8, state=None):
    """The same update written as the reference's five tensor statements, one rau_dev_* call each
    (what RAU.Tensor's mul / add / addcmul / sqrt / addcdiv give a script that keeps its own adam)."""
    import math
    if state is None:
        raise TypeError("adam_statements: pass a dict as `state` (it carries m, v, tmp and t between calls)")
    if "m" not in state:
        state["t"] = 0
        state["m"] = DevTensor.zeros(x.rau, x.numel())
        state["v"] = DevTensor.zeros(x.rau, x.numel())
        state["tmp"] = DevTensor.zeros(x.rau, x.numel())
    state["m"].mul(beta1).add(1 - beta1, dx)                      # optim_updates.lua:76
    state["v"].mul(beta2).addcmul(1 - beta2, dx, dx)              # :77
    state["tmp"].copy(state["v"]).sqrt().add_scalar(epsilon)      # :78
    state["t"] += 1
    step = lr * math.sqrt(1 - beta2 ** state["t"]) / (1 - beta1 ** state["t"])
    x.addcdiv(-step, state["m"], state["tmp"])                    # :86


def flat(rau, group):
    """(params, grads) of a parameter group as DevTensor views (m:getParameters(), SS:322-324)."""
    w, g, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
    L.check(rau._lib.rau_params(rau._h, L.GROUPS[group], C.byref(w), C.byref(g), C.byref(n)))
    return DevTensor.wrap(rau, w.value, n.value), DevTensor.wrap(rau, g.value, n.value)


def _cp(t):
    return None if t is None else C.c_void_p(t.ptr)


def feval_dev(rau, feats, x, x_len_host, y, hop_w, row_loop=False):
    """feval's tensor half (SS:443-596) over the module-level ABI with DevTensor glue only.

    feats [B,D,S] DevTensor, x: list of T int32 DevTensors [B], x_len_host: numpy [B] (the
    reference reads x_len on the host, SS:455), y int32 DevTensor [B].  row_loop=True copies the
    selected state rows one by one exactly like the reference's `for k=1,B` loop; otherwise one
    rau_dev_select_rows call per step does the same.  Returns (losses, correct counts, uni logits).
    """
    c, lib, h = rau.cfg, rau._lib, rau._h
    x_len_dev = DevTensor.ints(rau, x_len_host)
    max_len = int(max(x_len_host)) if len(x_len_host) else 0

    def select(dst, src, t):
        if row_loop:
            for k in range(c.B):
                if x_len_host[k] == t:
                    dst[k] = src[k]
        else:
            dst.select_rows(src, x_len_dev, t)

    state = [DevTensor.zeros(rau, c.B, c.Q)]
    we = []
    rnn_out = DevTensor.zeros(rau, c.B, c.Q)
    for t in range(max_len):
        o = C.c_void_p()
        L.check(lib.rau_embed_forward(h, t, _cp(x[t]), C.byref(o)))
        we.append(DevTensor.wrap(rau, o, c.B, c.E))
        o = C.c_void_p()
        L.check(lib.rau_deeplstm_forward(h, t, _cp(we[t]), _cp(state[t]), C.byref(o)))
        state.append(DevTensor.wrap(rau, o, c.B, c.Q))
        select(rnn_out, state[t + 1], t + 1)
    att_c, att_h = [DevTensor.zeros(rau, c.B, c.R)], [DevTensor.zeros(rau, c.B, c.R)]
    uni = DevTensor.zeros(rau, c.B, c.K)
    logits, losses, correct = [], [], []
    for hop in range(c.H):
        outs = [C.c_void_p() for _ in range(5)]
        L.check(lib.rau_multimodal_forward(h, hop, _cp(rnn_out), _cp(feats), _cp(att_c[hop]),
                                           _cp(att_h[hop]), *[C.byref(o) for o in outs]))
        lg = DevTensor.wrap(rau, outs[0], c.B, c.K)
        logits.append(lg)
        att_c.append(DevTensor.wrap(rau, outs[3], c.B, c.R))
        att_h.append(DevTensor.wrap(rau, outs[4], c.B, c.R))
        _, ans = lg.max(2)                                   # SS:488
        correct.append(ans.eq_sum(y))                        # SS:489-492
        uni.add(lg)                                          # SS:522-526
        loss = C.c_float()
        L.check(lib.rau_criterion_forward(h, hop, _cp(lg), _cp(y), C.byref(loss)))
        losses.append(loss.value)
    d_c = d_h = None
    d_q = DevTensor.zeros(rau, c.B, c.Q)
    for hop in reversed(range(c.H)):
        o = C.c_void_p()
        L.check(lib.rau_criterion_backward(h, hop, _cp(logits[hop]), _cp(y), float(hop_w[hop]),
                                           C.byref(o)))
        dl = DevTensor.wrap(rau, o, c.B, c.K)
        outs = [C.c_void_p() for _ in range(4)]
        L.check(lib.rau_multimodal_backward(h, hop, _cp(rnn_out), _cp(feats), _cp(att_c[hop]),
                                            _cp(att_h[hop]), _cp(dl), None, None, _cp(d_c),
                                            _cp(d_h), C.byref(outs[0]), None, C.byref(outs[2]),
                                            C.byref(outs[3])))
        d_q.add(DevTensor.wrap(rau, outs[0], c.B, c.Q))      # ConcatTable backward, SS:579
        d_c = DevTensor.wrap(rau, outs[2], c.B, c.R)
        d_h = DevTensor.wrap(rau, outs[3], c.B, c.R)
    d_state = DevTensor.zeros(rau, c.B, c.Q)
    d_out = DevTensor.zeros(rau, c.B, c.Q)
    for t in reversed(range(max_len)):
        d_out.copy(d_state)
        select(d_out, d_q, t + 1)                            # rows REPLACED by dq, SS:584-591
        dx, ds = C.c_void_p(), C.c_void_p()
        L.check(lib.rau_deeplstm_backward(h, t, _cp(we[t]), _cp(state[t]), _cp(d_out),
                                          C.byref(dx), C.byref(ds)))
        d_state = DevTensor.wrap(rau, ds, c.B, c.Q)
        L.check(lib.rau_embed_backward(h, t, _cp(x[t]), dx))
    return losses, correct, uni
