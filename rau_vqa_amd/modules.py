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
