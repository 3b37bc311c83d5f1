"""PyTorch-CPU autograd restatement of the reference graph -- TEST INFRASTRUCTURE ONLY.

Second, independently written restatement of the RAU path, at the module
granularity of the reference's nngraph (separate dropout / linear / 1x1 conv /
tanh / add / softmax ... ops, backward by autograd).  Two uses:

* tests/test_oracle_agree.py: cross-validates oracle/rau_cpu.cc (hand-derived
  backward) in fp64 -- a transcription error in one is caught by the other.
* bench.py cpu_baseline: "reference graph restated on PyTorch-CPU, N cores"
  (CPU-A).  Never labelled Torch7.

Reference lines (under /root/reference; SS =
experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua):
  word_embed SS:203-206, DeepLSTM model/DeepLSTM.lua:14-71, q_embed SS:231-236,
  i_embed SS:238-242, attbycontent SS:244-252, attselect SS:254-263,
  classifier SS:265-283 (+ model/ATTLSTM.lua:4-74), attbymemory SS:285-290,
  multimodal SS:292-307, feval SS:428-596.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F


def _split(flat, specs):
    """Views into a flat parameter vector: specs = [(name, out, in)], weight then bias."""
    out, off = {}, 0
    for name, o, i in specs:
        out[name + ".W"] = flat[off:off + o * i].view(o, i)
        off += o * i
        out[name + ".b"] = flat[off:off + o]
        off += o
    assert off == flat.numel(), (off, flat.numel())
    return out


def mult_specs(sh):
    Q = 4 * sh.Rq
    return [("q_proj", sh.M, Q), ("h_proj", sh.M, sh.R), ("i_embed", sh.M, sh.D),
            ("att_q", sh.A, sh.M), ("att_i", sh.A, sh.M), ("att_score", 1, sh.A),
            ("att_mem", sh.S, sh.R), ("feat_attprob", sh.M, sh.S),
            ("lstm_i2h", 4 * sh.R, sh.M), ("lstm_h2h", 4 * sh.R, sh.R),
            ("lstm_out", sh.M, sh.R), ("cls", sh.K, sh.M), ("do_pred", 1, sh.M)]


def rnn_specs(sh):
    return [("l1_i2h", 4 * sh.Rq, sh.E), ("l1_h2h", 4 * sh.Rq, sh.Rq),
            ("l2_i2h", 4 * sh.Rq, sh.Rq), ("l2_h2h", 4 * sh.Rq, sh.Rq)]


_NUDGE = 0.0   # see step(bf16_nudge=...)


def _rb(x):
    """Round to bfloat16 (RNE, via float32 like the device's staging path) and back.  With _NUDGE = eta
    the value is scaled by (1 + eta) first: an operand further than eta (relative) from a bf16 rounding
    boundary rounds to the same bf16 value as before, one closer than that flips to its neighbour --
    exactly the operands whose rounding the device's own f32 error (~3e-7) can decide differently."""
    v = x.detach().to(torch.float32)
    if _NUDGE:
        v = v * (1.0 + _NUDGE)
    return v.to(torch.bfloat16).to(x.dtype)


class _Conv1x1BF16(torch.autograd.Function):
    """y[b,o,s] = sum_i W[o,i] x[b,i,s] (+ bias) as librau's RAU_BF16 mode computes it: every GEMM
    of the forward AND the backward rounds both of its operands to bf16 and accumulates exactly;
    the bias, and everything outside the GEMMs, is untouched.  (The straight-through backward
    of a rounded forward would not round dY; the device does, because dY is a GEMM operand.)"""

    @staticmethod
    def forward(ctx, x, W, b, exact_dx=False):
        ctx.save_for_backward(x, W)
        ctx.exact_dx = exact_dx
        return torch.einsum("oi,bis->bos", _rb(W), _rb(x)) + b.view(1, -1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dyr = _rb(dy)
        # exact_dx: on 14x14 maps whose sizes dgrad16.hip does not take (M % 128, A % 32) the attention
        # dgrad runs on the exact-f32 per-sample kernel in every dtype: no operand is rounded
        dx = (torch.einsum("oi,bos->bis", W, dy) if ctx.exact_dx
              else torch.einsum("oi,bos->bis", _rb(W), dyr))
        dW = torch.einsum("bos,bis->oi", dyr, _rb(x))
        return dx, dW, dy.sum((0, 2)), None


class _LinearBF16W(torch.autograd.Function):
    """y = x W^T + b as librau's RAU_BF16 mode computes a Linear layer (BASELINE.json configs[2]: "bf16 MFMA
    gate/classifier GEMMs"): each of its three products -- forward y = x W^T, input gradient dx = dy W,
    weight gradient dW = dy^T x -- rounds BOTH of its operands to bf16 (round to nearest even) and
    accumulates exactly (the device: in f32); the bias is added unrounded and the bias gradient is the
    column sum of the unrounded dy."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return F.linear(_rb(x), _rb(W), b)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dyr = _rb(dy)
        return dyr @ _rb(W), dyr.t() @ _rb(x), dy.sum(0)


_BF16_LINEAR = False   # set by step(): the Linear layers' three products with bf16-rounded operands


@contextlib.contextmanager
def bf16_emulation(nudge=0.0):
    """The module-granular functions below (deep_lstm, multimodal with bf16=True, ...) and their autograd
    backward under the RAU_BF16 emulation, outside step(): tests of ONE clone with given inputs."""
    global _NUDGE, _BF16_LINEAR
    old = (_NUDGE, _BF16_LINEAR)
    _NUDGE, _BF16_LINEAR = float(nudge), True
    try:
        yield
    finally:
        _NUDGE, _BF16_LINEAR = old


def _lin(x, W, b):
    """nn.Linear; in the bf16 emulation its products round their operands (see _LinearBF16W)."""
    return _LinearBF16W.apply(x, W, b) if _BF16_LINEAR else F.linear(x, W, b)


def _conv1x1(x3, W, b, bf16, exact_dx=False):
    """1x1 SpatialConvolution on [B, C, S] (reference SS:240, SS:247)."""
    if bf16:
        return _Conv1x1BF16.apply(x3, W, b, exact_dx)
    return F.conv2d(x3.unsqueeze(3), W.view(W.shape[0], W.shape[1], 1, 1), b).squeeze(3)


def _drop(x, mask, p):
    """nn.Dropout v2 with an explicit keep mask (None = evaluate mode)."""
    if mask is None:
        return x
    return x * (mask.to(x.dtype) / (1.0 - p))


def deep_lstm(sh, P, x, state, mask_mid):
    """model/DeepLSTM.lua:14-71.  state = [c1 h1 c2 h2] packed on dim 1."""
    Rq = sh.Rq
    outs = []
    inp = x
    for L in range(2):
        prev_c = state[:, 2 * L * Rq:(2 * L + 1) * Rq]
        prev_h = state[:, (2 * L + 1) * Rq:(2 * L + 2) * Rq]
        if L == 1:
            inp = _drop(outs[1], mask_mid, sh.p_rnn)
        n = "l%d" % (L + 1)
        sums = _lin(inp, P[n + "_i2h.W"], P[n + "_i2h.b"]) + \
            _lin(prev_h, P[n + "_h2h.W"], P[n + "_h2h.b"])
        sig = torch.sigmoid(sums[:, :3 * Rq])
        in_gate, forget_gate, out_gate = sig[:, :Rq], sig[:, Rq:2 * Rq], sig[:, 2 * Rq:]
        in_transform = torch.tanh(sums[:, 3 * Rq:])
        next_c = forget_gate * prev_c + in_gate * in_transform
        next_h = out_gate * torch.tanh(next_c)
        outs += [next_c, next_h]
    return torch.cat(outs, 1)


def att_lstm(sh, P, x, prev_c, prev_h):
    """model/ATTLSTM.lua:4-28 with one layer and dropout 0."""
    R = sh.R
    gates = _lin(x, P["lstm_i2h.W"], P["lstm_i2h.b"]) + \
        _lin(prev_h, P["lstm_h2h.W"], P["lstm_h2h.b"])
    g = gates.view(-1, 4, R)
    in_gate = torch.sigmoid(g[:, 0])
    in_transform = torch.tanh(g[:, 1])
    forget_gate = torch.sigmoid(g[:, 2])
    out_gate = torch.sigmoid(g[:, 3])
    next_c = forget_gate * prev_c + in_gate * in_transform
    next_h = out_gate * torch.tanh(next_c)
    return next_c, next_h


def multimodal(sh, P, q, feats4d, prev_c, prev_h, mq, mx, mmf, bf16=False):
    """SS:292-307: returns (logits, do_pred, attprob, next_c, next_h).
    bf16=True emulates librau's RAU_BF16 mode on the two 1x1 convolutions."""
    B = q.shape[0]
    # q_embed
    qf = torch.tanh(_lin(_drop(q, mq, sh.p_q), P["q_proj.W"], P["q_proj.b"]) +
                    _lin(prev_h, P["h_proj.W"], P["h_proj.b"]))
    # i_embed: Dropout -> 1x1 conv -> Tanh -> Reshape(M, S)
    xi = _drop(feats4d, mx, sh.p_x)
    ifeat = torch.tanh(_conv1x1(xi.reshape(B, sh.D, sh.S), P["i_embed.W"], P["i_embed.b"], bf16))
    # attbycontent
    qatt = _lin(qf, P["att_q.W"], P["att_q.b"]).unsqueeze(2).expand(B, sh.A, sh.S)
    # librau dispatch: per-sample exact-f32 dgrad when 176 < S <= 208 and S % 4 == 0 (gemm_sample.hip)
    iproj = _conv1x1(ifeat, P["att_i.W"], P["att_i.b"], bf16,
                     exact_dx=(sh.S % 4 == 0 and 176 < sh.S <= 208 and sh.M % 4 == 0
                               and not (sh.S == 196 and sh.M % 128 == 0 and sh.A % 32 == 0)))
    addfeat = torch.tanh(iproj + qatt).reshape(B, sh.A, sh.S, 1)
    attscore = F.conv2d(addfeat, P["att_score.W"].view(1, sh.A, 1, 1),
                        P["att_score.b"]).reshape(B, sh.S)
    # attbymemory
    attprob = torch.softmax(attscore + _lin(prev_h, P["att_mem.W"], P["att_mem.b"]), dim=1)
    # attselect
    attfeat = (ifeat * attprob.unsqueeze(1).expand(B, sh.M, sh.S)).sum(2)
    # classifier
    join_input = (qf + attfeat) + _lin(attprob, P["feat_attprob.W"], P["feat_attprob.b"])
    next_c, next_h = att_lstm(sh, P, join_input, prev_c, prev_h)
    merge = _drop(join_input + _lin(next_h, P["lstm_out.W"], P["lstm_out.b"]), mmf, sh.p_mf)
    score = _lin(merge, P["cls.W"], P["cls.b"])
    # (a one-column Linear: a dot product per row in the criterion head's kernel, exact f32 in every dtype)
    do_pred = torch.sigmoid(F.linear(merge, P["do_pred.W"], P["do_pred.b"])).sum(1)
    return score, do_pred, attprob, next_c, next_h


def step(sh, params, feats, tokens, lens, labels, masks=None, hop_w=None,
         backward=True, dtype=torch.float64, bf16=False, bf16_nudge=0.0, per_sample_abs=False):
    """One feval forward(+backward), SS:428-596.  Same I/O convention as oracle.step.
    bf16_nudge (with bf16=True): flip the rounding of every GEMM operand that lies within that
    relative distance of a bf16 rounding boundary (tests/test_gpu_bf16.py derives its bar from it).
    per_sample_abs: also return gabs_{embed,rnn,mult} = sum_b |g_b|, g_b the gradient of sample b's
    share of the loss (B extra backward passes): the UN-CANCELLED magnitude of every gradient element
    over the batch sum -- the scale a rounding error has to be read against where the sum itself
    nearly cancels (a bias gradient summing dz over a 2-position softmax, whose rows sum to zero)."""
    global _NUDGE, _BF16_LINEAR
    _NUDGE = float(bf16_nudge) if bf16 else 0.0
    _BF16_LINEAR = bool(bf16)
    try:
        return _step(sh, params, feats, tokens, lens, labels, masks, hop_w, backward, dtype, bf16,
                     per_sample_abs)
    finally:
        _NUDGE = 0.0
        _BF16_LINEAR = False


def _step(sh, params, feats, tokens, lens, labels, masks, hop_w, backward, dtype, bf16,
          per_sample_abs=False):
    t = lambda a: torch.as_tensor(a).to(dtype)
    flat = {k: t(params[k]).clone().requires_grad_(backward) for k in ("embed", "rnn", "mult")}
    Emb = flat["embed"].view(sh.V, sh.E)
    Pr = _split(flat["rnn"], rnn_specs(sh))
    Pm = _split(flat["mult"], mult_specs(sh))
    feats4d = t(feats).reshape(sh.B, sh.D, sh.S, 1)
    tokens = torch.as_tensor(tokens).long()
    lens = torch.as_tensor(lens).long()
    mk = lambda k: None if (masks is None or masks.get(k) is None) else torch.as_tensor(masks[k])
    m_we, m_rnn, m_q, m_x, m_mf = mk("we"), mk("rnn"), mk("q"), mk("x"), mk("mf")
    if hop_w is None:
        hop_w = [float(sh.H)] * sh.H
    B, Q = sh.B, 4 * sh.Rq
    # encoder
    max_len = int(lens.max())
    state = torch.zeros(B, Q, dtype=dtype)
    q = torch.zeros(B, Q, dtype=dtype)
    for tt in range(1, max_len + 1):
        we = torch.tanh(_drop(Emb[tokens[tt - 1] - 1], None if m_we is None else m_we[tt - 1], sh.p_we))
        state = deep_lstm(sh, Pr, we, state, None if m_rnn is None else m_rnn[tt - 1])
        sel = (lens == tt).unsqueeze(1)
        q = torch.where(sel, state, q)
    # hops
    c = torch.zeros(B, sh.R, dtype=dtype)
    h = torch.zeros(B, sh.R, dtype=dtype)
    out = {"losses": [], "argmax": [], "logits": [], "dopred": [], "att": [],
           "att_c": [], "att_h": []}
    total = 0.0
    rows = torch.zeros(B, dtype=dtype)
    y = None if labels is None else torch.as_tensor(labels).long() - 1
    for hop in range(sh.H):
        score, dp, a, c, h = multimodal(
            sh, Pm, q, feats4d, c, h,
            None if m_q is None else m_q[hop],
            None if m_x is None else m_x[hop].reshape(sh.B, sh.D, sh.S, 1),
            None if m_mf is None else m_mf[hop], bf16=bf16)
        out["logits"].append(score.detach())
        out["dopred"].append(dp.detach())
        out["att"].append(a.detach())
        out["att_c"].append(c.detach())
        out["att_h"].append(h.detach())
        out["argmax"].append(torch.argmax(score.detach(), dim=1) + 1)
        if y is not None:
            loss = F.cross_entropy(score, y)           # CrossEntropyCriterion, SS:518
            out["losses"].append(loss.detach())
            total = total + float(hop_w[hop]) * loss   # dpred:mul(w), SS:569
            if per_sample_abs:                         # the same loss, one term per sample
                rows = rows + float(hop_w[hop]) * F.cross_entropy(score, y, reduction="none") / B
    res = {k: torch.stack(v).numpy() for k, v in out.items() if v}
    res["q"] = q.detach().numpy()
    if backward and per_sample_abs:
        keys = ("embed", "rnn", "mult")
        acc = {k: torch.zeros_like(flat[k]) for k in keys}
        for b in range(B):
            gs = torch.autograd.grad(rows[b], [flat[k] for k in keys], retain_graph=True, allow_unused=True)
            for k, g in zip(keys, gs):
                if g is not None:
                    acc[k] += g.abs()
        for k in keys:
            res["gabs_" + k] = acc[k].numpy()
    if backward:
        total.backward()
        res["g_embed"] = flat["embed"].grad.numpy()
        res["g_rnn"] = flat["rnn"].grad.numpy()
        res["g_mult"] = flat["mult"].grad.numpy()
    return res
