"""Shared helpers for the parity tests: problem construction and comparison."""
from __future__ import annotations

import numpy as np

import oracle
from rau_vqa_amd import synth

SMALL = dict(B=8, T=6, V=50, E=8, Rq=16, D=24, S=12, M=40, A=20, R=16, K=12, H=3)
EDGE = dict(B=5, T=4, V=9, E=4, Rq=4, D=4, S=4, M=4, A=4, R=4, K=4, H=1)
MEDIUM = dict(B=70, T=9, V=300, E=200, Rq=64, D=72, S=196, M=136, A=132, R=68, K=1000, H=2)

OUT_KEYS = ("losses", "logits", "dopred", "att", "q", "att_c", "att_h")
GRAD_KEYS = ("g_embed", "g_rnn", "g_mult")


def shapes(d, **over) -> oracle.Shapes:
    kw = dict(d)
    kw.update(over)
    return oracle.Shapes(**kw)


def make_problem(sh: oracle.Shapes, seed=123, lens="ragged", dtype=np.float32, scale=None):
    ne, nr, nm = oracle.group_sizes(sh)
    batch = synth.make_batch(sh.B, sh.T, sh.V, sh.D, sh.S, sh.K, seed=seed, lens=lens, dtype=dtype)
    lo, hi = (-0.08, 0.08) if scale is None else (-scale, scale)
    params = synth.make_params({"embed": ne, "rnn": nr, "mult": nm}, seed=seed, lo=lo, hi=hi,
                               dtype=dtype)
    probs = {k: getattr(sh, "p_" + k) for k in oracle.MASK_SITES}
    masks = synth.make_masks(oracle.mask_shapes(sh), probs, seed=seed)
    return batch, params, masks


def rel_err(a, b):
    """max |a-b| / max |b| (max-norm relative error of a whole tensor)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.max(np.abs(b))
    if den == 0:
        return float(np.max(np.abs(a - b)))
    return float(np.max(np.abs(a - b)) / den)


def layer_slices(layout):
    """[(name, slice)] from a rau_layout listing [(name, offset, rows, cols)]."""
    return [(n, slice(off, off + r * c)) for n, off, r, c in layout]


def argmax_margin_ok(logits_ref, got, ref, margin=1e-5):
    """Answer indices must be EQUAL wherever the reference decides them: rows whose top-2 margin in the
    fp64 reference logits exceeds `margin` (relative; 1e-5 = 25x the measured f32 logit error of
    ~4e-7, the same bar as the 1k-sample tests).  Returns (ok, decided, total); the caller reports
    total - decided = the rows a 1e-5 tie leaves undecided (0 on every committed seed)."""
    srt = np.sort(logits_ref, axis=-1)
    gap = srt[..., -1] - srt[..., -2]
    decided = gap > margin * np.maximum(1.0, np.abs(srt[..., -1]))
    return bool(np.all(got[decided] == ref[decided])), int(decided.sum()), int(decided.size)
