"""The two CPU restatements must agree: oracle/rau_cpu.cc (hand-derived backward)
against oracle/ref_torch.py (PyTorch autograd), fp64, all outputs and all three
gradient groups.  This is what stands in for reference golden vectors (the
reference ships none: parity unpinned)."""
import numpy as np
import pytest

import oracle
from oracle import ref_torch
from tests import util

TOL = 1e-10

CASES = [
    ("small_train", util.SMALL, "ragged", True, None),
    ("edge_train", util.EDGE, "ragged", True, None),
    ("small_eval", util.SMALL, "ragged", False, None),
    ("small_full_len", util.SMALL, "full", True, None),
    ("zero_len_gated", util.SMALL, [0, 6, 1, 3, 6, 2, 0, 4], True, [1.0, 0.0, 1.0]),
    ("wide", dict(B=3, T=3, V=12, E=12, Rq=8, D=20, S=16, M=12, A=8, R=12, K=16, H=2),
     "ragged", True, [1.0, 1.0]),
]


@pytest.mark.parametrize("name,dims,lens,train,hop_w", CASES, ids=[c[0] for c in CASES])
def test_cpp_oracle_matches_autograd(name, dims, lens, train, hop_w):
    sh = util.shapes(dims)
    batch, params, masks = util.make_problem(sh, lens=lens, dtype=np.float64, scale=0.5)
    m = masks if train else None
    a = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], batch["labels"],
                    m, hop_w, dtype=np.float64)
    b = ref_torch.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                       batch["labels"], m, hop_w)
    for k in util.OUT_KEYS + util.GRAD_KEYS:
        err = float(np.max(np.abs(np.asarray(a[k], np.float64) - b[k])))
        assert err < TOL, (k, err)
    assert np.array_equal(a["argmax"], b["argmax"])


def test_f32_oracle_close_to_f64():
    sh = util.shapes(util.SMALL)
    batch, params, masks = util.make_problem(sh, scale=0.5)
    a = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], batch["labels"],
                    masks, dtype=np.float32)
    b = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], batch["labels"],
                    masks, dtype=np.float64)
    for k in util.OUT_KEYS + util.GRAD_KEYS:
        assert util.rel_err(a[k], b[k]) < 1e-4, k


def test_gradients_accumulate_like_accGradParameters():
    """Backward adds into the grad buffers (nn accGradParameters, SS:429-431 zeroes them)."""
    sh = util.shapes(util.EDGE)
    batch, params, masks = util.make_problem(sh, dtype=np.float64, scale=0.5)
    a = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], batch["labels"],
                    masks, dtype=np.float64)
    # linearity in hop weights: doubling every w doubles every gradient
    b = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], batch["labels"],
                    masks, hop_w=np.full(sh.H, 2.0 * sh.H), dtype=np.float64)
    for k in util.GRAD_KEYS:
        assert np.allclose(2.0 * a[k], b[k], rtol=1e-12, atol=1e-14)


def test_oracle_rejects_bad_ids():
    sh = util.shapes(util.EDGE)
    batch, params, masks = util.make_problem(sh, dtype=np.float64)
    bad = batch["tokens"].copy()
    bad[0, 0] = sh.V + 1
    with pytest.raises(ValueError):
        oracle.step(sh, params, batch["feats"], bad, batch["lens"], batch["labels"], masks,
                    dtype=np.float64)
    badl = batch["labels"].copy()
    badl[0] = 0
    with pytest.raises(ValueError):
        oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], badl, masks,
                    dtype=np.float64)
