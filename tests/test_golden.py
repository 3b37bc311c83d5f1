"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle must reproduce them (guards the oracle against silent edits).
GPU: librau.so must match them to the parity bar (fp32 1e-4 rel, indices exact)."""
import glob
import os

import numpy as np
import pytest

import oracle
from tests import util

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def load(path):
    z = np.load(path, allow_pickle=False)
    dims = {str(n): int(v) for n, v in zip(z["dim_names"], z["dims"])}
    sh = util.shapes(dims)
    batch = {k: z["in_" + k] for k in ("feats", "tokens", "lens", "labels")}
    params = {k: z["p_" + k] for k in ("embed", "rnn", "mult")}
    masks = {}
    for k, shape in oracle.mask_shapes(sh).items():
        n = int(np.prod(shape))
        masks[k] = np.unpackbits(z["m_" + k])[:n].reshape(shape)
    return sh, batch, params, masks, bool(z["train"]), z["hop_w"], z


def test_fixtures_exist():
    assert len(FIXTURES) >= 3


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_reproduces_fixture(path):
    sh, batch, params, masks, train, hop_w, z = load(path)
    ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                      batch["labels"], masks if train else None, hop_w, dtype=np.float64)
    for k in util.OUT_KEYS + util.GRAD_KEYS:
        assert util.rel_err(ref[k], z["o_" + k]) < 1e-6, k
    assert np.array_equal(ref["argmax"], z["o_argmax"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_hip_matches_fixture(path):
    from tests.test_gpu_parity import run_gpu
    sh, batch, params, masks, train, hop_w, z = load(path)
    got, _ = run_gpu(sh, batch, params, masks, hop_w, "train" if train else "eval")
    for k in util.OUT_KEYS + util.GRAD_KEYS:
        assert util.rel_err(got[k], z["o_" + k]) < 1e-4, k
    ok, _, _ = util.argmax_margin_ok(z["o_logits"], got["argmax"], z["o_argmax"])
    assert ok
