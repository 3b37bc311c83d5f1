"""Gradient noise + clip + Adam (SS:597-630, utils/optim_updates.lua:59-87)."""
import numpy as np
import pytest

import oracle


def adam_ref(x, g, m, v, noise, step_t, t, lr, b1=0.9, b2=0.999, eps=1e-8, eta=0.01,
             gamma=0.55, clip=0.1):
    g = g + noise * np.sqrt(eta / ((step_t + 1) * gamma))   # gamma multiplies (SS:598)
    n = np.linalg.norm(g)
    if n > clip:
        g = g * (clip / n)
    m = m * b1 + (1 - b1) * g
    v = v * b2 + (1 - b2) * g * g
    step = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return x - step * m / (np.sqrt(v) + eps), g, m, v, n        # eps OUTSIDE the sqrt


def test_oracle_update_matches_numpy():
    rng = np.random.default_rng(0)
    n = 1000
    x, g = rng.standard_normal(n), rng.standard_normal(n) * 0.01
    m, v, noise = np.zeros(n), np.zeros(n), rng.standard_normal(n)
    xs, gs, ms, vs = x.copy(), g.copy(), m.copy(), v.copy()
    for t in range(1, 4):
        norm = oracle.noise_clip_adam(xs, gs, ms, vs, noise, step_t=t - 1, adam_t=t, lr=3e-3)
        x, g2, m, v, n_ref = adam_ref(x, g, m, v, noise, t - 1, t, 3e-3)
        assert abs(norm - n_ref) < 1e-12
        assert np.allclose(xs, x, rtol=1e-12, atol=1e-14)
        assert np.allclose(gs, g2, rtol=1e-12, atol=1e-14)
        g = rng.standard_normal(n) * 0.01
        gs[:] = g


@pytest.mark.gpu
def test_hip_update_matches_oracle_without_noise():
    from rau_vqa_amd.model import RAU, Config
    from tests import util
    sh = util.shapes(util.SMALL)
    _, params, _ = util.make_problem(sh)
    rng = np.random.default_rng(3)
    grads = {k: (rng.standard_normal(v.size) * 0.01).astype(np.float32) for k, v in params.items()}
    m = RAU(Config(**{k: getattr(sh, k) for k in
                      ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H")}))
    m.set_params(params)
    ref = {k: v.astype(np.float64) for k, v in params.items()}
    mom = {k: (np.zeros(v.size), np.zeros(v.size)) for k, v in params.items()}
    for t in range(1, 4):
        m.set_grads(grads)
        norms = m.update(step_t=t - 1, lr=3e-3, mult_lr=3e-4, eta=0.0)
        for i, k in enumerate(("embed", "rnn", "mult")):
            g = grads[k].astype(np.float64)
            lr = 3e-4 if k == "mult" else 3e-3         # SS:770-772
            n = oracle.noise_clip_adam(ref[k], g, mom[k][0], mom[k][1], None, t - 1, t, lr)
            assert abs(norms[i] - n) < 1e-5 * max(1.0, n)
        got = m.get_params()
        for k in ref:
            assert util.rel_err(got[k], ref[k]) < 1e-5, k
    m.close()
