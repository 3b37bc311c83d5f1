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


def _philox_vec(c0, c1, c2, c3, k0, k1):
    """Vectorised numpy Philox4x32-10 (same rounds as tests/test_philox.py, Random123 KATs there)."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = [np.asarray(x, np.uint64) for x in (c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask, p1 & mask,
             ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask, p0 & mask]
        k0, k1 = (k0 + np.uint64(W0)) & mask, (k1 + np.uint64(W1)) & mask
    return c


def device_normals(n, noise_seed, step_t, group):
    """Host restatement of the normals rau_noise_clip_adam draws for flat element e of `group`
    (kernels.hip k_add_noise_sqnorm): Philox counter (e/4, group, 0xA5A5), key noise_seed +
    3*step_t + group, two Box-Muller pairs per 128-bit block."""
    q = np.arange((n + 3) // 4, dtype=np.uint64)
    seed = (noise_seed + step_t * 3 + group) & 0xFFFFFFFFFFFFFFFF
    o = _philox_vec(q & np.uint64(0xFFFFFFFF), q >> np.uint64(32), np.full_like(q, group),
                    np.full_like(q, 0xA5A5), seed & 0xFFFFFFFF, seed >> 32)
    z = np.empty((q.size, 4))
    for j, (a, b) in enumerate(((o[0], o[1]), (o[2], o[3]))):
        u1 = ((a >> np.uint64(8)).astype(np.float64) + 1.0) / 16777216.0
        u2 = (b >> np.uint64(8)).astype(np.float64) / 16777216.0
        r = np.sqrt(-2.0 * np.log(u1))
        z[:, 2 * j] = r * np.cos(2 * np.pi * u2)
        z[:, 2 * j + 1] = r * np.sin(2 * np.pi * u2)
    return z.reshape(-1)[:n]


@pytest.mark.gpu
def test_hip_gradient_noise_statistics_and_stream():
    """The noise half of SS:597-605: zero gradients, clip far away, lr = 0, then the gradient
    buffers hold exactly the injected noise.  Checks sigma = sqrt(eta / ((t+1) * gamma)) (gamma
    MULTIPLIES, SS:598), mean, the element-wise values against a host restatement of the Philox
    -> Box-Muller stream, reproducibility, and that (step, group) change the draw."""
    from rau_vqa_amd.model import RAU, Config
    dims = dict(B=4, T=4, V=6000, E=200, Rq=64, D=64, S=196, M=512, A=256, R=512, K=1000, H=2)
    m = RAU(Config(**dims))
    m.init_uniform(seed=1)
    p0 = m.get_params()
    sizes = m.group_sizes()
    assert sizes["mult"] > 1_000_000 and sizes["embed"] > 1_000_000
    eta, gamma = 0.01, 0.55
    zero = {k: np.zeros(n, np.float32) for k, n in sizes.items()}

    def draw(step_t, seed):
        m.set_grads(zero)
        norms = m.update(step_t=step_t, lr=0.0, mult_lr=0.0, eta=eta, gamma=gamma, clip=1e30,
                         noise_seed=seed)
        return m.get_grads(), norms

    for step_t in (0, 41):
        g, norms = draw(step_t, seed=77)
        sigma = np.sqrt(eta / ((step_t + 1) * gamma))
        for gi, k in enumerate(("embed", "rnn", "mult")):
            x = g[k].astype(np.float64)
            n = x.size
            assert abs(x.mean()) < 4 * sigma / np.sqrt(n), (k, x.mean())
            assert abs(x.std() / sigma - 1.0) < 0.01, (k, x.std(), sigma)
            assert abs(np.mean(np.abs(x) > 2 * sigma) - 0.0455) < 0.002     # normal tails
            ref = device_normals(n, 77, step_t, gi) * sigma
            assert np.max(np.abs(x - ref)) < 2e-5 * sigma * 6 + 1e-9, k
            assert abs(norms[gi] - np.linalg.norm(x)) < 1e-4 * np.linalg.norm(x)
    a, _ = draw(3, seed=5)
    b, _ = draw(3, seed=5)
    c, _ = draw(4, seed=5)
    d, _ = draw(3, seed=6)
    for k in a:
        assert np.array_equal(a[k], b[k])                   # same (seed, step): same bits
        assert not np.array_equal(a[k], c[k])               # another step: another draw
        assert not np.array_equal(a[k], d[k])
    assert not np.array_equal(a["rnn"][:1000], a["mult"][:1000])   # groups draw independently
    # lr = 0: parameters untouched by all of the above
    p1 = m.get_params()
    for k in p0:
        assert np.array_equal(p0[k], p1[k])
    m.close()
