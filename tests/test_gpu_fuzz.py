"""Seeded shape fuzzing of the HIP path against the fp64 oracle: odd batch sizes, feature maps
whose position count is not a multiple of 4 (internal pitch padding), single-hop and odd-hop
networks (launch-group partition), reduction lengths that end inside a K-step, ragged and
zero-length questions, both modes.  Same bar as test_gpu_parity."""
import numpy as np
import pytest

from tests import util
from tests.test_gpu_parity import check

pytestmark = pytest.mark.gpu


def draw(rng):
    m4 = lambda lo, hi: int(rng.integers(lo, hi + 1)) * 4
    return dict(B=int(rng.integers(1, 40)), T=int(rng.integers(1, 9)), V=int(rng.integers(5, 80)),
                E=m4(1, 12), Rq=m4(1, 12), D=m4(1, 40), S=int(rng.integers(1, 60)),
                M=m4(1, 40), A=m4(1, 24), R=m4(1, 12), K=m4(1, 30), H=int(rng.integers(1, 6)))


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_shapes(seed):
    rng = np.random.default_rng(1000 + seed)
    dims = draw(rng)
    sh = util.shapes(dims)
    lens = rng.integers(0, dims["T"] + 1, dims["B"]).astype(np.int32)
    if lens.max() == 0:
        lens[0] = dims["T"]
    hop_w = rng.choice([0.0, 1.0, float(dims["H"])], dims["H"]).astype(np.float32)
    if not hop_w.any():
        hop_w[0] = 1.0
    mode = "train" if seed % 3 else "eval"
    check(sh, seed=seed, lens=lens, mode=mode, hop_w=hop_w, scale=0.3)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_shapes_module_level_feval(seed):
    """The reference's loops over the module-level calls on random shapes: against the step-level
    path on the same ctx and masks (same numbers up to summation order)."""
    import torch
    from rau_vqa_amd import modules
    from tests.test_gpu_modules import make_model, cuda
    rng = np.random.default_rng(2000 + seed)
    dims = draw(rng)
    dims["S"] = max(4, dims["S"] // 4 * 4)          # module-level calls need S % 4 == 0
    sh = util.shapes(dims)
    lens = rng.integers(0, dims["T"] + 1, dims["B"]).astype(np.int32)
    lens[0] = dims["T"]
    batch, params, masks = util.make_problem(sh, seed=seed, lens=lens, scale=0.3)
    hop_w = np.full(sh.H, 1.0, np.float32)
    m = make_model(sh, params, masks)
    m.zero_grads()
    modules.feval(m, cuda(batch["feats"]), cuda(batch["tokens"], torch.int32),
                  cuda(batch["lens"], torch.int32), cuda(batch["labels"], torch.int32), hop_w)
    m.sync()
    g_mod = m.get_grads()
    m.set_batch(batch["feats"], batch["tokens"], batch["lens"], batch["labels"])
    m.zero_grads()
    m.forward()
    m.backward(hop_w)
    g_step = m.get_grads()
    m.close()
    for k in g_step:
        assert util.rel_err(g_mod[k], g_step[k]) < 2e-5, (k, dims)
