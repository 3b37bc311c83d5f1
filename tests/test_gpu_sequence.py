"""Stale-state check: ONE context runs several different steps in a row -- train / evaluate, new question
lengths (the unroll length changes), gated hops, explicit masks or device Philox masks read back,
gradients zeroed or accumulated -- and every step must match the fp64 autograd oracle run from scratch on
that step's inputs (1e-4, like tests/test_gpu_parity.py).  tools/soak_sequence.py is the long form."""
import numpy as np
import pytest

import oracle
from oracle import ref_torch
from rau_vqa_amd import synth
from tests import util
from tests.test_gpu_fuzz import draw

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_one_context_many_different_steps(seed):
    from rau_vqa_amd.model import RAU, Config
    rng = np.random.default_rng(9000 + seed)
    dims = draw(rng)
    if seed % 2:   # above the 64-sample switch (fused attention kernels, grouped conv launches)
        dims["B"] = int(rng.integers(65, 100))
        dims["S"] = int(rng.choice([196, 49, dims["S"]]))
    sh = util.shapes(dims)
    _, params, _ = util.make_problem(sh, seed=seed, scale=0.3)
    m = RAU(Config(**{k: getattr(sh, k) for k in
                      ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H",
                       "p_we", "p_rnn", "p_q", "p_x", "p_mf")}))
    m.set_params(params)
    acc = None
    for it in range(5):
        lens = rng.integers(0, dims["T"] + 1, dims["B"]).astype(np.int32)
        if lens.max() == 0:
            lens[0] = max(1, dims["T"] // 2)
        batch = synth.make_batch(sh.B, sh.T, sh.V, sh.D, sh.S, sh.K, seed=1000 * seed + it, lens=lens)
        hop_w = rng.choice([0.0, 1.0, float(dims["H"])], dims["H"]).astype(np.float32)
        if not hop_w.any():
            hop_w[int(rng.integers(0, dims["H"]))] = 1.0
        train = bool(rng.integers(0, 3))
        masks = None
        if train:
            m.training()
            if rng.integers(0, 2):
                probs = {k: getattr(sh, "p_" + k) for k in oracle.MASK_SITES}
                masks = synth.make_masks(oracle.mask_shapes(sh), probs, seed=77 * seed + it)
                m.set_masks(masks)
            else:
                m.set_dropout_seed(500 + seed, it)
                masks = {k: m.get_mask(k) for k in oracle.MASK_SITES}
        else:
            m.evaluate()
        m.set_batch(batch["feats"], batch["tokens"], batch["lens"], batch["labels"])
        zero = acc is None or bool(rng.integers(0, 2))
        if zero:
            m.zero_grads()
        m.forward()
        out = m.outputs()
        m.backward(hop_w)
        g = m.get_grads()
        ref = ref_torch.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                             batch["labels"], masks, hop_w)
        new = {k: ref["g_" + k].astype(np.float64) for k in ("embed", "rnn", "mult")}
        acc = new if zero else {k: acc[k] + new[k] for k in new}
        errs = {k: util.rel_err(out[k], ref[k]) for k in util.OUT_KEYS}
        for k in acc:
            errs["g_" + k] = (util.rel_err(g[k], acc[k]) if np.max(np.abs(acc[k])) > 1e-12
                              else float(np.max(np.abs(g[k] - acc[k]))))
        bad = {k: v for k, v in errs.items() if not v < TOL}
        assert not bad, f"step {it} train={train} zero={zero} hop_w={hop_w} dims={dims}: {bad}"
    m.close()
