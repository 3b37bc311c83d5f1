"""rau_graph_step: zero_grads + forward + backward replayed as one hipGraph must give the same
bits as the three calls -- across steps with changing Philox keys, hop weights, batches and a
changing longest-question length (which re-captures)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _mk(sh, params, dtype="f32"):
    from rau_vqa_amd.model import RAU, Config
    m = RAU(Config(**{k: getattr(sh, k) for k in
                      ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H")}, dtype=dtype))
    m.set_params(params)
    m.training()
    return m


@pytest.mark.parametrize("dims", [util.SMALL, util.MEDIUM])
def test_graph_step_matches_eager_bitwise(dims):
    sh = util.shapes(dims)
    _, params, _ = util.make_problem(sh, scale=0.3)
    eager, graph = _mk(sh, params), _mk(sh, params)
    rng = np.random.default_rng(3)
    for it in range(5):
        lens = "ragged" if it % 2 else np.full(sh.B, max(1, sh.T - it), np.int32)
        batch, _, _ = util.make_problem(sh, seed=50 + it, lens=lens, scale=0.3)
        hop_w = rng.choice([1.0, float(sh.H)], sh.H).astype(np.float32)
        if it == 3:
            hop_w[-1] = 0.0                      # fewer active hops: another graph shape
        outs = []
        for m, use_graph in ((eager, False), (graph, True)):
            m.set_batch(batch["feats"], batch["tokens"], batch["lens"], batch["labels"])
            m.set_dropout_seed(11, it)
            if use_graph:
                m.graph_step(hop_w)
            else:
                m.zero_grads()
                m.forward()
                m.backward(hop_w)
            g = m.get_grads()
            outs.append((m.losses(), m.logits(), g["embed"], g["rnn"], g["mult"]))
        for a, b in zip(*outs):
            assert np.array_equal(a, b), f"step {it}"
    # the update consumes graph-produced gradients like any others
    assert np.all(np.isfinite(graph.update(step_t=0, eta=0.0)))
    eager.close()
    graph.close()
