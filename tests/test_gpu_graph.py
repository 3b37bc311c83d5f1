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


# bf16 mode takes other kernels (dgrad16 / wgrad16 / the bf16-storing dropout and attention-backward
# forms) and the side-stream split of the chain's non-recurrent GEMMs; BF16_TILES has the widths
# dgrad16.hip (M % 128, A % 32) and wgrad16.hip (M, D % 256) accept on a 14x14 map.
BF16_TILES = dict(B=12, T=6, V=60, E=64, Rq=64, D=256, S=196, M=256, A=64, R=64, K=200, H=3)


@pytest.mark.parametrize("dims,dtype", [(util.SMALL, "f32"), (util.MEDIUM, "f32"),
                                        (util.SMALL, "bf16"), (util.MEDIUM, "bf16"), (BF16_TILES, "bf16")],
                         ids=["small-f32", "medium-f32", "small-bf16", "medium-bf16", "tiles-bf16"])
def test_graph_step_matches_eager_bitwise(dims, dtype):
    sh = util.shapes(dims)
    _, params, _ = util.make_problem(sh, scale=0.3)
    eager, graph = _mk(sh, params, dtype), _mk(sh, params, dtype)
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
