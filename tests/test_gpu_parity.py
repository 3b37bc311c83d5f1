"""GPU parity: librau.so (HIP, through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): integer answer indices bit-exact, fp32
logits / gradients within 1e-4 relative (max-norm per tensor) of the oracle,
here evaluated against the oracle in float64 on the same seeded inputs and the
same explicit dropout masks.
"""
import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu

TOL = 1e-4


def run_gpu(sh, batch, params, masks, hop_w, mode="train"):
    from rau_vqa_amd.model import RAU, Config
    cfg = Config(**{k: getattr(sh, k) for k in
                    ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H",
                     "p_we", "p_rnn", "p_q", "p_x", "p_mf")})
    m = RAU(cfg)
    m.set_params(params)
    if mode == "train":
        m.training()
        m.set_masks(masks)
    else:
        m.evaluate()
    m.set_batch(batch["feats"], batch["tokens"], batch["lens"], batch["labels"])
    m.zero_grads()
    m.forward()
    out = m.outputs()
    m.backward(hop_w)
    g = m.get_grads()
    out.update({"g_embed": g["embed"], "g_rnn": g["rnn"], "g_mult": g["mult"]})
    layouts = {k: m.layout(k) for k in ("embed", "rnn", "mult")}
    m.close()
    return out, layouts


def check(sh, seed=123, lens="ragged", mode="train", hop_w=None, scale=None, torch_oracle=False):
    """torch_oracle: take the fp64 reference from the autograd restatement (oracle/ref_torch.py,
    BLAS-backed, seconds at the model's real dimensions) instead of the C++ oracle; the two agree
    to 1e-10 (tests/test_oracle_agree.py)."""
    batch, params, masks = util.make_problem(sh, seed=seed, lens=lens, scale=scale)
    if hop_w is None:
        hop_w = np.full(sh.H, float(sh.H), np.float32)
    if torch_oracle:
        from oracle import ref_torch
        ref = ref_torch.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                             batch["labels"], masks if mode == "train" else None, hop_w)
    else:
        ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"],
                          batch["labels"], masks if mode == "train" else None, hop_w,
                          dtype=np.float64)
    got, layouts = run_gpu(sh, batch, params, masks, hop_w, mode)
    errs = {}
    for k in util.OUT_KEYS:
        errs[k] = util.rel_err(got[k], ref[k])
    for grp in ("embed", "rnn", "mult"):
        for name, sl in util.layer_slices(layouts[grp]):
            r = ref["g_" + grp][sl]
            if np.max(np.abs(r)) < 1e-12:   # e.g. att_score bias grad == 0 analytically
                errs[name] = float(np.max(np.abs(got["g_" + grp][sl] - r)))
            else:
                errs[name] = util.rel_err(got["g_" + grp][sl], r)
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, f"relative errors above {TOL}: {bad}\nall: {errs}"
    ok, decided, total = util.argmax_margin_ok(ref["logits"], got["argmax"], ref["argmax"])
    assert ok, "argmax mismatch on a decided row"
    print(f"argmax: {decided} of {total} rows decided at a 1e-5 margin, all equal; "
          f"{total - decided} undecided")
    return errs


def test_small_train():
    check(util.shapes(util.SMALL), scale=0.5)


def test_edge_minimal_dims():
    check(util.shapes(util.EDGE), scale=0.5)


def test_small_eval_mode():
    check(util.shapes(util.SMALL), mode="eval", scale=0.5)


def test_medium_ragged_s196_k1000():
    check(util.shapes(util.MEDIUM), scale=0.2)


def test_zero_length_rows_and_hop_gating():
    sh = util.shapes(util.SMALL)
    lens = np.array([0, 6, 1, 3, 6, 2, 0, 4], np.int32)
    check(sh, lens=lens, hop_w=np.array([1.0, 0.0, 1.0], np.float32), scale=0.5)


@pytest.mark.parametrize("w", [[1.0, 1.0, 0.0], [3.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
def test_trailing_hops_gated_off(w):
    """Full / ResNet late-epoch gating (Full:582-589): hops behind the last weighted one get no
    gradient at all; the library skips their backward, the result must not change."""
    sh = util.shapes(util.SMALL)
    check(sh, hop_w=np.array(w, np.float32), scale=0.5)
    check(sh, hop_w=np.array(w, np.float32), mode="eval", scale=0.5)


def test_resnet_like_channels_d2048():
    """cnnout_dim = 2048 (Ours_ResNet, reference ResNet:38,217): long K loop in i_embed."""
    dims = dict(B=6, T=4, V=40, E=200, Rq=32, D=2048, S=196, M=64, A=32, R=32, K=1000, H=2)
    check(util.shapes(dims), scale=0.05)


def test_batch_100_default_sizes_of_heads():
    """opt.batch_size default 100 (SS:48): tile edges in every flattened-column GEMM."""
    dims = dict(B=100, T=5, V=60, E=200, Rq=32, D=64, S=196, M=128, A=64, R=32, K=1000, H=2)
    check(util.shapes(dims), scale=0.1)


def test_7x7_feature_map_s49():
    """cnnout_w = cnnout_h = 7 is the scripts' own default (SS:36-37, 224x224 pool5 features):
    S = 49 is not a multiple of 4, the library pads the position pitch to 52 internally."""
    dims = dict(B=6, T=5, V=40, E=8, Rq=16, D=24, S=49, M=40, A=20, R=16, K=12, H=3)
    check(util.shapes(dims), scale=0.5)
    check(util.shapes(dims), mode="eval", scale=0.5)


def test_7x7_full_width_heads():
    dims = dict(B=20, T=5, V=60, E=200, Rq=32, D=512, S=49, M=128, A=64, R=32, K=1000, H=2)
    check(util.shapes(dims), scale=0.1)


def test_snapshot_round_trip_through_t7(tmp_path):
    """torch.save(checkpoint) / embed_param:copy(snap.params[1]) (SS:1188-1197, Eval.lua:344-347)."""
    from rau_vqa_amd.model import RAU, Config
    sh = util.shapes(util.SMALL)
    cfg = Config(**{k: getattr(sh, k) for k in
                    ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H")})
    a, b = RAU(cfg), RAU(cfg)
    a.init_uniform(seed=5)
    b.init_uniform(seed=6)
    p = tmp_path / "snapshot_iter000010_epoch0.01.t7"
    a.save_snapshot(p, it=10, epoch=0.01, opt={"nhop": sh.H, "alg_name": "Ours_SS"})
    it, epoch, opt = b.load_snapshot(p)
    assert (it, opt["nhop"]) == (10, sh.H)
    pa, pb = a.get_params(), b.get_params()
    for g in pa:
        np.testing.assert_array_equal(pa[g], pb[g])
    a.close()
    b.close()


def test_multfeat_wider_than_every_recurrent_width():
    """M = 128 beside R = 8, Rq = 4, K = 32 (found by tools/soak.py): the split-K workspace of the hop
    loop used to be sized from 4R / 4Rq / K / Q only, and the deferred dj GEMM ([B, M] partials) did
    not fit -> `kernel small_gemm: invalid argument`."""
    dims = dict(B=18, T=6, V=25, E=20, Rq=4, D=136, S=4, M=128, A=20, R=8, K=32, H=4)
    check(util.shapes(dims), scale=0.3, mode="eval")
    check(util.shapes(dims), scale=0.3, mode="train")


def test_explicit_masks_use_the_configured_p_not_its_1_256_quantisation():
    """Caller-supplied masks are nn.Dropout's: kept elements are scaled by 1/(1-p) with the
    CONFIGURED p (the device's own Philox masks quantise p to 1/256 and scale by that value, which
    is exact for the reference's 0.5).  p = 0.3 is not a multiple of 1/256 (0.3008 after rounding:
    a 0.1 % scale error before round 3), p = 0.001 rounds to 0 (the masks were ignored)."""
    for p in (0.3, 0.001):
        sh = util.shapes(util.SMALL, p_we=p, p_rnn=p, p_q=p, p_x=p, p_mf=p)
        check(sh, scale=0.5)
