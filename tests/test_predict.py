"""predict_result merging (SS:633-705) and MC masking (SS:884-897) against a
literal per-sample loop restatement."""
import numpy as np
import pytest

from rau_vqa_amd import predict


def loop_restatement(logits, dopred, att, mc_ans):
    H, B, K = logits.shape
    S = att.shape[2]
    sel_p, sel_a = np.zeros((B, K), np.float32), np.zeros((B, S), np.float32)
    for b in range(B):
        done = False
        for h in range(H):
            fire = dopred[h, b] > 0.5 or h == H - 1
            if fire and not done:
                sel_p[b] += logits[h, b]
                sel_a[b] += att[h, b]
                done = True
    uni_p = np.float32(0) + sum(logits[h] for h in range(H)) / np.float32(H)
    tab = [logits[h] for h in range(H)] + [uni_p, sel_p]
    oe = np.zeros((H + 2, B), np.int32)
    mc = np.zeros((H + 2, B), np.int32)
    for i, p in enumerate(tab):
        for b in range(B):
            best = 0
            for k in range(1, K):
                if p[b, k] > p[b, best]:
                    best = k
            oe[i, b] = best + 1
            cand = {a - 1 for a in mc_ans[b] if a != 0}
            masked = [p[b, k] if k in cand else np.float32(0) for k in range(K)]
            best = 0
            for k in range(1, K):
                if masked[k] > masked[best]:
                    best = k
            mc[i, b] = best + 1
    return sel_p, sel_a, uni_p, oe, mc


def test_merge_and_answers_match_loop_restatement():
    rng = np.random.default_rng(0)
    H, B, K, S = 4, 9, 12, 6
    logits = rng.standard_normal((H, B, K)).astype(np.float32)
    logits[:, 0, :] = -np.abs(logits[:, 0, :])          # all-negative row: MC quirk shows
    logits[1, 2, 3] = logits[1, 2, 7] = 5.0             # tie: first max wins
    dopred = rng.random((H, B)).astype(np.float32)
    dopred[:, 1] = 0.0                                   # never fires -> last hop forced
    dopred[:, 3] = 1.0                                   # fires at hop 1
    att = rng.random((H, B, S)).astype(np.float32)
    mc = rng.integers(0, K + 1, size=(B, 5)).astype(np.int32)
    tab_pred, tab_att = predict.merge_hops(logits, dopred, att)
    oe, mcans = predict.answers(tab_pred, mc)
    sel_p, sel_a, uni_p, oe_r, mc_r = loop_restatement(logits, dopred, att, mc)
    assert np.allclose(tab_pred[H + 1], sel_p) and np.allclose(tab_att[H + 1], sel_a)
    assert np.allclose(tab_pred[H], uni_p, rtol=1e-6)
    assert np.array_equal(oe, oe_r) and np.array_equal(mcans, mc_r)
    assert np.array_equal(tab_pred[H + 1][1], logits[H - 1, 1])   # forced last hop
    assert np.array_equal(tab_pred[H + 1][3], logits[0, 3])       # first firing hop
    assert oe[1, 2] == 4                                          # first of the tied maxima


@pytest.mark.gpu
def test_predict_result_on_device_matches_oracle_forward():
    import oracle
    from rau_vqa_amd.model import RAU, Config
    from tests import util
    sh = util.shapes(util.SMALL)
    batch, params, _ = util.make_problem(sh, scale=0.5)
    ref = oracle.step(sh, params, batch["feats"], batch["tokens"], batch["lens"], None, None,
                      backward=False, dtype=np.float64)
    m = RAU(Config(**{k: getattr(sh, k) for k in
                      ("B", "T", "V", "E", "Rq", "D", "S", "M", "A", "R", "K", "H")}))
    m.set_params(params)
    mc = np.random.default_rng(1).integers(0, sh.K + 1, size=(sh.B, 4)).astype(np.int32)
    out = predict.predict_result(m, batch["feats"], batch["tokens"], batch["lens"], mc)
    m.close()
    rp, ra = predict.merge_hops(ref["logits"].astype(np.float32), ref["dopred"].astype(np.float32),
                                ref["att"].astype(np.float32))
    for a, b in zip(out["tab_pred"], rp):
        assert util.rel_err(a, b) < 1e-4
    for a, b in zip(out["tab_att"], ra):
        assert util.rel_err(a, b) < 1e-4
    roe, rmc = predict.answers(rp, mc)
    assert np.array_equal(out["oe"], roe) and np.array_equal(out["mc"], rmc)


def test_select_att_accumulates_across_batches_like_the_reference():
    """SS:671-674 zero every merge buffer except test_select_att: carried over, it accumulates."""
    rng = np.random.default_rng(3)
    H, B, K, S = 3, 5, 7, 4
    lg = rng.standard_normal((H, B, K)).astype(np.float32)
    dp = rng.random((H, B)).astype(np.float32)
    att = rng.random((H, B, S)).astype(np.float32)
    _, ta1 = predict.merge_hops(lg, dp, att)
    _, ta2 = predict.merge_hops(lg, dp, att, select_att_state=ta1[-1])
    assert np.allclose(ta2[-1], 2.0 * ta1[-1])
    assert np.array_equal(ta2[-2], ta1[-2])            # uni_att IS zeroed every batch
