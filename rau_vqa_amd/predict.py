"""Inference path: the reference's ``predict_result`` (SS:633-705) and the answer
selection of its eval loop (SS:877-897) on top of ``rau_forward`` in evaluate mode.

The heavy part (encoder + H hops; in evaluate mode dropout is the identity, so
i_embed and the attention pre-activation are hop-invariant and computed once) runs
in librau.so.  What is left is bookkeeping on [H,B,K] / [H,B,S] host arrays:
``do_pred > 0.5`` thresholding, "select" (first hop whose do_pred fires, last hop
forced, SS:683-688) and "uni" (mean over hops, SS:699-700) merging, multiple-choice
masking and first-max argmax.  The reference's quirks are reproduced, not fixed:
MC masking multiplies the RAW logits by a 0/1 mask (SS:894), so masked-out answers
become 0 and can beat negative valid logits.
"""
from __future__ import annotations

import numpy as np


def merge_hops(logits, dopred, att, select_att_state=None):
    """tab_pred / tab_att of SS:690-704: H per-hop entries, then uni, then select.

    select_att_state: the reference zeroes test_select_pred, test_uni_* and test_did_pred at the
    top of predict_result (SS:671-674) but NOT test_select_att, so its "select" attention map
    keeps accumulating across batches (it only feeds the attention PNGs).  Pass the array a
    previous call returned as tab_att[-1] to reproduce that; None starts from zeros (what the
    first batch sees)."""
    H, B, K = logits.shape
    did = np.zeros(B, np.float32)                       # test_did_pred:zero(), SS:674
    select_pred = np.zeros((B, K), np.float32)
    select_att = (np.zeros((B, att.shape[2]), np.float32) if select_att_state is None
                  else np.array(select_att_state, np.float32))   # never zeroed, SS:671-674
    for h in range(H):
        do = (dopred[h] > 0.5).astype(np.float32)       # SS:683
        if h == H - 1:
            do[:] = 1.0                                 # always predict in the final step, SS:685
        cur = np.clip(do - did, 0.0, 1.0)               # SS:686
        select_pred += logits[h] * cur[:, None]         # SS:687
        select_att += att[h] * cur[:, None]             # SS:688
        did = np.clip(did + do, 0.0, 1.0)               # SS:697
    uni_pred = logits.sum(0, dtype=np.float32) / np.float32(H)   # SS:680, 699
    uni_att = att.sum(0, dtype=np.float32) / np.float32(H)       # SS:681, 700
    tab_pred = [logits[h] for h in range(H)] + [uni_pred, select_pred]
    tab_att = [att[h] for h in range(H)] + [uni_att, select_att]
    return tab_pred, tab_att


def first_max(x):
    """torch.max(x, 2) index: FIRST maximal entry, 1-based (SS:896, 900)."""
    return np.argmax(x, axis=1).astype(np.int32) + 1


def answers(tab_pred, mc_ans=None):
    """Open-ended and multiple-choice answer ids (1-based) per entry of tab_pred.

    mc_ans: int array [B, nMultChoice] of candidate answer ids, 0 = empty slot
    (loader.lua:96, 1377).  Returns (oe [H+2, B], mc [H+2, B] or None).
    """
    oe = np.stack([first_max(p) for p in tab_pred])
    if mc_ans is None:
        return oe, None
    B, K = tab_pred[0].shape
    mask = np.zeros((B, K), np.float32)                 # test_mc_mask, SS:886-893
    for b in range(B):
        for a in mc_ans[b]:
            if a != 0:
                mask[b, a - 1] = 1.0
    mc = np.stack([first_max(p * mask) for p in tab_pred])   # cmul on raw logits, SS:894
    return oe, mc


def predict_result(rau, feats, tokens, lens, mc_ans=None, select_att_state=None):
    """SS:633-705 + SS:877-900 for one batch: returns dict(tab_pred, tab_att, oe, mc).
    select_att_state: see merge_hops (carry tab_att[-1] from batch to batch to reproduce the
    reference's never-zeroed test_select_att)."""
    rau.evaluate()
    rau.set_batch(feats, tokens, lens, None)
    rau.forward()
    tab_pred, tab_att = merge_hops(rau.logits(), rau.dopred(), rau.attention(), select_att_state)
    oe, mc = answers(tab_pred, mc_ans)
    return {"tab_pred": tab_pred, "tab_att": tab_att, "oe": oe, "mc": mc}
