"""rau_vqa_amd/loader.py against the reference loader's contract
(utils/vqa_prepro_loader.lua:837-1010, 1219-1291, 1294-1473) on a synthetic dataset laid out
like data_prepro.{json,h5->npz} + per-image COCO_*.t7 feature files."""
import json

import numpy as np
import pytest

from rau_vqa_amd import loader, t7

N, T, D, W, H, NIMG, NMC = 23, 6, 4, 2, 2, 5, 3


@pytest.fixture()
def dataset(tmp_path):
    rng = np.random.default_rng(0)
    lens = rng.integers(1, T + 1, N)
    q = np.zeros((N, T), np.int64)
    for i, l in enumerate(lens):
        q[i, :l] = rng.integers(1, 9, l)          # 0 = padding in the prepro file
    imgs = [f"train2014/COCO_train2014_{i:012d}.jpg" for i in range(NIMG)]
    feats = {}
    fdir = tmp_path / "feat"
    fdir.mkdir()
    for i, name in enumerate(imgs):
        f = rng.standard_normal((D, W, H)).astype(np.float32)
        feats[i + 1] = f
        t7.save(fdir / loader.feature_name(name), f)
    np.savez(tmp_path / "data_prepro.npz", ques_train=q, ques_length_train=lens,
             img_pos_train=rng.integers(1, NIMG + 1, N), question_id_train=np.arange(N) + 1000,
             answers=rng.integers(1, 11, N), ques_test=q[:9], ques_length_test=lens[:9],
             img_pos_test=rng.integers(1, NIMG + 1, 9), question_id_test=np.arange(9) + 5000,
             MC_ans_test=rng.integers(1, 11, (9, NMC)))
    info = {"ix_to_word": {str(i): f"w{i}" for i in range(1, 9)},
            "ix_to_ans": {str(i): f"a{i}" for i in range(1, 11)},
            "unique_img_train": imgs, "unique_img_test": imgs}
    (tmp_path / "data_prepro.json").write_text(json.dumps(info))
    return tmp_path, str(fdir), q, lens, feats


def test_load_data_vocabulary_and_shift(dataset):
    root, fdir, q, lens, _ = dataset
    v = loader.load_data(str(root), batch_size=4)
    assert v.vocab_size == 9 and v.answer_size == 10 and v.seq_len == T     # + ZEROPAD
    assert v.vocab_dict[1] == "ZEROPAD" and v.vocab_dict[2] == "w1" and v.vocab_map["w8"] == 9
    np.testing.assert_array_equal(v.train_data.qs.question, q + 1)          # padding 0 -> id 1
    assert loader.feature_name("val2014/COCO_val2014_000000533942.jpg") == "COCO_val2014_000000533942.t7"


def test_next_batch_feat_contract_inorder(dataset):
    root, fdir, q, lens, feats = dataset
    v = loader.load_data(str(root), batch_size=4)
    tr = v.train_data
    tr.set_batch_order_option(2)
    tr.reorder()
    f, x, xl, a, qid = tr.next_batch_feat(fdir, D, W, H)
    assert f.shape == (4, D, W, H) and f.dtype == np.float32
    assert x.shape == (T, 4) and x.dtype == np.int32                        # transposed [T,B]
    np.testing.assert_array_equal(x.T, q[:4] + 1)
    np.testing.assert_array_equal(xl, lens[:4])
    np.testing.assert_array_equal(qid, np.arange(4) + 1000)
    for i in range(4):
        np.testing.assert_array_equal(f[i], feats[int(tr.qs.img_list[i])])
    te = v.test_data
    _, _, _, mc, _ = te.next_batch_feat([fdir], D, W, H)
    assert mc.shape == (4, NMC)                                             # mc_ans for test
    with pytest.raises(t7.T7Error, match="shape"):
        tr.next_batch_feat(fdir, D, W + 1, H)                               # the loader's asserts


def test_epoch_wrap_and_order_options(dataset):
    root, fdir, q, lens, _ = dataset
    v = loader.load_data(str(root), batch_size=4)
    tr = v.train_data
    tr.set_batch_order_option(2)
    tr.reorder()
    seen = []
    for _ in range(5):
        seen.append(tr.next_batch_feat(fdir, D, W, H)[4])
    assert tr.batch_index == 0            # 23 // 4 = 5 full batches, then the order is re-drawn
    np.testing.assert_array_equal(tr.next_batch_feat(fdir, D, W, H)[4], seen[0])
    tr.set_batch_order_option(3)
    tr.reorder()
    ls = lens[tr.batch_order]
    assert np.all(np.diff(ls) >= 0)
    tr.set_batch_order_option(4)
    tr.reorder()
    ls4 = lens[tr.batch_order]
    assert np.all(np.diff(ls4) >= 0) and sorted(tr.batch_order) == list(range(N))
    tr.set_batch_order_option(1)
    tr.reorder()
    assert sorted(tr.batch_order) == list(range(N))
    with pytest.raises(ValueError):
        tr.set_batch_order_option(7)


def test_prefetch_returns_the_same_batches(dataset):
    root, fdir, *_ = dataset
    a = loader.load_data(str(root), batch_size=4, prefetch=False, seed=7).train_data
    b = loader.load_data(str(root), batch_size=4, prefetch=True, seed=7).train_data
    for d in (a, b):
        d.set_batch_order_option(1)
        d.reorder()
    for _ in range(12):                    # crosses two epoch boundaries
        xa, xb = a.next_batch_feat(fdir, D, W, H), b.next_batch_feat(fdir, D, W, H)
        for u, w in zip(xa, xb):
            np.testing.assert_array_equal(u, w)


class _FakeRau:
    """Records what SlotFeeder does to the two upload slots (the device half is tested on the GPU in
    tests/test_gpu_async_batch.py)."""

    def __init__(self, B):
        self.slots = [{"feats": np.zeros((B, D, W * H), np.float32), "tokens": np.zeros((T, B), np.int32),
                       "lens": np.zeros(B, np.int32), "labels": np.zeros(B, np.int32)} for _ in range(2)]
        self.log, self.uploaded = [], [None, None]

    def batch_slot(self, s):
        return self.slots[s]

    def set_batch_async(self, s, has_labels=True):
        self.log.append(("upload", s))
        self.uploaded[s] = {k: v.copy() for k, v in self.slots[s].items()}
        self.uploaded[s]["has_labels"] = has_labels

    def use_batch(self, s):
        self.log.append(("use", s))
        self.current = self.uploaded[s]


def test_slot_feeder_alternates_slots_and_prefetches_in_place(dataset):
    """SlotFeeder = next_batch_feat's prefetch + the ctx's two pinned upload slots: batches arrive
    in the loader's order, alternate between the slots, and from the second batch on the worker
    has assembled the features directly in the slot's staging (no copy on the consumer side)."""
    root, fdir, q, lens, feats = dataset
    B = 4
    ref = loader.load_data(str(root), batch_size=B).train_data
    v = loader.load_data(str(root), batch_size=B)
    rau = _FakeRau(B)
    feeder = loader.SlotFeeder(rau, v.train_data, fdir, D, W, H)
    for it in range(9):                                   # runs over the epoch wrap (23 examples / 4)
        f, x, xl, a, qid = ref.next_batch_feat(fdir, D, W, H)
        cur = rau.current
        np.testing.assert_array_equal(cur["feats"], f.reshape(B, D, -1))
        np.testing.assert_array_equal(cur["tokens"], x)
        np.testing.assert_array_equal(cur["lens"], xl)
        np.testing.assert_array_equal(cur["labels"], a)
        np.testing.assert_array_equal(feeder.qids, qid)
        assert rau.log[-2:] == [("upload", it & 1), ("use", it & 1)]
        if it < 8:
            feeder.next()
    # test split: multiple-choice ids instead of labels
    rau2 = _FakeRau(3)
    v2 = loader.load_data(str(root), batch_size=4, test_batch_size=3)
    loader.SlotFeeder(rau2, v2.test_data, fdir, D, W, H)
    assert rau2.current["has_labels"] is False
